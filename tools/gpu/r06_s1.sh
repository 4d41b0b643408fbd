#!/bin/bash
# round 6, GPU session 1: is the tree green on this box; fresh phase profile of the tree kernel; iteration-cap ablation of the headline kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s1; mkdir -p $O
cd $R
python tools/gpu/hopper_iter_ablation.py DartHopper-v1 > $O/hopper_iter_ablation.txt 2>&1
STEPS=500 python tools/gpu/hopper_iter_ablation.py DartWalker2d-v1 > $O/walker2d_iter_ablation.txt 2>&1
for p in 64 32; do PREC=$p python tools/diag_spatial_stats.py > $O/humanwalker_phases_f$p.txt 2>&1; done
python bench.py --no-extras > $O/bench_hopper.txt 2>&1
python bench.py --no-extras --env-id DartHumanWalker-v1 --steps 40 --warmup 3 > $O/bench_hw.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cat $O/hopper_iter_ablation.txt $O/walker2d_iter_ablation.txt $O/humanwalker_phases_f64.txt
grep -h '"metric"' $O/bench_hopper.txt $O/bench_hw.txt | cut -c1-200
