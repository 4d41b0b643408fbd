#!/bin/bash
# round 6, GPU session 2: the tree after the vote / second-tier deletion, per-solve iteration caps, late H^-1 parking (cheetah) and the fused MT19937 reset
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s2; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
ONLY=cheetah bash tools/gpu/ab_bench.sh ch0 base ch3 > $O/ab_cheetah.txt 2>&1; cat $O/ab_cheetah.txt
python tools/gpu/mt_fused_ab.py > $O/mt_fused_ab.txt 2>&1; cat $O/mt_fused_ab.txt
python tools/gpu/cheetah_floor_probe.py > $O/cheetah_floor.txt 2>&1; cat $O/cheetah_floor.txt
python bench.py --no-extras > $O/bench_hopper.txt 2>&1; grep -h '"metric"' $O/bench_hopper.txt | cut -c1-160
