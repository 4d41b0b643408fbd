#!/bin/bash
# round 6, GPU session 12: limit slots, final form (half cheetah: lanes beyond the slots to the wave solvers; Walker2d: wave vote) -- timing, full suite, floor probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s12; mkdir -p $O
cd $R
ONLY=walker2d bash tools/gpu/ab_bench.sh base > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
ONLY=cheetah bash tools/gpu/ab_bench.sh c6 base > $O/ab_cheetah.txt 2>&1; cat $O/ab_cheetah.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
python tools/gpu/cheetah_floor_probe.py 2>&1 | grep -v amdgpu | tee $O/cheetah_floor.txt
