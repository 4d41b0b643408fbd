#!/bin/bash
# round 6, GPU session 15: Hopper limit prefix (first two joints' limit rows in the small tier) against all three rows (hpfx3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s15; mkdir -p $O
cd $R
ONLY=hopper bash tools/gpu/ab_bench.sh hpfx3 base > $O/ab_hopper.txt 2>&1; cat $O/ab_hopper.txt
ONLY=walker2d bash tools/gpu/ab_bench.sh base > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
