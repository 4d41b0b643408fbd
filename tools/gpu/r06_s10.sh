#!/bin/bash
# round 6, GPU session 10: limit slots with the warm sets in joint layout -- timing and the full suite (batch independence across the vote)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s10; mkdir -p $O
cd $R
ONLY=walker2d bash tools/gpu/ab_bench.sh base > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
ONLY=cheetah bash tools/gpu/ab_bench.sh c6 base > $O/ab_cheetah.txt 2>&1; cat $O/ab_cheetah.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
