#!/bin/bash
# round 6, GPU session 9: Hopper with 2 limit slots and NO fallback tier (timing only: is the second inlined tier what costs?), half cheetah 3 limit slots vs 6
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s9; mkdir -p $O
cd $R
ONLY=hopper bash tools/gpu/ab_bench.sh h2nofb base > $O/ab_hopper.txt 2>&1; cat $O/ab_hopper.txt
ONLY=cheetah bash tools/gpu/ab_bench.sh c6 base > $O/ab_cheetah.txt 2>&1; cat $O/ab_cheetah.txt
ONLY=walker2d bash tools/gpu/ab_bench.sh base > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
