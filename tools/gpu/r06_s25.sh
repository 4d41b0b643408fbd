#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s25; mkdir -p $O
cd $R
bash tools/gpu/ab_bench.sh pre base > $O/ab.txt 2>&1; cat $O/ab.txt
