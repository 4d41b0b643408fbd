#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s30; mkdir -p $O
cd $R
ONLY=hopper bash tools/gpu/ab_bench.sh base slowhint > $O/ab_h.txt 2>&1; cat $O/ab_h.txt
ONLY=walker2d bash tools/gpu/ab_bench.sh base slowhint > $O/ab_w.txt 2>&1; cat $O/ab_w.txt
