#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s27; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; echo "suite: $(tail -1 $O/pytest_gpu.txt)"; grep FAILED $O/pytest_gpu.txt | head
ONLY=walker2d bash tools/gpu/ab_bench.sh base 2>&1 | tail -2
