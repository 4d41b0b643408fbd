#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02p3; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_long_parity.py -m gpu -q -s > $OUT/long_parity.log 2>&1; grep -E "^Dart|passed|failed|Error|assert" $OUT/long_parity.log | cut -c1-600
