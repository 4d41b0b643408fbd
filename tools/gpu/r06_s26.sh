#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s26; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; echo "suite: $(tail -1 $O/pytest_gpu.txt)"
bash tools/gpu/r06_profile.sh
