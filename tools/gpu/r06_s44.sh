#!/bin/bash
# round 6, closing: the default bench line on the re-stamped tree, then the WHOLE suite five more times with fd 2 kept and the blocks traced
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py > gpurun_out/r06_bench_default.log 2>&1; grep '"metric"' gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_default.json; cut -c1-300 gpurun_out/r06_bench_default.json
bash tools/gpu/r06_s41.sh 5 e plain 1
