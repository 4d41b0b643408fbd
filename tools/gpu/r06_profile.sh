#!/bin/bash
# round 6 profile: rocprofv3 --stats + PMC passes of the three BASELINE configs (tools/profile_round.sh), the half cheetah's PMC passes, the other tasks' bench lines,
# the bench lines of the default and the driver's invocation
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -2 gpurun_out/r06_profile_round.log | cut -c1-200
bash tools/gpu/cheetah_pmc.sh > gpurun_out/r06_cheetah_pmc.log 2>&1; tail -4 gpurun_out/r06_cheetah_pmc.log
bash tools/gpu/bench_other_tasks.sh r06 > /dev/null 2>&1; cat gpurun_out/r06_bench_other_tasks.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.log 2>&1; grep '"metric"' gpurun_out/r06_bench_driver.log > gpurun_out/r06_bench_driver.json; cut -c1-400 gpurun_out/r06_bench_driver.json
python bench.py > gpurun_out/r06_bench_default.log 2>&1; grep '"metric"' gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_default.json; cut -c1-300 gpurun_out/r06_bench_default.json
