# Round-4 closing pass on one box: profile (rocprof stats + PMC), the two bench lines, the other tasks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_round.sh r04 > $R/gpurun_out/r04_profile.log 2>&1
cd $R
python tools/update_pmc_traffic.py gpurun_out/r04_rocprof.txt profiles/r04_rocprof.txt > gpurun_out/r04_update_pmc.log 2>&1; cp profiles/pmc_traffic.json gpurun_out/r04_pmc_traffic.json
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "bench driver rc=$?"
bash tools/gpu/bench_other_tasks.sh r04 > /dev/null 2>&1; cat gpurun_out/r04_bench_other_tasks.txt
cut -c1-400 gpurun_out/r04_bench_driver.json
