# Round-4 last GPU pass (budget: ~3 minutes): the repeatability suite after the lane kernels' compile-time tables, the LDS-threshold test,
# the round profile (stamps), the bench lines, smoke.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_repeatability.py "tests/test_gpu_spatial.py::test_tree_kernel_lds_block_on_the_device_matches_the_thresholds" -q -p no:cacheprovider > gpurun_out/r04_pytest_last.log 2>&1; tail -3 gpurun_out/r04_pytest_last.log
bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1; tail -1 gpurun_out/r04_profile.log | cut -c1-200
cd $R
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "bench driver rc=$?"
