# round 5, final session: the bench lines of the final tree (default invocation, the driver's window), the other tasks, the whole GPU suite, smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s14; mkdir -p $O
cd $R
python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python bench.py --steps 20 --warmup 5 2> $O/bench_driver.err | tail -1 > $O/bench_driver.json
bash tools/gpu/bench_other_tasks.sh r05 > /dev/null 2>&1; cp gpurun_out/r05_bench_other_tasks.txt $O/bench_other_tasks.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 2400 python -X faulthandler -m pytest tests -q -m gpu > $O/tests_full.txt 2>&1; echo "pytest rc=$?" | tee $O/tests.txt; tail -4 $O/tests_full.txt | cut -c1-200
cut -c1-400 $O/bench_default.json; cut -c1-400 $O/bench_driver.json
