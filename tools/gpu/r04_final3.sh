# Round-4 closing pass after the tree-kernel work of the second half (Gauss-Jordan pivoting solver, tied broadcasts, LDS-typed pointers):
# the whole GPU suite, the round profile (counters stamped with the source hashes of this tree), the other tasks, both bench lines, smoke.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04_pytest_gpu.log 2>&1; tail -3 gpurun_out/r04_pytest_gpu.log
bash tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log
cd $R
bash tools/gpu/bench_other_tasks.sh r04 > /dev/null 2>&1; cat gpurun_out/r04_bench_other_tasks.txt
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "bench driver rc=$?"
python -c "
import json
for f in ('gpurun_out/r04_bench_default.json','gpurun_out/r04_bench_driver.json'):
    d=json.load(open(f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r.get('stale'), (r.get('valu') or {}).get('frac'))
    for o in d.get('other_configs', []): print('   ', o.get('env_id') or o.get('workload'), o.get('precision'), o.get('value'), o.get('ms_per_step'))
"
python -c "import __graft_entry__ as g; g.smoke()"
