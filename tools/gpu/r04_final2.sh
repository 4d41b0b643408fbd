# Round-4 closing pass, second half (after the device Gaussian resets): the new GPU test, the other tasks, both bench lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_cr_log.py -x -q -p no:cacheprovider 2>&1 | tail -2
bash tools/gpu/bench_other_tasks.sh r04 > /dev/null 2>&1; cat gpurun_out/r04_bench_other_tasks.txt
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "bench driver rc=$?"
python -c "
import json
for f in ('gpurun_out/r04_bench_default.json','gpurun_out/r04_bench_driver.json'):
    d=json.load(open(f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r.get('stale'), (r.get('valu') or {}).get('frac'))
"
python -c "import __graft_entry__ as g; g.smoke()"
