# Round-4 last pass: both bench lines with the counters of this tree (pmc_traffic.json / flops_per_env_step.json freshly stamped), smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "bench driver rc=$?"
python -c "
import json
for f in ('gpurun_out/r04_bench_default.json','gpurun_out/r04_bench_driver.json'):
    d=json.load(open(f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r.get('stale'), (r.get('valu') or {}).get('frac'))
    for o in d.get('other_configs', []): print('   ', o.get('config',{}).get('workload','')[:40], o.get('value'), o.get('ms_per_step'), (o.get('roofline',{}).get('valu') or {}).get('frac'), o.get('roofline',{}).get('stale'))
"
python -c "import __graft_entry__ as g; g.smoke()"
