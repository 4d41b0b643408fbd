# the default bench line with the counters of the final tree (profiles/pmc_traffic.json, flops_per_env_step.json stamped with its hashes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench default rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/r04_bench_default.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r.get('stale'), (r.get('valu') or {}).get('frac'))
for o in d.get('other_configs', []): print('   ', o.get('value'), o.get('ms_per_step'), (o.get('roofline',{}).get('valu') or {}).get('frac'), o.get('roofline',{}).get('stale'))
"
