#!/bin/bash
# round 6, final verification as the driver runs it: smoke, the GPU suite (twice: stability), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tee $O/smoke.txt
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu_$i.txt 2>&1; echo "suite run $i: $(tail -1 $O/pytest_gpu_$i.txt)"; done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $O/bench_driver.json; cut -c1-260 $O/bench_driver.json
python -c "
import json; p=json.load(open('$O/bench_driver.json')); r=p['roofline']; print('roofline', {k:r.get(k) for k in ('bound','achieved','peak','frac','traffic','stale')}); print('valu', r.get('valu',{}).get('frac'), r.get('valu',{}).get('stale_note')); print('cpu_baseline', p['cpu_baseline']['value'], p['cpu_baseline']['kind'], p['cpu_baseline']['cores'])"
ls $R/gpurun_out/pytest_faulthandler_*.log 2>/dev/null | head
