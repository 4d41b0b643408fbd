#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s23; mkdir -p $O
cd $R
ONLY=walker2d bash tools/gpu/ab_bench.sh base nofb > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
ONLY=hopper bash tools/gpu/ab_bench.sh base nofb > $O/ab_hopper.txt 2>&1; cat $O/ab_hopper.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | python -c "
import json,sys; p=json.loads(sys.stdin.read()); r=p['roofline']; print('value', p['value'], 'traffic', r.get('traffic'), 'valu', r.get('valu',{}).get('frac'), r.get('valu',{}).get('stale_note'))"
