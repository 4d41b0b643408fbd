cd $GRAFT_REPO_ROOT
python -c "
from dart_env_amd.model_card import card_for
from dart_env_amd import stepper as st
for e in ['DartHopper-v1','DartWalker2d-v1']:
    for p in (32,64):
        s=st.HipStepper(card_for(e),64,precision=p); print(e,p,'static kernel:',s.query(st.Q_STATIC_KERNEL)); s.close()
" 2>&1 | grep -v amdgpu
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for e in DartHopper-v1 DartWalker2d-v1; do
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --env-id $e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', '%.3e steps/s'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
DART_GENERIC_KERNEL=1 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --env-id $e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e generic', '%.3e steps/s'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --precision 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hopper f64', '%.3e steps/s'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
