# Round-4 A/B: in-tree library vs abtest/lib_${V:-pipe}.so (rsqrt chain of the systolic Cholesky software-pipelined), times + bitwise states
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARIANTS="base ${V:-pipe}" bash $R/tools/gpu/r04_ab_tree.sh 2>&1 | grep -v amdgpu
for e in DartHumanWalker-v1 DartWalker3d-v1; do for p in 64 32; do
ENV_ID=$e N=2048 STEPS=12 PREC=$p python $R/tools/gpu/ab_states.py /tmp/a.npz 2>&1 | grep -v amdgpu
ENV_ID=$e N=2048 STEPS=12 PREC=$p DART_STEPPER_LIB=$R/abtest/lib_${V:-pipe}.so python $R/tools/gpu/ab_states.py /tmp/b.npz 2>&1 | grep -v amdgpu
python -c "
import numpy as np
a=np.load('/tmp/a.npz'); b=np.load('/tmp/b.npz')
print('$e f$p bitwise equal states:', np.array_equal(a['q'],b['q']) and np.array_equal(a['dq'],b['dq']), 'max |dq diff|', float(np.nanmax(np.abs(a['dq']-b['dq']))))
"; done; done
