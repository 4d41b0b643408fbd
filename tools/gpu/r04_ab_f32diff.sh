R=${GRAFT_REPO_ROOT:-$(pwd)}
export ENV_ID=DartHumanWalker-v1 N=2048 STEPS=6 PREC=32
python $R/tools/gpu/ab_states.py /tmp/a.npz 2>&1 | grep -v amdgpu
python $R/tools/gpu/ab_states.py /tmp/a2.npz 2>&1 | grep -v amdgpu
DART_STEPPER_LIB=$R/abtest/lib_${V:-tie}.so python $R/tools/gpu/ab_states.py /tmp/b.npz 2>&1 | grep -v amdgpu
DART_STEPPER_LIB=$R/abtest/lib_${V:-tie}.so python $R/tools/gpu/ab_states.py /tmp/b2.npz 2>&1 | grep -v amdgpu
echo "base vs base:"; python $R/tools/gpu/ab_compare.py /tmp/a.npz /tmp/a2.npz
echo "tie vs tie:"; python $R/tools/gpu/ab_compare.py /tmp/b.npz /tmp/b2.npz
echo "base vs tie:"; python $R/tools/gpu/ab_compare.py /tmp/a.npz /tmp/b.npz
