#!/bin/bash
# round 6, GPU session 20: per-contact-point Jacobian entries in the ground-contact tree kernels (jold = per-row entries, rounds 4-5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s20; mkdir -p $O
cd $R
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 16384 --steps 40 --warmup 3 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f$3 %.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in jold base; do run $v DartHumanWalker-v1 64; run $v DartHumanWalker-v1 32; run $v DartDog-v1 64; run $v DartDog-v1 32; done; done | tee $O/ab_tree.txt
unset DART_STEPPER_LIB
PREC=64 python tools/diag_spatial_stats.py 2>&1 | grep -v amdgpu | head -2 | cut -c1-300 | tee $O/phases.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
