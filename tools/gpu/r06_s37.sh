#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s37; mkdir -p $O
cd $R
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 16384 --steps 40 --warmup 3 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f$3 %.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in base thint; do run $v DartHumanWalker-v1 64; run $v DartHumanWalker-v1 32; run $v DartDog-v1 64; run $v DartWalker3d-v1 64; done; done | tee $O/ab_tree.txt
