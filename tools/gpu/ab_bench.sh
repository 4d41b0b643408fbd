#!/bin/bash
# A/B of library variants on the bench configurations: tools/gpu/ab_bench.sh <variant> [<variant> ...]   (abtest/lib_<variant>.so; "base" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab; mkdir -p $O
run() {  # variant env precision envs steps
  if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs $4 --steps $5 --warmup 50 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-22s f%s  %.4f ms  %.3e' % ('$1', '$2', '$3', d['ms_per_step'], d['value']))"
}
for rep in 1 2; do
for v in "$@"; do
  if [ -z "$ONLY" ] || [ "$ONLY" = hopper ]; then run $v DartHopper-v1 64 65536 2000; run $v DartHopper-v1 32 65536 2000; fi
  if [ -z "$ONLY" ] || [ "$ONLY" = walker2d ]; then run $v DartWalker2d-v1 64 65536 500; run $v DartWalker2d-v1 32 65536 500; fi
  if [ -z "$ONLY" ] || [ "$ONLY" = cheetah ]; then run $v DartHalfCheetah-v1 64 65536 100; run $v DartHalfCheetah-v1 32 65536 100; fi
done
done | tee $O/ab_$(echo "$@" | tr ' ' '_').txt
