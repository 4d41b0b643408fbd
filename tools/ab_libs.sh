#!/bin/bash
# A/B of product-library builds on one box: tools/ab_libs.sh "<env envs steps prec>;..." libA.so libB.so ...
#   -> kernel ms per batched step (HIP events) for every config x library
CFGS=${1:-"DartHumanWalker-v1 16384 30 32;DartHumanWalker-v1 16384 20 64;DartWalker3d-v1 16384 30 32;DartDog-v1 16384 60 32;DartHalfCheetah-v1 65536 30 32"}
shift
IFS=';' read -ra LIST <<< "$CFGS"
for e in "${LIST[@]}"; do
  set -- $e
  line="$1 n=$2 f$4:"
  for lib in "${@:5}" $LIBS; do :; done
  for lib in $LIBS; do
    v=$(DART_STEPPER_LIB=$lib python bench.py --env-id $1 --envs $2 --steps $3 --warmup 5 --precision $4 --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['kernel_ms'])")
    line="$line  $(basename $lib .so)=$v"
  done
  echo "$line"
done
