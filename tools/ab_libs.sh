#!/bin/bash
# usage: ab.sh libA libB  -> env-steps/s of several envs for both libs
for e in "DartHumanWalker-v1 16384 30" "DartWalker3d-v1 16384 30" "DartDog-v1 16384 60" "DartHalfCheetah-v1 65536 30" "DartCartPole-v1 65536 100" "DartSnake7Link-v1 65536 40"; do
  set -- $e
  for lib in "$LIBA" "$LIBB"; do
    v=$(DART_STEPPER_LIB=$lib python bench.py --env-id $1 --envs $2 --steps $3 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.3f' % (d['value'], d['roofline']['kernel_ms']))")
    echo "$1 $(basename $lib) $v"
  done
done
