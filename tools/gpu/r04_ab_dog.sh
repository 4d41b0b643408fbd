R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision 64 --envs 16384 --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f64 %.3f ms (kernel %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in base sm2; do run $v DartDog-v1; done; done
