R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision 64 --steps 1500 --warmup 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f64 %.3f us (kernel %.3f)' % (1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in base sched_max-ilp sched_iterative-minreg; do run $v DartHopper-v1; run $v DartWalker2d-v1; done; done
