R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f; mkdir -p $O
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision 64 --steps 2000 --warmup 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f64 %.3f us (kernel %.3f)' % (1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for v in base hophlds; do run $v DartHopper-v1; done; done > $O/ab_hopper_hlds.txt 2>&1; cat $O/ab_hopper_hlds.txt
unset DART_STEPPER_LIB
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
