R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id DartHalfCheetah-v1 --precision $2 --envs 65536 --steps 200 --warmup 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 DartHalfCheetah-v1 f$2: %.3f ms (kernel %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in base f32vote2 f32vote3; do run $v 32; done; run base 64; done
python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "cheetah or repeat or lane_kernels_added" 2>&1 | tail -3
