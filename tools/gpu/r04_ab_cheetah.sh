# Round-4 A/B, half cheetah fp64: lanes beyond the small register tier served by the whole wave when a wave has at most K of them
# (-DDART_F64_WAVE_VOTE=K; base = 0 = the big fp64 tier for every such wave, round 3).  One box, alternating libraries.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id DartHalfCheetah-v1 --precision $2 --envs 65536 --steps 150 --warmup $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 DartHalfCheetah-v1 f$2 warmup $3: %.3f ms (kernel %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in ${VARIANTS:-base vote2 vote4 vote8}; do run $v 64 20; run $v 64 200; done; done
