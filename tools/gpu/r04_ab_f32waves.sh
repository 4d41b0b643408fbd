# fp32 pattern kernel: 3 waves / SIMD (168 VGPRs, 476 B scratch; in-tree) vs 2 (255, 148 B) vs 1 (256 + 30 AGPRs, 16 B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id DartHumanWalker-v1 --precision 32 --envs 16384 --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 HumanWalker f32 %.3f ms (kernel %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in base f32w2 f32w1; do run $v; done; done
