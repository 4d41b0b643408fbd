#!/bin/bash
# kernel ms per batched step and env-steps/s of the env ids besides the three profiled ones, both precisions -> gpurun_out/<tag>_bench_other_tasks.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/${TAG}_bench_other_tasks.txt
: > $OUT
run() {  # env envs steps
  for p in 64 32; do
    python $R/bench.py --no-extras --env-id $1 --envs $2 --precision $p --steps $3 --warmup 20 2>/dev/null | grep '"metric"' | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-30s %-5s f$p %.4f %.3e' % ('$1', 'batch $2'[6:], d['ms_per_step'], d['value']))" >> $OUT
  done
}
run DartSnake7Link-v1 65536 500
run DartHalfCheetah-v1 65536 200
run DartCartPole-v1 65536 1000
run DartReacher-v1 65536 500
run DartReacher3d-v1 65536 300
run DartDoubleInvertedPendulumEnv-v1 65536 1000
run DartCartPoleSwingUp-v1 65536 1000
run DartWalker3d-v1 16384 40
run DartDog-v1 16384 40
run DartWalker3dSPD-v1 16384 20
cat $OUT
