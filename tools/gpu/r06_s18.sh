#!/bin/bash
# round 6, GPU session 18: machine-scheduler strategies -- gcn-max-ilp (device side only) and schedule-metric-bias 0 against the default (gcn-max-occupancy)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s18; mkdir -p $O
cd $R
bash tools/gpu/ab_bench.sh base ilp bias0 > $O/ab_lane.txt 2>&1; cat $O/ab_lane.txt
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 16384 --steps 40 --warmup 3 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f$3 %.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in base ilps; do run $v DartHumanWalker-v1 64; run $v DartHumanWalker-v1 32; run $v DartDog-v1 64; run $v DartWalker3d-v1 64; done; done | tee $O/ab_tree.txt
