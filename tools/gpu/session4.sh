#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s4; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # variant env precision envs steps warmup
  if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  timeout 300 python $R/bench.py --no-extras --env-id $2 --precision $3 --envs $4 --steps $5 --warmup $6 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-22s f%s  %.4f ms  %.3e' % ('$1', '$2', '$3', d['roofline']['kernel_ms'], d['value']))"
}
echo "== planar A/B: coalesced observation stores + HostOut epilogue (base) against the library before it (pre_hostout)"
for rep in 1 2 3; do for v in base pre_hostout; do
  run $v DartHopper-v1 64 65536 2000 200; run $v DartHopper-v1 32 65536 2000 200; run $v DartWalker2d-v1 64 65536 500 50; run $v DartWalker2d-v1 32 65536 500 50
done; done 2>&1 | tee $O/ab_planar.txt
for v in base pre_hostout; do run $v DartHalfCheetah-v1 64 65536 100 20; run $v DartHalfCheetah-v1 32 65536 100 20; run $v DartSnake7Link-v1 64 65536 500 50; done 2>&1 | tee -a $O/ab_planar.txt
unset DART_STEPPER_LIB
echo "== host path"
timeout 600 python tools/gpu/host_path_c.py 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/host_path_c.txt
timeout 600 python tools/gpu/host_step_breakdown.py 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/host_step_breakdown.txt
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
echo "== done"
