#!/bin/bash
# round 6, GPU session 3: LDS-broadcast Cholesky / substitution of the pattern tree kernel (SP_CHOL_LDS) against round 5's all-register scheme (chol0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s3; mkdir -p $O
cd $R
run() {  # variant precision
  if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python bench.py --no-extras --env-id DartHumanWalker-v1 --precision $2 --steps 40 --warmup 5 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s HumanWalker f%s  %.4f ms' % ('$1', '$2', d['ms_per_step']))"
}
for rep in 1 2; do for v in chol0 base; do run $v 64; run $v 32; done; done | tee $O/ab_humanwalker.txt
unset DART_STEPPER_LIB
timeout 1500 python -m pytest tests/test_gpu_first_launch.py tests/test_gpu_spatial.py tests/test_gpu_long_parity.py -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python tools/gpu/kernel_time_windows.py DartWalker2d-v1 DartHopper-v1 > $O/windows.txt 2>&1; cat $O/windows.txt
for p in 64; do PREC=$p python tools/diag_spatial_stats.py > $O/humanwalker_phases_f$p.txt 2>&1; head -2 $O/humanwalker_phases_f$p.txt | cut -c1-300; done
