#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
for rep in lean report; do
  extra=""; [ $rep = report ] && extra="--report"
  DART_STEPPER_LIB=$R/abtest/lib_ctab.so timeout 300 python tools/gpu/first_launch_bisect.py --prec 32 $extra 2>&1 | grep -v "^$" | tee $O/bisect_ctab_f32_$rep.txt
done
timeout 300 python tools/gpu/first_launch_bisect.py --prec 32 2>&1 | tee $O/bisect_intree_f32_lean.txt
timeout 300 python tools/gpu/first_launch_bisect.py --prec 64 2>&1 | tee $O/bisect_intree_f64_lean.txt
timeout 300 python tools/gpu/first_launch_bisect.py --env DartHopper-v1 --prec 64 2>&1 | tee $O/bisect_hopper_f64.txt
timeout 300 python tools/gpu/first_launch_bisect.py --env DartWalker2d-v1 --prec 64 2>&1 | tee $O/bisect_walker2d_f64.txt
echo "== pytest -m gpu"; (time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
