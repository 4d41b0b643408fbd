#!/bin/bash
# tools/gpu/first_launch.sh -- the first-launch experiment matrix (one fresh process per line); output: gpurun_out/first_launch.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
P="python $R/tools/gpu/first_launch_probe.py"
{
for lib in base ctab; do
  if [ $lib != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$lib.so; else unset DART_STEPPER_LIB; fi
  for prec in 32 64; do
    $P --prec $prec --poison none
    $P --prec $prec --poison none            # a second fresh process: is the FIRST rollout itself reproducible across processes?
    for k in scratch lds regs; do
      $P --prec $prec --poison $k --when later --pattern 0x7fc00000
      $P --prec $prec --poison $k --when both --pattern 0x7fc00000
      $P --prec $prec --poison $k --when both --pattern 0x00000000
    done
  done
  $P --prec 32 --report --poison none
done
} 2>&1 | grep -v "^$" | tee $O/first_launch.txt
