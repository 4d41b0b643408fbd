#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s8; mkdir -p $O
cd $R
for v in nolim base; do
  if [ "$v" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$v.so; else unset DART_STEPPER_LIB; fi
  for p in 32 64; do
    echo "== $v f$p"; python bench.py --no-extras --precision $p --steps 500 --warmup 100 --stats 2>&1 | grep "pivoting\|\"metric\"" | cut -c1-330
  done
done > $O/hopper_stats.txt 2>&1
cat $O/hopper_stats.txt
