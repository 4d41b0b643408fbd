#!/bin/bash
# round-2 probe 1: baseline numbers the plan needs (fp64 speed of every config, tree-kernel phase breakdown, Walker2d PMC)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02p1
mkdir -p $OUT
cd $R
for p in 32 64; do
  PREC=$p timeout 300 python tools/diag_spatial_stats.py > $OUT/hw_phases_f$p.txt 2>&1
done
for cfg in "DartHopper-v1 2000 200" "DartWalker2d-v1 500 50" "DartHumanWalker-v1 30 3"; do
  set -- $cfg
  for p in 32 64; do
    timeout 300 python bench.py --env-id $1 --steps $2 --warmup $3 --precision $p --no-cpu-baseline > $OUT/bench_${1}_f$p.json 2> $OUT/bench_${1}_f$p.err
  done
done
timeout 300 python bench.py --env-id DartWalker2d-v1 --all-bodies-collide --envs 65536 --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_w2d_allcaps_f32.json 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --env-id DartWalker2d-v1 --steps 50 --warmup 5"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $OUT/w2d_pmc_sq -- $B > $OUT/w2d_pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/w2d_pmc_fetch -- $B > $OUT/w2d_pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w2d_pmc_write -- $B > $OUT/w2d_pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/w2d_pmc_valu -- $B > $OUT/w2d_pmc4.log 2>&1
cd $R
python tools/summarize_rocprof.py $OUT gpurun_out/r02p1_rocprof.txt > /dev/null 2>&1
find $OUT -name '*.db' -delete
head -c 600 $OUT/bench_*_f64.json
cat $OUT/hw_phases_f32.txt $OUT/hw_phases_f64.txt
