#!/bin/bash
# Round profile: rocprofv3 kernel-trace --stats and PMC passes of bench.py (run on the GPU box through gpurun).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/..., summary gpurun_out/<tag>_rocprof.txt
# Counters are collected in their own passes (FETCH_SIZE / WRITE_SIZE / SQ_*), never together with a trace domain other than the kernel trace.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/tools/source_hash.py > $OUT/source_hash.json   # ties every counter of this run to the kernel sources of the library that ran
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-extras"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"
for p in 64 32; do
  rocprofv3 --kernel-trace --stats -d $OUT/hopper_f${p}_trace -- $B --precision $p --steps 2000 --warmup 200 > $OUT/hopper_f${p}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/hopper_f${p}_pmc_fetch -- $B --precision $p --steps 100 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/hopper_f${p}_pmc_write -- $B --precision $p --steps 100 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/hopper_f${p}_pmc_sq -- $B --precision $p --steps 100 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/walker2d_f${p}_trace -- $B --precision $p --env-id DartWalker2d-v1 --steps 500 --warmup 50 > $OUT/w2d_f${p}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/walker2d_f${p}_pmc_fetch -- $B --precision $p --env-id DartWalker2d-v1 --steps 50 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/walker2d_f${p}_pmc_write -- $B --precision $p --env-id DartWalker2d-v1 --steps 50 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/walker2d_f${p}_pmc_sq -- $B --precision $p --env-id DartWalker2d-v1 --steps 50 --warmup 5 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/humanwalker_f${p}_trace -- $B --precision $p --env-id DartHumanWalker-v1 --steps 40 --warmup 3 > $OUT/hw_f${p}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/humanwalker_f${p}_pmc_fetch -- $B --precision $p --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/humanwalker_f${p}_pmc_write -- $B --precision $p --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/humanwalker_f${p}_pmc_sq -- $B --precision $p --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/humanwalker_f${p}_pmc_lanes -- $B --precision $p --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $OUT/pmc.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/walker3d_f32_trace -- $B --precision 32 --env-id DartWalker3d-v1 --envs 16384 --steps 40 --warmup 3 > $OUT/w3_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/halfcheetah_f32_trace -- $B --precision 32 --env-id DartHalfCheetah-v1 --envs 65536 --steps 200 --warmup 20 > $OUT/hc_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/halfcheetah_f64_trace -- $B --precision 64 --env-id DartHalfCheetah-v1 --envs 65536 --steps 100 --warmup 10 > $OUT/hc64_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/snake_f64_trace -- $B --precision 64 --env-id DartSnake7Link-v1 --envs 65536 --steps 500 --warmup 50 > $OUT/snake_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/cartpole_f64_trace -- $B --precision 64 --env-id DartCartPole-v1 --envs 65536 --steps 1000 --warmup 50 > $OUT/cart_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dog_f32_trace -- $B --precision 32 --env-id DartDog-v1 --envs 16384 --steps 40 --warmup 3 > $OUT/dog_trace.log 2>&1
cd $R
python tools/summarize_rocprof.py $OUT gpurun_out/${TAG}_rocprof.txt > /dev/null
find $OUT -name '*.db' -delete   # raw rocpd databases (~60 MB): gpurun copies back at most 64 MiB
grep -h '"metric"' $OUT/hopper_f64_trace.log | cut -c1-300
