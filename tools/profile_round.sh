#!/bin/bash
# Round profile: rocprofv3 kernel-trace --stats and PMC passes of bench.py (run on the GPU box through gpurun).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/..., summary gpurun_out/<tag>_rocprof.txt
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/hopper_trace -- $B --steps 2000 --warmup 200 > $OUT/hopper_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/hopper_pmc_fetch -- $B --steps 100 --warmup 5 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/hopper_pmc_write -- $B --steps 100 --warmup 5 > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $OUT/hopper_pmc_sq -- $B --steps 100 --warmup 5 > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/walker2d_trace -- $B --env-id DartWalker2d-v1 --steps 500 --warmup 50 > $OUT/w2d_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/humanwalker_trace -- $B --env-id DartHumanWalker-v1 --steps 40 --warmup 3 > $OUT/hw_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d $OUT/humanwalker_pmc_sq -- $B --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $OUT/hw_pmc.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/walker3d_trace -- $B --env-id DartWalker3d-v1 --envs 16384 --steps 40 --warmup 3 > $OUT/w3_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dog_trace -- $B --env-id DartDog-v1 --envs 16384 --steps 40 --warmup 3 > $OUT/dog_trace.log 2>&1
cd $R
python tools/summarize_rocprof.py $OUT gpurun_out/${TAG}_rocprof.txt > /dev/null
find $OUT -name '*.db' -delete   # raw rocpd databases (~60 MB): gpurun copies back at most 64 MiB
grep -h '"metric"' $OUT/hopper_trace.log | cut -c1-300
