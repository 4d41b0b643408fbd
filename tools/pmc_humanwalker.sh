#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_hw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --env-id DartHumanWalker-v1 --steps 6 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -- $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/b -- $B > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d $OUT/c -- $B > $OUT/c.log 2>&1
cd $R
python tools/summarize_rocprof.py $OUT gpurun_out/pmc_hw.txt > /dev/null
find $OUT -name '*.db' -delete
grep "sp_step_kernel" gpurun_out/pmc_hw.txt
