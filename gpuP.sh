cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM -d /tmp/p1 -- python $R/bench.py --env-id DartHumanWalker-v1 --steps 4 --warmup 1 --no-cpu-baseline > /tmp/p1.log 2>&1
cd $R
python tools/summarize_rocprof.py /tmp/p1 /tmp/s1
cat /tmp/s1* | grep -E "sp_step" | sed "s/  */ /g"
