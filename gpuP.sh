cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for env in DartWalker2d-v1 DartHopper-v1; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM -d /tmp/p_$env -- python $R/bench.py --env-id $env --steps 20 --warmup 5 --no-cpu-baseline > /tmp/p1.log 2>&1
cd $R; python tools/summarize_rocprof.py /tmp/p_$env /tmp/s_$env; cat /tmp/s_$env* | grep -E "step_kernel" | sed "s/  */ /g" | cut -c1-40,100-200; cd /tmp
done
