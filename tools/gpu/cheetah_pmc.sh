# PMC passes of the half-cheetah step kernel (each counter group in its own run, kernel trace only): HBM bytes per launch and issue statistics
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hc_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-extras --env-id DartHalfCheetah-v1 --envs 65536"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"
for p in 64 32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/halfcheetah_f${p}_pmc_fetch -- $B --precision $p --steps 40 --warmup 20 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/halfcheetah_f${p}_pmc_write -- $B --precision $p --steps 40 --warmup 20 > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/halfcheetah_f${p}_pmc_sq -- $B --precision $p --steps 40 --warmup 20 > $OUT/pmc.log 2>&1
done
cd $R
python tools/summarize_rocprof.py $OUT gpurun_out/hc_pmc_rocprof.txt > /dev/null
find $OUT -name '*.db' -delete
grep "step_kernel" gpurun_out/hc_pmc_rocprof.txt | cut -c1-160
