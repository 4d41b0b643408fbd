#!/bin/bash
# round 6, GPU session 14: all-limits tier as a real call -- Walker2d timing, PMC traffic, parity gates
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s14; mkdir -p $O
cd $R
ONLY=walker2d bash tools/gpu/ab_bench.sh base > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
cd /tmp && export TMPDIR=/tmp
for p in 64 32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/w2d_f${p}_pmc_fetch -- python $R/bench.py --no-extras --precision $p --env-id DartWalker2d-v1 --steps 50 --warmup 5 > $O/pmc.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w2d_f${p}_pmc_write -- python $R/bench.py --no-extras --precision $p --env-id DartWalker2d-v1 --steps 50 --warmup 5 > $O/pmc.log 2>&1
done
cd $R
python tools/summarize_rocprof.py $O gpurun_out/r06_s14_rocprof.txt > /dev/null; find $O -name '*.db' -delete
grep "step_kernel" gpurun_out/r06_s14_rocprof.txt | cut -c1-150
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
