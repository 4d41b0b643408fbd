#!/bin/bash
# tools/gpu/host_latency.sh -- the host-latency probe under the runtime's wait knobs + a kernel trace of the step-sync-step pattern
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/host_latency; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PROBE_TAG=default python $R/tools/gpu/host_latency_probe.py 2>&1 | grep '^\[' | tee $O/probe.txt
HSA_ENABLE_INTERRUPT=0 PROBE_TAG=HSA_ENABLE_INTERRUPT=0 python $R/tools/gpu/host_latency_probe.py 2>&1 | grep '^\[' | tee -a $O/probe.txt
ROC_ACTIVE_WAIT_TIMEOUT=2000 PROBE_TAG=ROC_ACTIVE_WAIT_TIMEOUT=2000 python $R/tools/gpu/host_latency_probe.py 2>&1 | grep '^\[' | tee -a $O/probe.txt
PROBE_ONLY=s1 rocprofv3 --kernel-trace --stats -d $O/trace -o probe -- python $R/tools/gpu/host_latency_probe.py > $O/trace_stdout.txt 2>&1
python $R/tools/gpu/trace_gaps.py $O/trace step_kernel 2>&1 | tee $O/trace_gaps.txt
find $O -name '*.db' -delete
