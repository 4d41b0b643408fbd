#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s3; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
echo "== host path (ring of action batches)"
timeout 600 python tools/gpu/host_path_c.py 2>&1 | grep -v "^$" | tee $O/host_path_c.txt
timeout 600 python tools/gpu/host_step_breakdown.py 2>&1 | grep -v "^$" | tee $O/host_step_breakdown.txt
bash tools/gpu/host_latency.sh > $O/host_latency.log 2>&1; cat $R/gpurun_out/host_latency/probe.txt $R/gpurun_out/host_latency/trace_gaps.txt
PROBE_CONST=1 PROBE_TAG=constant-actions python tools/gpu/host_latency_probe.py 2>&1 | grep '^\[' | tee $O/probe_const.txt
echo "== bench quick lines"
for e in DartHopper-v1 DartWalker2d-v1; do python bench.py --no-extras --env-id $e --steps 500 --warmup 100 2>/dev/null | tail -1 | cut -c1-400; done
python bench.py --no-extras --env-id DartHumanWalker-v1 --steps 40 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
echo "== done"
