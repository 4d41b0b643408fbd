#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s34; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_host_buffers.py tests/test_gpu_golden_and_properties.py tests/test_gpu_bench_dist.py -q -p no:cacheprovider 2>&1 | tail -2
python tools/gpu/mt_fused_ab.py 2>&1 | grep -v amdgpu | grep "fused" | tee $O/host.txt
python tools/gpu/teardown_stress.py 300 2>&1 | tail -1
