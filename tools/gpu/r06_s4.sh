#!/bin/bash
# round 6, GPU session 4: crash hunt under host-side ASan, the full GPU suite on the cleaned tree, ring sensitivity of the Walker2d kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s4; mkdir -p $O
cd $R
bash tools/gpu/crash_hunt_asan.sh 300 2>&1 | tee $O/crash_hunt.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
for r in 8 16 64 256; do RING=$r NW=12 python tools/gpu/kernel_time_windows.py DartWalker2d-v1 DartHopper-v1 2>&1 | grep -v amdgpu.ids; done | tee $O/ring_windows.txt
