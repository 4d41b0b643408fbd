#!/bin/bash
# round 6, GPU session 17: where a wave's cycles go in the lane kernels (DART_WAVE_TIMING build of this tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s17; mkdir -p $O
cd $R
for e in DartHopper-v1 DartWalker2d-v1 DartHalfCheetah-v1; do bash tools/gpu/wave_timing.sh wt $e 2>&1 | grep -v amdgpu; done | tee $O/wave_timing.txt
