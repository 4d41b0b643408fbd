#!/bin/bash
# round 6, last session: output blocks moved to dart_alloc_output (hipHostMalloc).  Profile for the re-stamp, the tests that touch the blocks, then the
# hunting loop again (the suite's first quarter, 10 runs, fd 2 kept, blocks traced): the fault must be gone.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu/r06_profile.sh 2>&1 | tail -12 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_host_buffers.py tests/test_gpu_golden_and_properties.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
bash tools/gpu/r06_s41.sh 10 d plain 1 short
