#!/bin/bash
# round 6, GPU session 7: compacted joint-limit rows (topo_limit_slots) -- A/B against one row per limited joint, parity gates
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s7; mkdir -p $O
cd $R
ONLY=hopper bash tools/gpu/ab_bench.sh nolim base > $O/ab_hopper.txt 2>&1; cat $O/ab_hopper.txt
ONLY=walker2d bash tools/gpu/ab_bench.sh nolim base w3 > $O/ab_walker2d.txt 2>&1; cat $O/ab_walker2d.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
