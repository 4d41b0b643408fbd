#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s5; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
echo "== host path"
timeout 600 python tools/gpu/host_path_c.py 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/host_path_c.txt
timeout 600 python tools/gpu/host_step_breakdown.py 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/host_step_breakdown.txt
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (driver window)"; python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.json; echo
echo "== done"
