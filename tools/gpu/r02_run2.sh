#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02p2; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_long_parity.py -m gpu -x -q -s > $OUT/long_parity.log 2>&1; tail -15 $OUT/long_parity.log
timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_long_parity.py > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
(time timeout 900 python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; head -c 3000 $OUT/bench_default.json
(time timeout 300 python bench.py --steps 20 --warmup 5) > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -3 $OUT/bench_driver.err
