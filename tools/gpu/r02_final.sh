#!/bin/bash
# round-2 closing run: the whole -m gpu suite (incl. the 1 000-step untrimmed parity gates), the smoke entry, bench.py with its default and with the driver's flags
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02fin; mkdir -p $OUT; cd $R
(time timeout 1500 python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
(time timeout 900 python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
(time timeout 300 python bench.py --steps 20 --warmup 5) > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -3 $OUT/bench_driver.err
head -c 1500 $OUT/bench_driver.json
