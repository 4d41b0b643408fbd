#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02p4; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log | cut -c1-300
for p in 64 32; do for abc in -1 0; do for env in DartHopper-v1 DartWalker2d-v1; do
  timeout 300 python bench.py --env-id $env --precision $p --all-bodies-collide $abc --no-extras --steps 1000 --warmup 200 > $OUT/b_${env}_f${p}_abc${abc}.json 2>$OUT/err.txt
  python - $OUT/b_${env}_f${p}_abc${abc}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1].split("/")[-1], "%.4g env-steps/s kernel_ms %.4g" % (d["value"], d["roofline"]["kernel_ms"]), d["config"]["contact_set"], d["config"]["compile_time_model"])
PY
done; done; done
