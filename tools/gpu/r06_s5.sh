#!/bin/bash
# round 6, GPU session 5: crash hunt under GCC's ASan runtime; the baked half-cheetah model (CheetahStatic) against the runtime-parameter kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s5; mkdir -p $O
cd $R
bash tools/gpu/crash_hunt_asan.sh 500 2>&1 | tee $O/crash_hunt.txt
ONLY=cheetah bash tools/gpu/ab_bench.sh nostatic base > $O/ab_cheetah_static.txt 2>&1; cat $O/ab_cheetah_static.txt
python - > $O/static_query.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
for env in ("DartHalfCheetah-v1", "DartHopper-v1", "DartWalker2d-v1"):
    for p in (64, 32):
        g = st.HipStepper(card_for(env), 64, precision=p); print(env, p, "compile-time model:", g.query(st.Q_STATIC_KERNEL)); g.close()
PY
cat $O/static_query.txt
timeout 1500 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_golden_and_properties.py tests/test_gpu_repeatability.py tests/test_gpu_first_launch.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "cheetah or Cheetah or mt19937 or halfcheetah" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python tools/gpu/cheetah_floor_probe.py > $O/cheetah_floor.txt 2>&1; cat $O/cheetah_floor.txt
