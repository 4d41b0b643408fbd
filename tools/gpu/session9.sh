# round 5, session 9: four-env wave solver after the compile-time tables and the one-instruction broadcasts: sections of a pass, product timing,
# the cheetah's step kernel with compile-time ancestor tables (c4ct: fp64 lint-clean, fp32 trips the toolchain defect -- timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s9; mkdir -p $O
cd $R
DART_STEPPER_LIB=$R/abtest/lib_c4f.so python bench.py --no-extras --env-id DartHalfCheetah-v1 --precision 64 --envs 65536 --steps 100 --warmup 20 --stats 2>&1 | grep "stage 1" | python -c "
import sys, json, re
a = json.loads(re.search(r'(\[.*\])', sys.stdin.read()).group(1))
n = max(a[28], 1)
print('passes %d (120 launches x 1024 waves): cycles per pass: rows + Y + A %.0f, stage 1 %.0f, stage 2 %.0f, velocity update %.0f; rows per pass (all groups) %.1f' % (a[28], a[24]/n, a[25]/n, a[26]/n, a[27]/n, a[29]/n))
" | tee $O/timing_c4f.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 -1 64 2>&1 | grep -v Warning | tee $O/probe_c4.txt
DART_STEPPER_LIB=$R/abtest/lib_c4ct.so python tools/gpu/cheetah_coop4_probe.py c4ct -1 64 2>&1 | grep -v Warning | tee $O/probe_c4ct.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so timeout 600 python -m pytest tests/test_gpu_wave_blcp.py -q -m gpu 2>&1 | tail -2 | tee $O/harness.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so timeout 1200 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py -q -m gpu -k "cheetah or Cheetah or wave_mates or vote" 2>&1 | tail -4 | tee $O/tests_c4.txt
