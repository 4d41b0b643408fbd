# round 5, session 8: four-env wave solver with LDS-typed pointers -- timing builds (whole-wave sections; sections of a pass), product timing, tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8; mkdir -p $O
cd $R
bash tools/gpu/wave_timing.sh c4t DartHalfCheetah-v1 64 2>&1 | grep -v Warning | tee $O/timing_c4t.txt
DART_STEPPER_LIB=$R/abtest/lib_c4f.so python bench.py --no-extras --env-id DartHalfCheetah-v1 --precision 64 --envs 65536 --steps 100 --warmup 20 --stats 2>&1 | grep "stage 1" | python -c "
import sys, json, re
a = json.loads(re.search(r'(\[.*\])', sys.stdin.read()).group(1))
n = max(a[28], 1)
print('passes %d (120 launches x 1024 waves): cycles per pass: rows + Y + A %.0f, stage 1 %.0f, stage 2 %.0f, velocity update %.0f; rows per pass (all groups) %.1f' % (a[28], a[24]/n, a[25]/n, a[26]/n, a[27]/n, a[29]/n))
" | tee $O/timing_c4f.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 -1 2>&1 | grep -v Warning | tee $O/probe_c4.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so timeout 1200 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py -q -m gpu -k "wave_mates or vote" 2>&1 | tail -4 | tee $O/tests_c4.txt
