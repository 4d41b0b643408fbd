# round 5, session 7: where the half cheetah's cycles go with the four-env wave solver as the fp64 default (timing build), the default's timing,
# and the cheetah tests (oracle parity, repeatability, batch independence) against the new library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s7; mkdir -p $O
cd $R
bash tools/gpu/wave_timing.sh c4t DartHalfCheetah-v1 64 2>&1 | grep -v Warning | tee $O/timing_c4t.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 -1 0 64 2>&1 | grep -v Warning | tee $O/probe_c4.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so timeout 1200 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py tests/test_gpu_first_launch.py tests/test_gpu_bench_dist.py tests/test_gpu_long_parity.py -q -m gpu -k "cheetah or Cheetah or fallback or floor or vote or wave_mates or batch_independence" 2>&1 | tail -8 | tee $O/tests_c4.txt
