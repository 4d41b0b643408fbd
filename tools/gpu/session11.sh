# round 5, session 11: half-cheetah kernels without the big register tier in BOTH precisions (c4nt) against c4 (tier kept, vote 64)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s11; mkdir -p $O
cd $R
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 64 2>&1 | grep -v Warning | grep Dart | tee $O/probe.txt
DART_STEPPER_LIB=$R/abtest/lib_c4nt.so python tools/gpu/cheetah_coop4_probe.py c4nt -1 2>&1 | grep -v Warning | grep Dart | tee -a $O/probe.txt
DART_STEPPER_LIB=$R/abtest/lib_c4nt.so timeout 1500 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py tests/test_gpu_long_parity.py tests/test_gpu_first_launch.py tests/test_gpu_parity.py tests/test_gpu_generic.py -q -m gpu -k "cheetah or Cheetah or wave_mates or floor or fallback" 2>&1 | tail -6 | tee $O/tests_c4nt.txt
