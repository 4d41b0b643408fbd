# round 5, session 10: the fp64 half-cheetah kernel WITHOUT the big register tier (c4nt: -DDART_CHEETAH_TIER1_F64=0, kernel scratch 6.3 -> 2.5 KB per lane)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s10; mkdir -p $O
cd $R
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 -1 2>&1 | grep -v Warning | grep f64 | tee $O/probe.txt
DART_STEPPER_LIB=$R/abtest/lib_c4nt.so python tools/gpu/cheetah_coop4_probe.py c4nt -1 2>&1 | grep -v Warning | grep f64 | tee -a $O/probe.txt
DART_STEPPER_LIB=$R/abtest/lib_c4nt.so timeout 1200 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py tests/test_gpu_long_parity.py -q -m gpu -k "cheetah or Cheetah" 2>&1 | tail -6 | tee $O/tests_c4nt.txt
