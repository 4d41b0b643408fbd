# round 5, session 6: the four-envs-per-pass wave solver (abtest/lib_c4.so) against the shipped library -- solver harness, parity,
# repeatability, batch independence, then the timing sweep over the wave vote K
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_wave_blcp.py -q -x -m gpu 2>&1 | tail -3 > $O/harness.txt; cat $O/harness.txt
python tools/gpu/cheetah_coop4_probe.py base 0 3 2>&1 | grep -v Warning | tee $O/probe_base.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so python tools/gpu/cheetah_coop4_probe.py c4 0 3 63 2>&1 | grep -v Warning | tee $O/probe_c4.txt
DART_STEPPER_LIB=$R/abtest/lib_c4.so timeout 900 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_repeatability.py -q -m gpu -k "cheetah or Cheetah or fallback or floor or vote" 2>&1 | tail -8 | tee $O/tests_c4.txt
