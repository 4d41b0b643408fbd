R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_gpu_spatial.py tests/test_gpu_pgs_parity.py "tests/test_gpu_long_parity.py::test_fp64_untrimmed_rms_below_1e_4_over_1000_steps" -x -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="base prev walk" bash tools/gpu/r04_ab_tree.sh > $O/ab_tree.txt 2>&1; cat $O/ab_tree.txt
for p in 64 32; do PREC=$p python tools/diag_spatial_stats.py 2>/dev/null | grep "phase cycles" | sed "s/^/f$p HEAD /"; done > $O/phases.txt 2>&1; cat $O/phases.txt
