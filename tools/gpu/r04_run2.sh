# Round 4, second measurement pass (one box): tree-kernel correctness at HEAD, A/B of the round's tree-kernel changes, phase shares,
# stage-1 sweep variant of the lane kernels, host-path A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_gpu_spatial.py tests/test_gpu_long_parity.py tests/test_gpu_pgs_parity.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="base nobake walk" bash tools/gpu/r04_ab_tree.sh > $O/ab_tree.txt 2>&1; cat $O/ab_tree.txt
for p in 64 32; do PREC=$p python tools/diag_spatial_stats.py 2>/dev/null | grep "phase cycles" | sed "s/^/f$p HEAD /"; DART_STEPPER_LIB=$R/abtest/lib_walk.so PREC=$p python tools/diag_spatial_stats.py 2>/dev/null | grep "phase cycles" | sed "s/^/f$p walk+nobake /"; done > $O/phases.txt 2>&1; cat $O/phases.txt
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision 64 --steps 1500 --warmup 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f64 %.3f us (kernel %.3f)' % (1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in base s1sweep; do run $v DartHopper-v1; run $v DartWalker2d-v1; done; done > $O/ab_s1sweep.txt 2>&1; cat $O/ab_s1sweep.txt
unset DART_STEPPER_LIB
python tools/gpu/host_path_c.py > $O/host_path.txt 2>&1; cat $O/host_path.txt
