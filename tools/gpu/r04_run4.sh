R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e; mkdir -p $O
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="base prev2 f32w2" bash tools/gpu/r04_ab_tree.sh > $O/ab_tree.txt 2>&1; cat $O/ab_tree.txt
for p in 64 32; do PREC=$p python tools/diag_spatial_stats.py 2>/dev/null | grep -E "phase cycles|iters hist" | sed "s/^/f$p HEAD /"; done > $O/phases.txt 2>&1; cut -c1-400 $O/phases.txt
