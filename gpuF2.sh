cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --all-bodies-collide --steps 500 --warmup 100 > gpurun_out/bench_hopper_allcaps.json 2>/dev/null; tail -c 900 gpurun_out/bench_hopper_allcaps.json | head -c 600
