cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --env-id DartHumanWalker-v1 > gpurun_out/bench_hw.json 2> gpurun_out/bench_hw.err; tail -c 1500 gpurun_out/bench_hw.json
python bench.py --env-id DartWalker3d-v1 --envs 16384 > gpurun_out/bench_w3.json 2> gpurun_out/bench_w3.err; tail -c 600 gpurun_out/bench_w3.json
