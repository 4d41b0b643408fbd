cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x -k "spd" 2>&1 | tail -8
python bench.py --env-id DartWalker3dSPD-v1 --envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
