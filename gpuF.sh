cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x 2>&1 | tail -3
python bench.py --env-id DartHumanWalker-v1 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python bench.py --env-id DartWalker3d-v1 --envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python bench.py --env-id DartHalfCheetah-v1 --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
