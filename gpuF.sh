cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x 2>&1 | tail -5
PREC=32 python tools/diag_spatial_stats.py 1e-4 2>&1 | grep -v amdgpu | head -1
python bench.py --env-id DartHumanWalker-v1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python bench.py --env-id DartWalker3d-v1 --envs 16384 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
