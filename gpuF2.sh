cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for e in DartHumanWalker-v1 DartWalker3d-v1; do python bench.py --env-id $e --envs 16384 --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'])"; done
