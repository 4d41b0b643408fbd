cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x 2>&1 | tail -3
for e in DartHumanWalker-v1 DartWalker3d-v1 DartHalfCheetah-v1 DartSnake7Link-v1; do
python bench.py --env-id $e --envs $([ $e = DartHumanWalker-v1 -o $e = DartWalker3d-v1 ] && echo 16384 || echo 65536) --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'])"
done
