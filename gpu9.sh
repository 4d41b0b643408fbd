cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for e in DartHopper-v1 DartWalker2d-v1; do
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --env-id $e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', '%.3e steps/s'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --precision 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hopper f64', '%.3e steps/s'%d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
