cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --env-id DartWalker3d-v1 --envs 16384 --steps 200 --warmup 50 > gpurun_out/bench_w3.json 2>gpurun_out/bench_w3.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_w3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['rms_state_err'])"
python bench.py --env-id DartHumanWalker-v1 --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
