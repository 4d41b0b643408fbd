"""bench.py's multi-rank code path on real RCCL: process group on the GPU, barriers, max-over-ranks all_reduce, the env_offset
all_gather and the rollout all_gather_into_tensor -- with the one rank a 1-GPU box offers (`--force-dist`).  The N > 1 sharding
arithmetic itself is covered on CPU (tests/test_bench_plumbing.py, gloo, 2 ranks)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_its_collective_path_over_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--envs", "4096",
           "--force-dist", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["env_offsets"] == [0] and d["value"] > 0
    assert "gather_ms" in d and "gather_note" not in d, d.get("gather_note")
