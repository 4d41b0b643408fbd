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


ROLLOUT_SCRIPT = """
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import dart_env_amd.vector as V
from dart_env_amd.distributed import RolloutBuffer, ShardedDartVectorEnv
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
venv = ShardedDartVectorEnv("DartHopper-v1", 1024, seed=3)
buf = RolloutBuffer(venv, 16)
policy = lambda ob: torch.tanh(ob[:, :3] * 3.0 - ob[:, 5:8])
buf.collect(policy)
g = buf.gather(force_collective=True)          # the all_gather_into_tensor calls run over RCCL even with one rank
torch.cuda.synchronize()
ok = all(torch.equal(g[k][0], getattr(buf, k)) for k in ("obs", "actions", "rewards", "dones", "truncated"))
ob, r, d, infos = venv.step(buf.actions[0].cpu().numpy())
fo, fr, fd = venv.gather_last_step(force_collective=True)     # resident path (opt-in): the all-gather reads the step's outputs where they sit in HBM
import numpy as np
host_ok = bool(venv._resident_outputs() is not None and np.array_equal(fo, ob) and np.array_equal(fr, r) and np.array_equal(fd, d))
# the default honours its ARGUMENTS under RCCL as it does under gloo: transformed arrays are what comes back, not the device block
to, tr, td = venv.gather_rollout(ob * 0.5, np.clip(r, -0.25, 0.25), ~d, force_collective=True)
host_ok = host_ok and bool(np.array_equal(to, ob * 0.5) and np.allclose(tr, np.clip(r, -0.25, 0.25).astype(np.float32)) and np.array_equal(td, ~d))
try:
    venv.gather_rollout(ob.copy(), r, d, force_collective=True, resident=True)      # not the step's own arrays: refused
    host_ok = False
except ValueError:
    pass
# the device-resident form: step in HBM, one packed uint8 all-gather over RCCL, nothing touches the host
dob, drew, ddone, dtr = venv.step_device(buf.actions[1])
gob, grew, gdone = venv.gather_rollout_device(force_collective=True)
try:
    venv.gather_last_step(force_collective=True)      # a step_device() ran since: the block no longer holds a host step
    host_ok = False
except ValueError:
    pass
torch.cuda.synchronize()
dev_ok = bool(gob.is_cuda and gob.dtype == torch.float32 and gdone.dtype == torch.uint8 and torch.equal(gob, dob) and torch.equal(grew, drew)
              and torch.equal(gdone, ddone))
print(json.dumps({"ok": bool(ok), "shape": list(g["obs"].shape), "backend": dist.get_backend(), "dones": int(buf.dones.sum().item()),
                  "gather_rollout_rows": int(fo.shape[0]), "gather_rollout_resident_ok": host_ok, "device_gather_ok": dev_ok, "device_gather_rows": int(gob.shape[0])}))
venv.close()
dist.destroy_process_group()
"""


def test_rollout_buffer_gather_runs_over_rccl():
    """RolloutBuffer.gather() -- the north star's "RCCL ... only to gather rollouts" -- executed on the device through RCCL's
    all_gather_into_tensor with the one rank a 1-GPU box has; the 2-rank arithmetic is covered over gloo (tests/test_distributed_gloo.py)."""
    import tempfile
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(ROLLOUT_SCRIPT % ROOT)
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", path]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["backend"] == "nccl" and d["shape"] == [1, 17, 1024, 11] and d["dones"] > 0 and d["gather_rollout_rows"] == 1024
    assert d["device_gather_ok"] and d["device_gather_rows"] == 1024
    assert d["gather_rollout_resident_ok"]      # gather_rollout() took the shard from HBM (dart_device_outputs) and returned the step's own values
