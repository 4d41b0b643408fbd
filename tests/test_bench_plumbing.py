"""bench.py's N > 1 plumbing on CPU: rank / env_offset sharding, max-over-ranks timing, JSON assembly, the rollout gather and
the refusal of a WORLD_SIZE that contradicts --gpus.  The GPU shard is replaced by a numpy stand-in on the planar kernel
emulator (tests/emu_lib.py) injected through bench.main(env_factory=...), the collective backend is gloo."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class EmuBenchEnv:
    """stand-in for bench.HipBenchEnv: same surface, host arrays, device kernels compiled for the host"""

    def __init__(self, env_id, n, local_rank, precision, env_offset, ring=16, ring_seed=1234, all_bodies_collide=None, configure=()):
        from dart_env_amd import stepper as st
        from dart_env_amd.model_card import card_for
        from tests.emu_lib import EmuStepper
        self.card = card_for(env_id)
        self.n, self.ring_len, self.env_offset = n, ring, env_offset
        self.env = EmuStepper(self.card, n, precision=precision)
        self.env.configure(st.CFG_AUTORESET, 1); self.env.configure(st.CFG_SEED, 0); self.env.configure(st.CFG_ENV_OFFSET, env_offset)
        self.ring = np.random.RandomState(ring_seed).uniform(-1, 1, (ring, n, self.card.act_dim)).astype(np.float32)
        self.last = None

    def reset(self):
        self.env.reset()

    def run(self, k, base=0):
        for i in range(k):
            self.last = self.env.step(self.ring[(base + i) % self.ring_len])

    def timed_steps(self, k):
        t0 = time.perf_counter()
        self.run(k)
        return (time.perf_counter() - t0) / k * 1e3

    def mark(self, which):
        if which == 0:
            self._t0 = time.perf_counter()
        else:
            self._t1 = time.perf_counter()

    def elapsed_ms(self):
        return (self._t1 - self._t0) * 1e3

    def sync(self):
        pass

    def done_fraction(self):
        return float(self.last[2].mean())

    def packed_last(self):
        from dart_env_amd.distributed import ShardedDartVectorEnv
        o, r, d, _ = self.last
        return ShardedDartVectorEnv._pack(torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32)), torch.from_numpy(r.astype(np.float32)),
                                          torch.from_numpy(d.astype(np.uint8)))

    def is_static(self):
        return self.env.is_static

    def close(self):
        self.env.close()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from tests.test_bench_plumbing import EmuBenchEnv as E
    res = bench.main(["--gpus", str(world), "--envs", "64", "--steps", "3", "--warmup", "1"], env_factory=E, dist_backend="gloo")
    q.put((rank, res))


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_bench_line(world):
    """2 ranks, and the 8 ranks of BASELINE config 5's node: every rank a process, gloo instead of RCCL, the GPU shard replaced by the
    kernel emulator -- sharding by rank, barriers, max-over-ranks timing, per-rank clocks, the packed rollout all-gather."""
    port = 33500 + (os.getpid() % 2000) + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=600) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[1] is None                       # only rank 0 reports
    r = got[0]
    json.dumps(r)
    assert r["n_gpus"] == world and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["env_offsets"] == [64 * g for g in range(world)]   # disjoint contiguous shards keyed by rank
    assert r["config"]["envs_per_gpu"] == 64 and "DartHopper-v1" in r["config"]["workload"]
    assert abs(r["value"] - world * 64 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]   # whole-job aggregate over the slowest rank
    assert r["gather_ms"] > 0 and "gather_note" not in r
    assert r["gather_bytes_per_rank"] == 64 * (11 * 4 + 4 + 1)         # obs f32 | reward f32 | done u8: flags travel as bytes
    assert [pr["rank"] for pr in r["per_rank"]] == list(range(world))              # every rank's own clock is in the line ...
    assert max(pr["wall_ms_per_step"] for pr in r["per_rank"]) == pytest.approx(r["ms_per_step"])   # ... and `value` uses the slowest
    assert r["roofline"]["algorithmic_bytes_per_env_step"] == 157 and r["vs_baseline"] is None
    assert "cpu_baseline" not in r and "other_configs" not in r      # N = 1 extras only


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks_when_no_launcher_did(world):
    """`python bench.py --gpus N` with no RANK in the environment (the way the driver invokes --gpus 1) starts its own N
    ranks, one JSON line comes out of rank 0 and carries the world size the process group itself reports."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--envs", "64", "--steps", "3", "--warmup", "1",
                        "--dist-backend", "gloo", "--env-factory", "tests.test_bench_plumbing:EmuBenchEnv"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                                    # ONE line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == world and r["nccl_world_size"] == world and r["dist_backend"] == "gloo" and r["launcher"] == "self"
    assert r["config"]["env_offsets"] == [64 * g for g in range(world)]
    assert [pr["rank"] for pr in r["per_rank"]] == list(range(world)) and r["gather_ms"] > 0
    assert abs(r["value"] - world * 64 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


def test_self_launch_reports_a_dead_rank():
    """a rank that fails takes the launch down with a non-zero exit instead of leaving the others in a barrier"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "64", "--steps", "1", "--warmup", "0",
                        "--dist-backend", "gloo", "--env-factory", "tests.test_bench_plumbing:NoSuchFactory"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "self-launched rank" in p.stderr


def test_world_size_mismatch_is_fatal(monkeypatch):
    """under a launcher (RANK set) a WORLD_SIZE that contradicts --gpus is an error, not a silent re-launch"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"], env_factory=EmuBenchEnv, dist_backend="gloo")
    assert "WORLD_SIZE" in str(e.value)


@pytest.mark.parametrize("var", ["SLURM_PROCID", "OMPI_COMM_WORLD_RANK", "LOCAL_RANK"])
def test_self_launch_refuses_under_a_foreign_launcher(monkeypatch, var):
    """srun / mpirun set their own rank variables, not RANK -- N processes each starting N more would be N^2 on the GPUs"""
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SLURM_PROCID", "OMPI_COMM_WORLD_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv(var, "3")
    monkeypatch.setattr(bench, "self_launch", lambda *a, **k: pytest.fail("started ranks under a foreign launcher"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"], env_factory=EmuBenchEnv, dist_backend="gloo")
    assert var in str(e.value)


def test_shards_are_contiguous_and_disjoint():
    sys.path.insert(0, ROOT)
    import bench
    spans = [bench.shard_of(r, 65536) for r in range(8)]
    assert spans[0] == (0, 65536) and spans[7] == (7 * 65536, 65536)
    assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(7))
    assert bench.default_envs("DartHumanWalker-v1") == 16384 and bench.default_envs("DartHopper-v1") == 65536


def test_counter_files_are_used_only_while_their_kernel_sources_are_unchanged(tmp_path, monkeypatch):
    """roofline.traffic / valu_issue / valu are COPIED from committed counter files; every entry is stamped with the
    hash of the kernel sources it was measured on (tools/source_hash.py) and bench.py uses it only while the tree still hashes to that --
    otherwise the line says `stale` and carries no number."""
    sys.path.insert(0, ROOT)
    import bench
    from tools.source_hash import all_hashes, family_hash, family_files
    h = all_hashes()
    assert set(h) == {"planar", "spatial", "dart_stepper"} and all(len(v) == 16 for v in h.values())
    assert any(f.endswith("planar_kernel.hpp") for f in family_files("planar")) and any(f.endswith("spatial_kernel.hpp") for f in family_files("spatial"))
    prof = tmp_path / "profiles"
    prof.mkdir()
    good = {"bytes_per_launch": 123, "kernel": "dartk::step_kernel<double, ...>", "kernel_family": "planar", "source_hash": h["planar"],
            "source": "test", "valu_issue": {"x": 1}}
    stale = dict(good, source_hash="0123456789abcdef")
    unstamped = {k: v for k, v in good.items() if k not in ("source_hash", "kernel_family")}
    (prof / "pmc_traffic.json").write_text(json.dumps({"a": good, "b": stale, "c": unstamped}))
    (prof / "flops_per_env_step.json").write_text(json.dumps({
        "DartHopper-v1": {"flops_per_env_step": 1000.0, "kernel_family": "planar", "source_hash": h["planar"], "method": "m", "sample": "s"},
        "DartWalker2d-v1": {"flops_per_env_step": 1000.0, "kernel_family": "planar", "source_hash": "deadbeefdeadbeef"}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r = {"traffic": None}
    bench.attach_pmc(r, "a")
    assert r["traffic"] == 123 and r["stale"] is False and r["valu_issue"] == {"x": 1} and r["kernel_source_hash"]["this_tree"] == h["planar"]
    for key in ("b", "c"):
        r = {"traffic": None}
        bench.attach_pmc(r, key)
        assert r["traffic"] is None and r["stale"] is True and "valu_issue" not in r and "not used" in r["stale_note"]
    r = {}
    bench.attach_valu(r, "DartHopper-v1", 65536, "f64", 0.032)
    assert r["valu"]["flops_per_env_step"] == 1000.0 and 0 < r["valu"]["frac"] < 1
    r = {}
    bench.attach_valu(r, "DartWalker2d-v1", 65536, "f64", 0.12)
    assert r["valu"] == {"stale": True, "stale_note": r["valu"]["stale_note"]} and "frac" not in r["valu"]
    # the committed files themselves are stamped, family by family
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert pmc and all("source_hash" in e and e["kernel_family"] in ("planar", "spatial") for e in pmc.values())
