"""N>1 path on CPU: two gloo ranks each own a shard; the union must equal one unsharded batch, and the rollout
gather must return the full batch in global env order on every rank."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dart_env_amd.distributed import ShardedDartVectorEnv
    from tests.fake_stepper import OracleStepper
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    env = ShardedDartVectorEnv("DartHopper-v1", total, rank=rank, world_size=world, seed=5, stepper_factory=OracleStepper)
    acts = np.random.RandomState(0).uniform(-1, 1, (steps, total, 3)).astype(np.float32)
    env.reset()
    out = None
    for t in range(steps):
        obs, rew, done, _ = env.step(acts[t, env.offset:env.offset + env.count])
        out = env.gather_rollout(obs, rew, done)
    # resident=True is an opt-in for the last step's OWN arrays (round 6): the same result here (gloo has no device block to read),
    # a ValueError for any other arrays; the default gathers whatever it is given
    res = env.gather_last_step()
    assert all(np.array_equal(a, b) for a, b in zip(res, out))
    try:
        env.gather_rollout(obs.copy(), rew, done, resident=True)
        raise AssertionError("resident=True accepted arrays that are not the last step's")
    except ValueError:
        pass
    half = env.gather_rollout(obs * 0.5, rew, done)
    assert np.array_equal(half[0], out[0] * 0.5)
    # the tensor form of the same exchange (on a GPU: outputs that never left HBM; here CPU tensors over gloo): one packed uint8 block
    import torch
    to, tr, td = env.gather_rollout_device(torch.from_numpy(obs), torch.from_numpy(rew.astype(np.float32)), torch.from_numpy(done.astype(np.uint8)))
    assert td.dtype == torch.uint8 and np.array_equal(to.numpy(), out[0]) and np.array_equal(td.numpy() != 0, out[2])
    assert np.array_equal(tr.numpy(), out[1].astype(np.float32))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    env.close()


def test_two_rank_sharding_equals_single_batch():
    from dart_env_amd.distributed import ShardedDartVectorEnv
    from tests.fake_stepper import OracleStepper
    total, steps, world = 8, 12, 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process reference: same seed, all envs in one batch
    env = ShardedDartVectorEnv("DartHopper-v1", total, rank=0, world_size=1, seed=5, stepper_factory=OracleStepper)
    acts = np.random.RandomState(0).uniform(-1, 1, (steps, total, 3)).astype(np.float32)
    env.reset()
    for t in range(steps):
        obs, rew, done, _ = env.step(acts[t])
    for r in range(world):
        o, w, d = got[r]
        assert o.shape == (total, 11)
        assert np.array_equal(o, obs) and np.allclose(w, rew, atol=1e-5) and np.array_equal(d, done)


def _rollout_worker(rank, world, port, total, T, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dart_env_amd.distributed import RolloutBuffer, ShardedDartVectorEnv
    from tests.fake_stepper import OracleStepper
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    env = ShardedDartVectorEnv("DartHopper-v1", total, rank=rank, world_size=world, seed=9, stepper_factory=OracleStepper)
    buf = RolloutBuffer(env, T)
    policy = lambda ob: torch.tanh(ob[:, :3] * 3.0 - ob[:, 5:8])   # deterministic function of the observation
    full = buf.collect(policy).gather()
    q.put((rank, {k: v.numpy() for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()
    env.close()


def test_rollout_buffer_gather_two_ranks():
    """Trajectories collected shard-by-shard and gathered once equal the trajectory of the unsharded batch."""
    import torch
    from dart_env_amd.distributed import RolloutBuffer, ShardedDartVectorEnv
    from tests.fake_stepper import OracleStepper
    total, T, world = 6, 25, 2
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rollout_worker, args=(r, world, port, total, T, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    env = ShardedDartVectorEnv("DartHopper-v1", total, rank=0, world_size=1, seed=9, stepper_factory=OracleStepper)
    policy = lambda ob: torch.tanh(ob[:, :3] * 3.0 - ob[:, 5:8])
    ref = RolloutBuffer(env, T).collect(policy).gather()
    n = total // world
    for r in range(world):
        g = got[r]
        assert g["obs"].shape == (world, T + 1, n, 11) and g["dones"].shape == (world, T, n)
        for k in ("obs", "actions", "rewards", "dones", "truncated"):
            whole = ref[k].numpy()[0]                                  # (T[+1], total, ...)
            for s in range(world):
                assert np.allclose(g[k][s], whole[:, s * n:(s + 1) * n], atol=1e-6), (k, r, s)
    assert ref["dones"].sum() > 0
    env.close()
