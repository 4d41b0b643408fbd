"""Env-level data parallelism across the GPUs of one node: one process per GPU, contiguous env shards.

The reference's only parallelism is one OS process per env with pipes + shared memory
(reference gym/vector/async_vector_env.py:105-117).  Environments never interact (hopper.py / dart_env.py have no
cross-env term), so here rank g simply owns envs [g*n, (g+1)*n): no collective is needed to *step*.  The one exchange
the path has is handing the per-step rollout (obs, reward, done) to a consumer that wants the full batch --
``gather_rollout`` does that with a single all_gather (RCCL over xGMI with backend "nccl", gloo on CPU in tests).
"""
import os

import numpy as np


def shard_layout(total_envs: int, world_size: int, rank: int):
    """-> (offset, count) of the contiguous shard of `rank`; the first (total % world) ranks get one extra env."""
    base, extra = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def env_rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class ShardedDartVectorEnv:
    """This rank's shard of a `total_envs` batch; Philox reset streams are keyed by the GLOBAL env index, so the
    union of all shards is bit-identical to one big single-GPU batch."""

    def __init__(self, env_id, total_envs, rank=None, world_size=None, device=None, precision=32, seed=0,
                 stepper_factory=None):
        from .vector import DartVectorEnv
        r, lr, w = env_rank_info()
        self.rank = r if rank is None else rank
        self.world_size = w if world_size is None else world_size
        self.offset, self.count = shard_layout(total_envs, self.world_size, self.rank)
        self.total_envs = total_envs
        dev = lr if device is None else device
        self.venv = DartVectorEnv(env_id, self.count, device=dev, precision=precision, noise="philox",
                                  stepper_factory=stepper_factory, env_offset=self.offset)
        self.venv.seed([seed] * self.count)   # one Philox key for the whole job; streams differ by global index
        self.num_envs = self.count

    def reset(self):
        return self.venv.reset()

    def step(self, actions):
        return self.venv.step(actions)

    def close(self):
        self.venv.close()

    def gather_rollout(self, obs, reward, done):
        """all_gather (obs f32 (n,k), reward f64 (n,), done bool (n,)) -> full-batch arrays in global env order.
        Requires equal shard sizes (total_envs % world_size == 0) like the 8 x 65 536 config."""
        import torch
        import torch.distributed as dist
        if self.world_size == 1 or not dist.is_initialized():
            return obs, reward, done
        assert self.total_envs % self.world_size == 0, "gather_rollout needs equal shards"
        k = obs.shape[1]
        packed = np.concatenate([obs.astype(np.float32), reward.astype(np.float32)[:, None],
                                 done.astype(np.float32)[:, None]], axis=1)
        t = torch.from_numpy(packed)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        if dist.get_backend() == "nccl":   # one fused RCCL all-gather into a contiguous (world, n, k+2) buffer
            out = torch.empty((self.world_size * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous())
        else:
            parts = [torch.empty_like(t) for _ in range(self.world_size)]
            dist.all_gather(parts, t)
            out = torch.cat(parts, dim=0)
        full = out.reshape(-1, k + 2).cpu().numpy()
        return full[:, :k], full[:, k].astype(np.float64), full[:, k + 1] > 0.5
