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

    def __init__(self, env_id, total_envs, rank=None, world_size=None, device=None, precision=64, seed=0,
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


class RolloutBuffer:
    """T-step trajectory of this rank's shard kept where the stepper writes it: ``dart_step_device`` stores step t's
    observation / reward / done / truncated flags straight into slot t of HBM tensors (no host copy, no staging), the
    policy reads the previous slot.  ``gather()`` is the single collective of the path ("RCCL
    only to gather rollouts"): one all_gather per tensor of the whole trajectory, shard-major.

    policy(obs_t) -> actions: a callable on torch tensors of the stepper's device; ``obs[0]`` is the reset observation.
    With an injected stepper that has no ``step_device`` (CPU tests) the buffer steps through host arrays.
    """

    def __init__(self, venv, horizon):
        import torch
        self.venv = getattr(venv, "venv", venv)          # ShardedDartVectorEnv or DartVectorEnv
        self.st = self.venv.env._stepper
        self.T, self.n = int(horizon), self.venv.num_envs
        self.obs_dim, self.act_dim = self.venv.env.obs_dim, self.venv.env.act_dim
        self.on_device = hasattr(self.st, "step_device") and torch.cuda.is_available()
        dev = torch.device("cuda", self.st.device) if self.on_device else torch.device("cpu")
        self.obs = torch.empty((self.T + 1, self.n, self.obs_dim), dtype=torch.float32, device=dev)
        self.actions = torch.empty((self.T, self.n, self.act_dim), dtype=torch.float32, device=dev)
        self.rewards = torch.empty((self.T, self.n), dtype=torch.float32, device=dev)
        self.dones = torch.empty((self.T, self.n), dtype=torch.uint8, device=dev)
        self.truncated = torch.empty((self.T, self.n), dtype=torch.uint8, device=dev)
        self._started = False
        # a stream of our own: the C ABI treats a NULL stream argument as "the handle's internal stream", which is what
        # torch's default stream would pass
        self._stream = torch.cuda.Stream(device=dev) if self.on_device else None

    def collect(self, policy):
        """Fill the buffer with T steps; continues from the last observation of the previous call (auto-reset on)."""
        import torch
        if self.on_device:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):      # the policy's kernels and the step kernels share one stream
                stream = self._stream.cuda_stream
                if not self._started:
                    self.st.reset_device(0, self.obs[0].data_ptr(), stream)
                else:
                    self.obs[0].copy_(self.obs[self.T])
                for t in range(self.T):
                    a = policy(self.obs[t])
                    self.actions[t].copy_(a)
                    self.st.step_device(self.actions[t].data_ptr(), self.obs[t + 1].data_ptr(), self.rewards[t].data_ptr(),
                                        self.dones[t].data_ptr(), self.truncated[t].data_ptr(), stream)
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            if not self._started:
                self.obs[0] = torch.from_numpy(self.venv.reset())
            else:
                self.obs[0] = self.obs[self.T].clone()
            for t in range(self.T):
                a = policy(self.obs[t])
                self.actions[t] = a
                ob, r, d, infos = self.venv.step(a.numpy())
                self.obs[t + 1] = torch.from_numpy(ob)
                self.rewards[t] = torch.from_numpy(r.astype(np.float32))
                self.dones[t] = torch.from_numpy(d.astype(np.uint8))
                self.truncated[t] = torch.tensor([1 if i.get("TimeLimit.truncated", False) else 0 for i in infos], dtype=torch.uint8)
        self._started = True
        return self

    def gather(self, force_collective=False):
        """-> dict of tensors with a leading world dimension: obs (W, T+1, n, k), actions, rewards, dones, truncated.
        force_collective: issue the all-gathers even in a one-rank group (a 1-GPU box can then execute the RCCL path)."""
        import torch
        import torch.distributed as dist
        names = ("obs", "actions", "rewards", "dones", "truncated")
        if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
            return {k: getattr(self, k).unsqueeze(0) for k in names}
        w = dist.get_world_size()
        out = {}
        for k in names:
            t = getattr(self, k).contiguous()
            if dist.get_backend() == "nccl":      # RCCL: one fused all-gather per tensor
                full = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(full, t)
            else:
                parts = [torch.empty_like(t) for _ in range(w)]
                dist.all_gather(parts, t)
                full = torch.stack(parts, dim=0)
            out[k] = full
        return out
