"""``gym.vector``-shaped batched env (reference gym/vector/vector_env.py:8-143, sync_vector_env.py:50-84).

Observable contract kept from SyncVectorEnv: ``reset()`` -> (N, obs) float32; ``step(actions)`` -> (obs float32,
rewards float64, dones bool_, infos list-of-dict); a done env is reset inside ``step`` and the returned observation is
the POST-reset one (sync_vector_env.py:77-78); ``seed(int s)`` seeds env i with ``s + i`` (:50-58); TimeLimit sets
``info['TimeLimit.truncated']`` (wrappers/time_limit.py:18-20).  Misuse of the async pair raises the same-named errors
(gym/error.py:143-159).  What differs: there are no per-env Python objects or processes -- one HIP launch steps all N.
"""
import numpy as np

from . import spaces
from .envs.dart_env import BatchedDartEnv
from .model_card import TASKS
from .stepper import AlreadyPendingCallError, NoAsyncCallError, StepperError  # noqa: F401  (re-exported)
from . import stepper as _st


class ClosedEnvironmentError(RuntimeError):
    """Mirrors gym.error.ClosedEnvironmentError (reference gym/error.py:161-167)."""


class InfoList:
    """list-of-dict view built lazily (65 536 dict allocations per step would dominate the step)."""

    def __init__(self, truncated, extra=None):
        self._t = truncated
        self._extra = extra or {}     # env index -> dict merged into that env's info (e.g. {'episode': {...}})

    def __len__(self):
        return len(self._t)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        d = {"TimeLimit.truncated": True} if self._t[i] else {}
        if i in self._extra:
            d.update(self._extra[i])
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class DartVectorEnv:
    def __init__(self, env_id, num_envs, device=0, precision=64, noise="mt19937", copy=True, stepper_factory=None,
                 env_offset=0, all_bodies_collide=None, generic_kernel=False):
        """all_bodies_collide: None = the task's default card -- every collision shape of the robot against the floor, as DART
        has it; False restricts the contacts to the feet (BASELINE.json config[1]); DESIGN.md section 2."""
        from .model_card import card_for
        task = TASKS[env_id]
        self.env = BatchedDartEnv(env_id, num_envs, device, precision, noise,
                                  max_episode_steps=task.max_episode_steps, stepper_factory=stepper_factory,
                                  card=card_for(env_id, all_bodies_collide=all_bodies_collide, generic_kernel=generic_kernel)
                                  if (all_bodies_collide is not None or generic_kernel) else None)
        self.num_envs = num_envs
        self.precision = precision   # 64 = the product default (meets the north star's tolerance), 32 = the fast mode
        self.copy = copy
        self.single_observation_space = self.env.observation_space
        self.single_action_space = self.env.action_space
        self.observation_space = spaces.batch_space(self.single_observation_space, num_envs)
        self.action_space = spaces.Tuple((self.single_action_space,) * num_envs)
        self.closed = False
        self.viewer = None
        self.spec_id = env_id
        self._pending = False
        if self.env.device_noise:   # resets happen on the device, inside / right after the step kernel
            self.env._stepper.configure(_st.CFG_AUTORESET, 1)
            self.env._stepper.configure(_st.CFG_ENV_OFFSET, env_offset)

    def _alive(self):
        if self.closed:
            raise ClosedEnvironmentError("Trying to operate on `%s`, after a call to `close()`." % type(self).__name__)

    # ---- VectorEnv API ----
    def seed(self, seeds=None):
        self._alive()
        return self.env.seed(seeds)

    def reset_async(self):
        pass

    def reset_wait(self):
        self._alive()
        return self.env.reset(None)

    def reset(self):
        self.reset_async()
        return self.reset_wait()

    def step_async(self, actions):
        self._alive()
        if self._pending:
            raise AlreadyPendingCallError(_st.E_PENDING, "Calling `step_async` while waiting for a pending call to `step` to complete.")
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.env.act_dim)
        # copy=False hands out views of the library's one staging buffer; copy=True gets a page-locked block of its own per step
        zero_copy = not self.copy and self.env.device_noise and hasattr(self.env._stepper, "_views")
        if zero_copy:
            self.env.step_async(a, staged=True)
        else:
            self.env.step_async(a)
        self._pending = True

    def step_wait(self):
        self._alive()
        if not self._pending:
            raise NoAsyncCallError(_st.E_NOT_PENDING, "Calling `step_wait` without any prior call to `step_async`.")
        self._pending = False
        zero_copy = not self.copy and self.env.device_noise and hasattr(self.env._stepper, "_views")
        obs, rew, done, trunc = self.env.step_wait(copy=not zero_copy)
        if not self.env.device_noise and done.any():
            # post-reset observation for the done envs only (sync_vector_env.py:77-78); the other rows keep this step's
            # observation (a whole-batch refresh would lose what only the step knows, e.g. HumanWalker's foot-contact flags)
            obs = np.array(obs, copy=True)
            obs[done] = self.env.reset(done)[done]
        return obs, rew, done, InfoList(trunc)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self, **kwargs):
        if self.closed:
            return
        self.env.close()
        self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return "DartVectorEnv(%s, %d)" % (self.spec_id, self.num_envs)


def make(env_id, num_envs=1, asynchronous=True, wrappers=None, **kwargs):
    """gym.vector.make counterpart (reference gym/vector/__init__.py:12-61) for the Dart ids.  `asynchronous` only chose
    between worker processes and a serial loop in the reference; here every env of the batch advances in one kernel launch
    either way.  `wrappers` (callables applied to each single env in the reference) have no per-env object to wrap."""
    if wrappers:
        raise NotImplementedError("per-env wrappers do not apply to a batched device env; wrap the DartVectorEnv instead")
    return DartVectorEnv(env_id, num_envs, **kwargs)
