"""Episode bookkeeping around the single-env objects: the step budget (observable behaviour of reference
gym/wrappers/time_limit.py:5-25) and per-episode return / length records (gym/wrappers/record_episode_statistics.py:7-34).
The vector env keeps both on the device instead (csrc/planar_kernel.hpp, csrc/episode_kernels.hpp)."""
import time as _time
from collections import deque as _deque


class _EnvShell:
    """attribute pass-through to the wrapped env, shared by the two wrappers below"""

    def __init__(self, env):
        self.env = env
        self.action_space, self.observation_space = env.action_space, env.observation_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)


class TimeLimit(_EnvShell):
    """Ends an episode after `max_episode_steps` steps: the step that exhausts the budget reports done, and
    info["TimeLimit.truncated"] says whether the task itself had NOT ended there."""

    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None          # None until the first reset: stepping before that is a usage error

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def step(self, action):
        if self._elapsed_steps is None:
            raise AssertionError("Cannot call env.step() before calling reset()")
        out = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps < self._max_episode_steps:
            return out
        ob, reward, task_done, info = out
        info["TimeLimit.truncated"] = not task_done
        return ob, reward, True, info


class RecordEpisodeStatistics(_EnvShell):
    """Adds info['episode'] = {r: return, l: length, t: seconds since construction} to the step that ends an episode and keeps
    the last `deque_size` returns / lengths in return_queue / length_queue."""

    def __init__(self, env, deque_size=100):
        super().__init__(env)
        self.t0 = _time.time()
        self.return_queue, self.length_queue = _deque(maxlen=deque_size), _deque(maxlen=deque_size)
        self._clear()

    def _clear(self):
        self.episode_return, self.episode_length = 0.0, 0

    def reset(self, **kwargs):
        ob = self.env.reset(**kwargs)
        self._clear()
        return ob

    def step(self, action):
        ob, reward, done, info = self.env.step(action)
        self.episode_return += reward
        self.episode_length += 1
        if done:
            ret, length = self.episode_return, self.episode_length
            self.return_queue.append(ret)
            self.length_queue.append(length)
            info["episode"] = {"r": ret, "l": length, "t": round(_time.time() - self.t0, 6)}
            self._clear()
        return ob, reward, done, info


class VectorRecordEpisodeStatistics:
    """The same statistics for a DartVectorEnv.  The per-env return / length accumulators live on the device
    (DART_CFG_EPISODE_STATS, csrc/episode_kernels.hpp); the host only fetches the latched values of the envs that
    finished in this step.  With an injected stepper that has no device accumulators it accumulates on the host."""

    def __init__(self, venv, deque_size=100):
        import time
        from collections import deque
        from . import stepper as _st
        self.venv = venv
        self.num_envs = venv.num_envs
        self._time = time
        self.t0 = time.time()
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        st = venv.env._stepper
        self._device = hasattr(st, "episode_stats")
        if self._device:
            st.configure(_st.CFG_EPISODE_STATS, 1)
        else:
            import numpy as np
            self._ret = np.zeros(self.num_envs)
            self._len = np.zeros(self.num_envs, dtype=np.int64)

    def __getattr__(self, name):
        return getattr(self.venv, name)

    def reset(self):
        obs = self.venv.reset()
        if not self._device:
            self._ret[:] = 0.0
            self._len[:] = 0
        return obs

    def totals(self, clear=False):
        """(sum of returns, sum of lengths, finished episodes) since the last clear -- device path only."""
        return tuple(self.venv.env._stepper.episode_stats(per_env=False, clear_totals=clear)[2])

    def step(self, actions):
        from .vector import InfoList
        obs, rew, done, infos = self.venv.step(actions)
        extra = {}
        if self._device:
            if done.any():
                r, l, _ = self.venv.env._stepper.episode_stats()
        else:
            self._ret += rew
            self._len += 1
            r, l = self._ret.copy(), self._len.copy()
            self._ret[done] = 0.0
            self._len[done] = 0
        if done.any():
            t = round(self._time.time() - self.t0, 6)
            for i in done.nonzero()[0]:
                extra[int(i)] = {'episode': {'r': float(r[i]), 'l': int(l[i]), 't': t}}
                self.return_queue.append(float(r[i]))
                self.length_queue.append(int(l[i]))
        return obs, rew, done, InfoList(infos._t, extra)
