"""TimeLimit for the single-env objects, same observable behaviour as reference gym/wrappers/time_limit.py:5-25
(the vector env applies the limit on the device instead)."""


class TimeLimit:
    def __init__(self, env, max_episode_steps=None):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class RecordEpisodeStatistics:
    """Single-env wrapper, same behaviour as reference gym/wrappers/record_episode_statistics.py:7-34."""

    def __init__(self, env, deque_size=100):
        import time
        from collections import deque
        self.env = env
        self.action_space, self.observation_space = env.action_space, env.observation_space
        self._time = time
        self.t0 = time.time()
        self.episode_return = 0.0
        self.episode_length = 0
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kwargs):
        observation = self.env.reset(**kwargs)
        self.episode_return = 0.0
        self.episode_length = 0
        return observation

    def step(self, action):
        observation, reward, done, info = self.env.step(action)
        self.episode_return += reward
        self.episode_length += 1
        if done:
            info['episode'] = {'r': self.episode_return, 'l': self.episode_length,
                               't': round(self._time.time() - self.t0, 6)}
            self.return_queue.append(self.episode_return)
            self.length_queue.append(self.episode_length)
            self.episode_return = 0.0
            self.episode_length = 0
        return observation, reward, done, info


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
