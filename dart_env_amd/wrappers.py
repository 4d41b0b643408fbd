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
