"""DartHopper-v1 as a single-env object with the reference's class name and method surface
(reference gym/envs/dart/hopper.py:6-87).  The arithmetic of ``step`` -- clamp (:25-30), tau scaling (:31-32),
frame_skip world steps (:34), reward (:52-58), done (:60-62), observation (:67-74) -- runs in the HIP kernel."""
import numpy as np

from .dart_env import BatchedDartEnv


class _SingleEnv(BatchedDartEnv):
    """num_envs == 1 facade: un-batched arguments / return values like a reference env."""
    ENV_ID = None
    _unbatched = True   # robot_skeleton getters return un-batched arrays, like pydart2's

    def __init__(self, device=0, precision=64, stepper_factory=None):
        super().__init__(self.ENV_ID, num_envs=1, device=device, precision=precision, noise="mt19937-host",
                         max_episode_steps=0, stepper_factory=stepper_factory)
        self.control_bounds = np.array([[self.task.act_high] * self.act_dim, [self.task.act_low] * self.act_dim])
        self.action_scale = np.array([self.card.act_scale[k] for k in range(self.act_dim)])

    def seed(self, seed=None):
        super().seed(None if seed is None else [seed])
        if seed is not None:
            self._rng(0)
        return [self._seeds[0] if self._seeds[0] is not None else seed]

    def reset(self):
        self._last_obs = super().reset(None)[0].astype(np.float64)
        return self._last_obs.copy()

    def _info(self, done):
        return {}

    def step(self, a):
        obs, rew, done, _ = super().step(np.asarray(a, dtype=np.float32).reshape(1, self.act_dim))
        self._last_obs = obs[0].astype(np.float64)
        return self._last_obs.copy(), np.float64(rew[0]), bool(done[0]), self._info(bool(done[0]))

    def _get_obs(self):
        """The observation of the current state as the device computed it (hopper.py:67-74)."""
        return self._last_obs.copy()

    def set_state(self, qpos, qvel):
        assert np.shape(qpos) == (self.ndofs,) and np.shape(qvel) == (self.ndofs,)  # dart_env.py:146
        super().set_state(qpos, qvel)

    def state_vector(self):
        return super().state_vector()[0]


class DartHopperEnv(_SingleEnv):
    ENV_ID = "DartHopper-v1"
