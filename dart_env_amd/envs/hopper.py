"""DartHopper-v1 as a single-env object with the reference's class name and method surface
(reference gym/envs/dart/hopper.py:6-87).  The arithmetic of ``step`` -- clamp (:25-30), tau scaling (:31-32),
frame_skip world steps (:34), reward (:52-58), done (:60-62), observation (:67-74) -- runs in the HIP kernel."""
import numpy as np

from .dart_env import BatchedDartEnv


class _SingleEnv(BatchedDartEnv):
    """num_envs == 1 facade: un-batched arguments / return values like a reference env."""
    ENV_ID = None

    def __init__(self, device=0, precision=32, stepper_factory=None):
        super().__init__(self.ENV_ID, num_envs=1, device=device, precision=precision, noise="mt19937",
                         max_episode_steps=0, stepper_factory=stepper_factory)
        self.control_bounds = np.array([[self.card.act_high[k] for k in range(self.act_dim)],
                                        [self.card.act_low[k] for k in range(self.act_dim)]])
        self.action_scale = np.array([self.card.act_scale[k] for k in range(self.act_dim)])

    def seed(self, seed=None):
        super().seed(None if seed is None else [seed])
        if seed is not None:
            self._rng(0)
        return [self._seeds[0] if self._seeds[0] is not None else seed]

    def reset(self):
        return super().reset(None)[0].astype(np.float64)

    def step(self, a):
        obs, rew, done, _ = super().step(np.asarray(a, dtype=np.float32).reshape(1, self.act_dim))
        return obs[0].astype(np.float64), np.float64(rew[0]), bool(done[0]), {}

    def _get_obs(self):
        q, dq = self._stepper.get_state()
        state = np.concatenate([q[0, 1:], np.clip(dq[0], -self.card.obs_vel_clip, self.card.obs_vel_clip)])
        state[0] = q[0, 1] + self._root_height0
        return state

    @property
    def _root_height0(self):
        # height of bodynodes[2].com() at q = 0 (pelvis frame origin; COM offset 0 for both planar models)
        from ..model_card import load_model
        m = load_model(self.task.model)
        return float(m.bodies[0].T_pj[1, 3] + m.bodies[2].com[1])

    def set_state(self, qpos, qvel):
        assert np.shape(qpos) == (self.ndofs,) and np.shape(qvel) == (self.ndofs,)  # dart_env.py:146
        super().set_state(qpos, qvel)

    def state_vector(self):
        return super().state_vector()[0]


class DartHopperEnv(_SingleEnv):
    ENV_ID = "DartHopper-v1"
