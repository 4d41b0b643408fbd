"""A second user-defined task on the batched DartEnv base class (compare reference gym/envs/dart/hopper.py), on a model whose TREE
matches a compiled lane-kernel topology (tests/golden/assets/pogo.skel: floating planar root + three-link chain, four capsules):
the library then runs it one env per GPU lane instead of on the tree kernel -- TEST CODE shared by the CPU and GPU tests."""
import os

import numpy as np

from dart_env_amd.envs.dart_env import DartEnv

SKEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets", "pogo.skel")
SCALE = np.array([40.0, 30.0, 15.0])


class PogoEnv(DartEnv):
    def __init__(self, num_envs=1, **kw):
        self.control_bounds = np.array([[1.0] * 3, [-1.0] * 3])
        DartEnv.__init__(self, SKEL, 4, 11, self.control_bounds, num_envs=num_envs, **kw)

    def step(self, a):
        a = np.asarray(a, dtype=np.float64).reshape(self.num_envs, 3)
        tau = np.zeros((self.num_envs, self.robot_skeleton.ndofs))
        tau[:, 3:] = np.clip(a, -1, 1) * SCALE
        x_before = self.robot_skeleton.q[:, 0]
        self.do_simulation(tau, self.frame_skip)
        q = self.robot_skeleton.q
        reward = (q[:, 0] - x_before) / self.dt + 1.0 - 1e-3 * np.square(a).sum(axis=1)
        s = self.state_vector()
        done = ~(np.isfinite(s).all(axis=1) & (np.abs(s[:, 2]) < 0.8) & (s[:, 1] > -0.5))
        return self._get_obs(), reward, done, [{} for _ in range(self.num_envs)]

    def _get_obs(self):
        return np.concatenate([self.robot_skeleton.q[:, 1:], np.clip(self.robot_skeleton.dq, -10, 10)], axis=1)

    def reset_model(self):
        qpos = self.init_qpos + self.uniform(-.005, .005, self.robot_skeleton.ndofs)
        qvel = self.init_qvel + self.uniform(-.005, .005, self.robot_skeleton.ndofs)
        self.set_state(qpos, qvel)
        return self._get_obs()


def reference_rollout(card, seeds, actions, frame_skip=4):
    """The same task evaluated env by env on oracle worlds (what N reference envs on pydart2 would do)."""
    from dart_env_amd import seeding
    from tests.oracle_lib import OracleWorld
    n, T = len(seeds), len(actions)
    obs = np.zeros((T + 1, n, 11)); rew = np.zeros((T, n)); done = np.zeros((T, n), dtype=bool)
    dt = card.dt * frame_skip
    for i, sd in enumerate(seeds):
        rng, _ = seeding.np_random(sd)
        w = OracleWorld(card)
        w.reset()
        q0, dq0 = w.get_state()
        w.set_state(q0 + rng.uniform(low=-.005, high=.005, size=6), dq0 + rng.uniform(low=-.005, high=.005, size=6))
        ob = lambda: np.concatenate([w.q[1:], np.clip(w.dq, -10, 10)])
        obs[0, i] = ob()
        for t in range(T):
            a = np.asarray(actions[t][i], dtype=np.float64)
            tau = np.zeros(6); tau[3:] = np.clip(a, -1, 1) * SCALE
            tau = tau.astype(np.float32).astype(np.float64)      # the boundary carries generalized forces as float32
            xb = w.q[0]
            for _ in range(frame_skip):
                w.set_forces(tau); w.step()
            s = np.concatenate([w.q, w.dq])
            rew[t, i] = (w.q[0] - xb) / dt + 1.0 - 1e-3 * np.square(a).sum()
            done[t, i] = not (np.isfinite(s).all() and abs(s[2]) < 0.8 and s[1] > -0.5)
            obs[t + 1, i] = ob()
    return obs, rew, done
