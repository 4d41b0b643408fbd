"""Batch of fp64 oracle worlds stepped env-by-env (TEST INFRASTRUCTURE)."""
import numpy as np

from tests.oracle_lib import OracleWorld


class OracleBatch:
    def __init__(self, card, n, solver=OracleWorld.EXACT, k1=30, k2=30):
        self.card, self.n = card, n
        self.worlds = [OracleWorld(card, solver, k1, k2) for _ in range(n)]
        self.elapsed = np.zeros(n, dtype=np.int64)

    def reset(self, mask, qn, vn):
        for i, w in enumerate(self.worlds):
            if mask is None or mask[i]:
                w.reset()
                q, dq = w.get_state()
                w.set_state(q + qn[i], dq + vn[i])
                w.env_after_reset()
                self.elapsed[i] = 0

    def obs(self):
        return np.stack([w.env_obs() for w in self.worlds])

    def state(self):
        qs, dqs = zip(*[w.get_state() for w in self.worlds])
        return np.stack(qs), np.stack(dqs)

    def step(self, actions):
        """actions (n, act) float32 -> obs f64, reward f64, done (incl. time limit), truncated"""
        obs, rew, done, trunc = [], [], [], []
        for i, w in enumerate(self.worlds):
            o, r, d = w.env_step(actions[i].astype(np.float64))
            self.elapsed[i] += 1
            t = self.card.max_episode_steps > 0 and self.elapsed[i] >= self.card.max_episode_steps
            obs.append(o); rew.append(r); done.append(d or t); trunc.append(t and not d)
        return np.stack(obs), np.array(rew), np.array(done), np.array(trunc)
