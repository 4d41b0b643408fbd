"""Oracle-backed stand-in for dart_env_amd.stepper.HipStepper (TEST INFRASTRUCTURE): lets the host-side env layer
(seeding, reset noise, auto-reset, TimeLimit bookkeeping, dtypes, async misuse errors) run on a box without a GPU.
It is injected explicitly through ``stepper_factory=``; the product never imports it."""
import numpy as np

from dart_env_amd import stepper as st
from tests import oracle_lib as ol
from tests.batch_oracle import OracleBatch


class OracleStepper:
    def __init__(self, card, num_envs, device=0, precision=64):
        self.card, self.num_envs = card, num_envs
        self.ndofs, self.obs_dim, self.act_dim = card.ndofs, card.obs_dim, card.act_dim
        self.batch = OracleBatch(card, num_envs)
        self.cfg = {st.CFG_AUTORESET: 0, st.CFG_SEED: 0, st.CFG_ENV_OFFSET: 0}
        self.episode = np.zeros(num_envs, dtype=np.uint32)
        self._pending = None
        self.closed = False

    def configure(self, key, value):
        self.cfg[key] = value

    def query(self, what):
        return {st.Q_NUM_ENVS: self.num_envs, st.Q_NDOFS: self.ndofs, st.Q_OBS_DIM: self.obs_dim,
                st.Q_ACT_DIM: self.act_dim}[what]

    def _philox(self, i):
        self.episode[i] += 1
        return ol.philox_noise(int(self.cfg[st.CFG_SEED]), int(self.cfg[st.CFG_ENV_OFFSET]) + i, int(self.episode[i]),
                               self.card.reset_noise, self.card.reset_noise_vel, self.ndofs)

    def reset(self, mask=None, qpos_noise=None, qvel_noise=None, want_obs=True):
        n = self.num_envs
        m = np.ones(n, dtype=bool) if mask is None else np.asarray(mask).astype(bool)
        if qpos_noise is None:
            qn = np.zeros((n, self.ndofs)); vn = np.zeros((n, self.ndofs))
            for i in np.flatnonzero(m):
                qn[i], vn[i] = self._philox(i)
        else:
            qn, vn = np.asarray(qpos_noise).reshape(n, -1), np.asarray(qvel_noise).reshape(n, -1)
        self.batch.reset(m, qn, vn)
        return self.batch.obs().astype(np.float32) if want_obs else None

    def set_task_state(self, mask, values):
        m = np.ones(self.num_envs, dtype=bool) if mask is None else np.asarray(mask).astype(bool)
        for i in np.flatnonzero(m):
            self.batch.worlds[i].set_task_state(np.asarray(values)[i])

    def set_state(self, q, dq):
        for i, w in enumerate(self.batch.worlds):
            w.set_state(q[i], dq[i])

    def get_state(self):
        return self.batch.state()

    def counters(self):
        return self.batch.elapsed.astype(np.int32), self.episode.copy()

    def step(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.act_dim)
        obs, rew, done, trunc = self.batch.step(a)
        if self.cfg[st.CFG_AUTORESET] and done.any():
            self.reset(done, None, None, want_obs=False)
            obs = self.batch.obs()
        return obs.astype(np.float32), rew, done.astype(np.bool_), trunc.astype(np.bool_)

    def step_async(self, actions):
        if self._pending is not None:
            raise st.AlreadyPendingCallError(st.E_PENDING, "step_async called while a step is pending")
        self._pending = np.array(actions, dtype=np.float32)

    def step_wait(self):
        if self._pending is None:
            raise st.NoAsyncCallError(st.E_NOT_PENDING, "step_wait called without step_async")
        a, self._pending = self._pending, None
        return self.step(a)

    def close(self):
        self.closed = True
