"""DartHumanWalker-v1 single-env object (reference gym/envs/dart/human_walker.py:15-165): 29-dof "kima" human,
23 actions scaled by 1.5 x [120,...] (:18), frame_skip 15 (:29), observation = q[1:], clip(dq), two foot-contact
flags (:140-149), reward 2(1-|1-v|) + 2 - 0.5 sum|a| - 3|z_head|, zero when done (:109-128).  Runs on the generic
spatial kernel (box feet, Euler/universal joints, ankle springs)."""
import numpy as np

from .hopper import _SingleEnv


class DartHumanWalkerEnv(_SingleEnv):
    ENV_ID = "DartHumanWalker-v1"

    def _com(self, which):
        """world COM of card.aux_body[which] (0: the progress body bodynodes[1], 1: the head), human_walker.py:78-92"""
        st = self._stepper
        if not hasattr(st, "body_poses"):       # an injected stand-in stepper without pose getters
            return None
        return st.body_poses()[2][0, self.card.aux_body[which]]

    def step(self, a):
        a = np.asarray(a, dtype=np.float64)
        before = self._com(0)
        self._terms = None
        ob, reward, done, _ = super().step(a)
        after, head = self._com(0), self._com(1)
        if before is not None:
            # the three reward terms the reference also returns in `info` (human_walker.py:111-117, 135-137), recomputed on the
            # host from the body poses either side of the step; the reward itself comes from the kernel
            tv = float(self.card.aux_real[0])
            vel = (after[0] - before[0]) / self.dt
            self._terms = {"vel_rew": 2.0 * (tv - abs(tv - vel)), "action_pen": float(self.card.aux_real[2]) * float(np.abs(a).sum()),
                           "deviation_pen": float(self.card.aux_real[3]) * abs(float(head[2]))}
        return ob, reward, done, self._info(done)

    def _info(self, done):
        s = self.state_vector()
        broke = not (np.isfinite(s).all() and (np.abs(s[2:]) < 100).all())
        info = {"broke_sim": bool(broke), "done_return": done, "dyn_model_id": 0, "state_index": 0}
        if getattr(self, "_terms", None):
            info.update(self._terms)
        return info
