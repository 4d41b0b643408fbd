"""DartHumanWalker-v1 single-env object (reference gym/envs/dart/human_walker.py:15-165): 29-dof "kima" human,
23 actions scaled by 1.5 x [120,...] (:18), frame_skip 15 (:29), observation = q[1:], clip(dq), two foot-contact
flags (:140-149), reward 2(1-|1-v|) + 2 - 0.5 sum|a| - 3|z_head|, zero when done (:109-128).  Runs on the generic
spatial kernel (box feet, Euler/universal joints, ankle springs)."""
import numpy as np

from .hopper import _SingleEnv


class DartHumanWalkerEnv(_SingleEnv):
    ENV_ID = "DartHumanWalker-v1"

    def _coms(self):
        """world COMs of card.aux_body[0] (the progress body bodynodes[1]) and [1] (the head), human_walker.py:78-92 -- one pose
        launch + one D2H copy for both"""
        st = self._stepper
        if not hasattr(st, "body_poses"):       # an injected stand-in stepper without pose getters
            return None
        com = st.body_poses()[2][0]
        return com[self.card.aux_body[0]].copy(), com[self.card.aux_body[1]].copy()

    def reset(self):
        self._post = None                       # the cached post-step poses belong to the previous episode
        return super().reset()

    def set_state(self, *a, **k):
        self._post = None
        return super().set_state(*a, **k)

    def step(self, a):
        a = np.asarray(a, dtype=np.float64)
        # the pose before this step is the pose after the previous one unless the state was touched in between
        pre = getattr(self, "_post", None) or self._coms()
        self._terms = None
        ob, reward, done, _ = super().step(a)
        self._post = post = self._coms()
        if done:
            self._post = None                   # the env layer may auto-reset on done: never carry a pose across it
        before = None if pre is None else pre[0]
        after, head = (None, None) if post is None else post
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
