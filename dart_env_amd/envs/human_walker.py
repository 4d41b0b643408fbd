"""DartHumanWalker-v1 single-env object (reference gym/envs/dart/human_walker.py:15-165): 29-dof "kima" human,
23 actions scaled by 1.5 x [120,...] (:18), frame_skip 15 (:29), observation = q[1:], clip(dq), two foot-contact
flags (:140-149), reward 2(1-|1-v|) + 2 - 0.5 sum|a| - 3|z_head|, zero when done (:109-128).  Runs on the generic
spatial kernel (box feet, Euler/universal joints, ankle springs)."""
import numpy as np

from .hopper import _SingleEnv


class DartHumanWalkerEnv(_SingleEnv):
    ENV_ID = "DartHumanWalker-v1"

    def _info(self, done):
        s = self.state_vector()
        broke = not (np.isfinite(s).all() and (np.abs(s[2:]) < 100).all())
        # vel_rew / action_pen / deviation_pen of the reference's info dict stay on the device; the flags are kept
        return {"broke_sim": bool(broke), "done_return": done, "dyn_model_id": 0, "state_index": 0}
