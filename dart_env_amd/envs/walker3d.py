"""DartWalker3d-v1 single-env object (reference gym/envs/dart/walker3d.py:8-113): 21-dof biped with box links,
15 actions scaled 150 (waist) / 100 / 20 (ankles) (:10-13), frame_skip 4, obs = q[1:], clip(dq) (41) (:99-105),
reward dx/dt + 1 - 1e-3 sum a^2 - 0.2 knee-limit penalty - 1e-3 |z|, zero when done (:75-92), done on height in
(1.05, 2.0) and up/forward angles of bodynodes[0] < 0.84 (:87-89).  Runs on the generic spatial kernel.

The reference switches the skeleton's self-collision check on (:26): the card carries self_collision = 1 and the kernel
generates link-link box contacts (ODE dBoxBox restated) next to the link-ground ones."""
from .hopper import _SingleEnv


class DartWalker3dEnv(_SingleEnv):
    ENV_ID = "DartWalker3d-v1"
