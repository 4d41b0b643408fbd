"""DartHalfCheetah-v1 single-env object (reference gym/envs/dart/half_cheetah.py:5-114): planar 9-dof cheetah, dt 0.01 x
frame_skip 5 (:18), clamp to +-1 and scale [120, 90, 60, 120, 60, 30] (:30-41), reward dx/dt + 1 - 0.1 sum a^2, zeroed
when the state broke (:51-63), done additionally when |q[2]| >= 1.3 (:43-46), observation q[1:], dq unclipped
(:80-86).  Every capsule of the robot collides with the floor; since round 2 it runs on the planar register kernel
(csrc/planar_kernel.hpp, CheetahTopo: contact-slot tiers of 2 and 4 / 3, welded head folded into the torso, joint springs)."""
from .hopper import _SingleEnv


class DartHalfCheetahEnv(_SingleEnv):
    ENV_ID = "DartHalfCheetah-v1"
