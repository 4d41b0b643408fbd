"""DartDog-v1 single-env object (reference gym/envs/dart/dog.py:9-65): a quadruped whose trunk hangs on a FREE joint
(22 dofs; DART's FreeJoint coordinates: q[0:3] rotation vector, q[3:6] translation, dq[0:6] the trunk's body-frame twist),
16 leg torques x 200, reward 0.6 dx/dt + 1 - 1e-3 sum a^2, done when the trunk's COM leaves 0.7 < y < 1.8 or |z| >= 0.4.
The kernel integrates the root pose the way DART does (Q <- Q * exp(twist dt)) and runs the dynamics on an internal
translation + rotation chain re-centred on that pose every world step (csrc/spatial_free_root.hpp)."""
from .hopper import _SingleEnv


class DartDogEnv(_SingleEnv):
    ENV_ID = "DartDog-v1"
