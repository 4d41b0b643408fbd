"""DartCartPoleSwingUp-v1 (reference gym/envs/dart/cartpole_swingup.py:7-54) and DartDoubleInvertedPendulumEnv-v1
(reference gym/envs/dart/inverted_double_pendulum.py:8-65) single-env objects: cart on a rail with one / two poles and a
tip weight, dt 0.01 x frame_skip 2, tau[0] = a[0] * 40 without clamping.  Their reset_model draws more than two uniform
vectors (a third draw decides +-pi; Gaussian velocity noise), so reset noise always comes from the host numpy stream."""
from .hopper import _SingleEnv


class DartCartPoleSwingUpEnv(_SingleEnv):
    ENV_ID = "DartCartPoleSwingUp-v1"


class DartDoubleInvertedPendulumEnv(_SingleEnv):
    ENV_ID = "DartDoubleInvertedPendulumEnv-v1"
