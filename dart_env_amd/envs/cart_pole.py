"""DartCartPole-v1 single-env object (reference gym/envs/dart/cart_pole.py:5-39): cart on a prismatic rail + pole,
dt 0.02 x frame_skip 2, tau[0] = a[0] * 100 WITHOUT clamping (:15-16), observation [q, dq] (:26-27), reward 1 per
step, done when the observation is not finite or |q[1]| > 0.2 (:21-22), reset noise +-0.01 (:31-32).  No contacts
(the rail has no collision shape); joint limits +-1 m / +-1.57 rad are enforced (dart_env.py:64-67)."""
from .hopper import _SingleEnv


class DartCartPoleEnv(_SingleEnv):
    ENV_ID = "DartCartPole-v1"
