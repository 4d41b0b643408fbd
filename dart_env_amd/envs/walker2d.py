"""DartWalker2d-v1 single-env object (reference gym/envs/dart/walker2d.py:6-85): scale [100,100,20,100,100,20] (:9),
obs 17 (:10), done 0.8 < h < 2.0 and |ang| < 1.0 (:60-61); same kernel as the hopper with the 7-link tree topology."""
from .hopper import _SingleEnv


class DartWalker2dEnv(_SingleEnv):
    ENV_ID = "DartWalker2d-v1"
