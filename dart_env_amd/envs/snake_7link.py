"""DartSnake7Link-v1 single-env object (reference gym/envs/dart/snake_7link.py:6-124): seven capsule links sliding in the
x-z plane (root: prismatic x, prismatic z, revolute y), six actuated joints scaled by 200, frame_skip 4.  Before every
world step each body receives a fluid force along its own z axis opposing its COM velocity (:37-47) -- computed inside the
kernel's forward pass; reward dx/dt + 0.1 - 1e-3 sum a^2 - 0.1 |q[2]|, done when |q[2]| >= 1.5 or the state breaks."""
from .hopper import _SingleEnv


class DartSnake7LinkEnv(_SingleEnv):
    ENV_ID = "DartSnake7Link-v1"
