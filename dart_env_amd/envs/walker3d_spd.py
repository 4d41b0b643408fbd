"""DartWalker3dSPD-v1 single-env object (reference gym/envs/dart/walker3d_spd.py:9-138): the Walker3d model whose 15
actions are target joint angles; a stable-PD controller (:40-55) turns them into torques before EVERY world step from the
mass matrix, the Coriolis/gravity forces and the previous step's constraint forces -- all inside the kernel
(csrc/spatial_kernel.hpp::sp_spd_torque).  The reference declares obs_dim 42 but returns 41 numbers (:27, :115-121);
the observation here has the 41."""
from .hopper import _SingleEnv


class DartWalker3dSPDEnv(_SingleEnv):
    ENV_ID = "DartWalker3dSPD-v1"

    def _info(self, done):
        return {"done_return": done}
