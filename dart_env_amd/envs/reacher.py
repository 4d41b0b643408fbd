"""DartReacher-v1 (reference gym/envs/dart/reacher2d.py:5-66: 2-dof arm in the x-z plane, nothing collides) and
DartReacher3d-v1 (reference gym/envs/dart/reacher.py:5-61: 5-dof arm, reward / done from the fingertip distance BEFORE
the step) single-env objects.  Each env owns a reach target that reset_model resamples by rejection from the env's
np_random stream; it lives on the device as per-env task state (dart_set_task_state) and enters reward and observation
there."""
from .hopper import _SingleEnv


class DartReacher2dEnv(_SingleEnv):
    ENV_ID = "DartReacher-v1"

    @property
    def target(self):
        return self._task_state[0, :3].copy()


class DartReacherEnv(_SingleEnv):
    ENV_ID = "DartReacher3d-v1"

    @property
    def target(self):
        return self._task_state[0, :3].copy()
