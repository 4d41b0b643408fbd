"""``gym.make`` for the Dart ids this package serves (reference gym/envs/__init__.py:206-211, 265-276, 284-288;
gym/envs/registration.py:81-97: the registry wraps the env in TimeLimit(max_episode_steps))."""
from .model_card import TASKS
from .wrappers import TimeLimit


class EnvSpec:
    def __init__(self, task):
        self.id = task.env_id
        self.max_episode_steps = task.max_episode_steps
        self.reward_threshold = task.reward_threshold


def spec(env_id):
    if env_id not in TASKS:
        raise KeyError("No registered env with id: %s (served ids: %s)" % (env_id, sorted(TASKS)))
    return EnvSpec(TASKS[env_id])


def make(env_id, **kwargs):
    from .envs import (DartCartPoleEnv, DartCartPoleSwingUpEnv, DartDogEnv, DartDoubleInvertedPendulumEnv, DartHalfCheetahEnv, DartHopperEnv, DartHumanWalkerEnv, DartReacher2dEnv, DartReacherEnv, DartSnake7LinkEnv, DartWalker2dEnv,
                       DartWalker3dEnv, DartWalker3dSPDEnv)
    cls = {"DartHopper-v1": DartHopperEnv, "DartWalker2d-v1": DartWalker2dEnv,
           "DartWalker3d-v1": DartWalker3dEnv, "DartHumanWalker-v1": DartHumanWalkerEnv,
           "DartCartPole-v1": DartCartPoleEnv, "DartHalfCheetah-v1": DartHalfCheetahEnv,
           "DartCartPoleSwingUp-v1": DartCartPoleSwingUpEnv, "DartDoubleInvertedPendulumEnv-v1": DartDoubleInvertedPendulumEnv,
           "DartSnake7Link-v1": DartSnake7LinkEnv, "DartReacher-v1": DartReacher2dEnv, "DartReacher3d-v1": DartReacherEnv,
           "DartWalker3dSPD-v1": DartWalker3dSPDEnv, "DartDog-v1": DartDogEnv}
    s = spec(env_id)
    env = cls[env_id](**kwargs)
    env.spec = s
    return TimeLimit(env, max_episode_steps=s.max_episode_steps)
