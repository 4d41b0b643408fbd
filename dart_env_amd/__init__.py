"""dart_env_amd -- MI355X-native batched replacement for the DartEnv step path of DartEnv/dart-env.

    import dart_env_amd
    env  = dart_env_amd.make("DartHopper-v1")                 # single env, TimeLimit-wrapped, like gym.make
    venv = dart_env_amd.vector.make("DartHopper-v1", 65536)   # gym.vector-shaped, one HIP launch per step
"""
from . import vector  # noqa: F401
from .registration import make, spec  # noqa: F401
from .vector import DartVectorEnv  # noqa: F401

__version__ = "0.1.0"
