from .dart_env import BatchedDartEnv, DartEnv  # noqa: F401
from .hopper import DartHopperEnv  # noqa: F401
from .walker2d import DartWalker2dEnv  # noqa: F401
from .human_walker import DartHumanWalkerEnv  # noqa: F401
from .walker3d import DartWalker3dEnv  # noqa: F401
from .cart_pole import DartCartPoleEnv  # noqa: F401
from .half_cheetah import DartHalfCheetahEnv  # noqa: F401
from .cartpole_swingup import DartCartPoleSwingUpEnv, DartDoubleInvertedPendulumEnv  # noqa: F401
from .snake_7link import DartSnake7LinkEnv  # noqa: F401
from .reacher import DartReacher2dEnv, DartReacherEnv  # noqa: F401
from .walker3d_spd import DartWalker3dSPDEnv  # noqa: F401
from .dog import DartDogEnv  # noqa: F401
