"""dartk::log_cr ON THE DEVICE (csrc/cr_log.hpp through tests/gpu_kernels/crlog_harness.hip): correctly rounded against 50-digit decimal
arithmetic, like the host build of the same header (tests/test_cr_log.py) -- which it was NOT until the double-double primitives carried
`fp contract(off)`: hipcc's default -ffp-contract=fast fuses the product of an error-free transformation into the add that consumes it after
inlining, and 18.8 % of the results came out one ulp off.  Also: the legacy-Gaussian factor f = sqrt(-2 log(r2) / r2) built from it equals
the host's wherever the host libm's log is correctly rounded (division and sqrt are IEEE-exact on both sides)."""
import ctypes as C
import math
import os
from decimal import Decimal, getcontext

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_kernels", "libcrlog_harness.so")


def test_device_log_cr_is_correctly_rounded():
    import torch  # noqa: F401  (one HSA runtime per process: torch's first, as dart_env_amd.stepper does)
    L = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    L.crlog_run.argtypes = [dp, dp, dp, C.c_long]
    rng = np.random.RandomState(2)
    x1 = 2 * rng.random_sample(40000) - 1; x2 = 2 * rng.random_sample(40000) - 1
    r2 = x1 * x1 + x2 * x2
    r2 = np.ascontiguousarray(np.concatenate([r2[(r2 < 1) & (r2 > 0)], [1.0, 1 - 2.0 ** -53, 2.0 ** -106, 0.5, 0.7071067811865476, 0.7071067811865475, 3e-5]]))
    lg = np.zeros_like(r2); f = np.zeros_like(r2)
    assert L.crlog_run(r2.ctypes.data_as(dp), lg.ctypes.data_as(dp), f.ctypes.data_as(dp), len(r2)) == 0
    getcontext().prec = 50
    cr = np.array([float(Decimal(float(v)).ln()) for v in r2])
    assert np.array_equal(lg, cr), int((lg != cr).sum())
    host = np.array([math.log(v) for v in r2])
    agree = host == cr                                             # where glibc's log is correctly rounded ...
    assert agree.mean() > 0.995
    inner = r2 < 1.0
    hf = np.array([math.sqrt(-2.0 * math.log(v) / v) for v in r2[inner]])
    assert np.array_equal(f[inner][agree[inner]], hf[agree[inner]])   # ... the device's Gaussian factor is the host's, bit for bit
