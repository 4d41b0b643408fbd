"""csrc/cr_log.hpp on the host: dartk::log_cr is the correctly rounded natural logarithm (against 50-digit decimal arithmetic) on the
arguments numpy's legacy polar Gaussian feeds it -- r2 = x1^2 + x2^2 in (0, 1) -- and on the edges.  The device uses it for the double
pendulum's Gaussian reset noise (csrc/mt19937_kernels.hpp) because the device libm's log is only 1-ulp accurate."""
import ctypes as C
import math
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_emu")


def test_log_cr_is_correctly_rounded():
    subprocess.check_call(["make", "-s", "-C", DIR, "libdart_crlog.so"])
    L = C.CDLL(os.path.join(DIR, "libdart_crlog.so"))
    L.cr_log_many.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long]
    rng = np.random.RandomState(3)
    x1 = 2 * rng.random_sample(30000) - 1; x2 = 2 * rng.random_sample(30000) - 1
    r2 = x1 * x1 + x2 * x2
    r2 = r2[(r2 < 1) & (r2 > 0)]
    edges = np.array([1.0, 1 - 2.0 ** -53, 2.0 ** -106, 0.5, 0.7071067811865476, 0.7071067811865475, 1e-300, 3e-5, 0.25, 0.9999999])
    x = np.concatenate([r2, edges, 10.0 ** rng.uniform(-30, 0, 2000)])
    x = x[x <= 1.0]
    out = np.zeros_like(x)
    L.cr_log_many(x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), len(x))
    getcontext().prec = 50
    wrong = [float(v) for v, o in zip(x, out) if float(Decimal(float(v)).ln()) != o]
    assert not wrong, wrong[:5]
    # and how far the host libm is from that: a fraction of a percent of the draws, which is what keeps the device stream from being
    # bit-exact with numpy's in every single draw (tests/test_gpu_golden_and_properties.py::test_device_gaussian_resets_...)
    off = sum(1 for v, o in zip(x, out) if math.log(v) != o)
    assert off < 0.005 * len(x)
