"""The lane kernels' boxed-LCP solver (blcp_bpp / blcp_bpp_mixed in dart_env_amd/csrc/planar_kernel.hpp) on its own, in the host build of
the kernel source (tests/kernel_emu/emu_blcp.cpp): random contact-shaped problems at the row counts of the Hopper / Walker2d /
half-cheetah tiers, checked against the complementarity conditions and -- up to seven rows -- the unique solution found by enumerating
active sets.  The GPU counterpart for the wave-cooperative solver is tests/test_gpu_wave_blcp.py, whose generators this test shares."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_wave_blcp import enumerate_solution, kkt_violation

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_emu")


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-s", "-C", DIR, "libdart_lane_blcp.so"])
    lib = C.CDLL(os.path.join(DIR, "libdart_lane_blcp.so"))
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    lib.lane_blcp_run.argtypes = [C.c_int] * 5 + [dp, dp, dp, dp, up, up, up, dp, C.c_int]
    lib.lane_blcp_run.restype = C.c_int
    return lib


def problems(rng, n, M, zero_bounds, cfm):
    tri = M * (M + 1) // 2
    A = np.zeros((n, tri)); b = np.zeros((n, M)); lo = np.zeros((n, M)); hi = np.zeros((n, M))
    pin = np.zeros(n, np.uint32); U = np.zeros(n, np.uint32); full = []
    for p in range(n):
        cols = rng.randint(max(2, M // 2), M + 3)          # from a rank-deficient Jacobian to a well-conditioned one
        G = rng.normal(size=(M, cols))
        act = rng.uniform(size=M) < 0.8                    # inactive rows are decoupled unit rows, as constraint_phase leaves them
        Ap = (G @ G.T / cols) * np.outer(act, act) + np.diag(np.where(act, cfm, 1.0))
        full.append(Ap)
        A[p] = [Ap[i, j] for i in range(M) for j in range(i + 1)]
        b[p] = np.where(act, rng.normal(size=M) * rng.choice([0.1, 1.0, 30.0]), 0.0)
        kinds = rng.choice(3 if zero_bounds else 4, size=M, p=[.5, .2, .3] if zero_bounds else [.4, .15, .15, .3])
        for i, kd in enumerate(kinds):
            if not act[i] or kd == 2: lo[p, i] = hi[p, i] = 0.0; pin[p] |= np.uint32(1 << i)
            elif kd == 0: hi[p, i] = np.inf
            elif kd == 1: lo[p, i] = -np.inf; U[p] |= np.uint32(1 << i)
            else:
                h = abs(rng.normal()) * rng.choice([0.05, 1.0]); lo[p, i], hi[p, i] = -h, h
    return A, b, lo, hi, pin, U, full


@pytest.mark.parametrize("M", [5, 7, 10, 14])
@pytest.mark.parametrize("real,zero_bounds,presolve32", [(64, 0, 0), (64, 1, 0), (64, 0, 1), (32, 0, 0)])
def test_lane_solver_returns_the_lcp_solution(L, M, real, zero_bounds, presolve32):
    rng = np.random.RandomState(100 * M + real + 2 * zero_bounds + presolve32)
    n = 400
    A, b, lo, hi, pin, U, full = problems(rng, n, M, bool(zero_bounds), 1e-3)
    F = np.zeros(n, np.uint32); Uio = U.copy(); x = np.zeros((n, M))
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    assert L.lane_blcp_run(n, M, real, zero_bounds, presolve32, P(A, C.c_double), P(b, C.c_double), P(lo, C.c_double), P(hi, C.c_double),
                           P(pin, C.c_uint32), P(F, C.c_uint32), P(Uio, C.c_uint32), P(x, C.c_double), 200) == 0
    viol = np.array([kkt_violation(full[p], b[p], lo[p].astype(np.float32 if real == 32 else np.float64).astype(np.float64),
                                   hi[p].astype(np.float32 if real == 32 else np.float64).astype(np.float64), x[p], 1e-9 if real == 64 else 1e-5)
                     for p in range(n)])
    assert np.all(np.isfinite(x))
    if real == 64:
        assert viol.max() < 1e-10, (viol.max(), int(np.argmax(viol)))       # every one of them, to the solver's own tolerance
        small = [p for p in range(n) if M <= 7][:60]
        for p in small:
            ref = enumerate_solution(full[p], b[p], lo[p], hi[p])
            assert np.abs(ref - x[p]).max() < 1e-6 * (1 + np.abs(ref).max()), p
    else:
        assert np.median(viol) < 1e-6 and np.percentile(viol, 99) < 1e-4      # fp32 on condition numbers up to ~1e4 (the rare borderline row aside)
