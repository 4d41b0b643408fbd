"""The wave-cooperative boxed-LCP solver (dart_env_amd/csrc/wave_blcp.hpp: Gauss-Jordan elimination of the free block in registers, rows =
lanes, pivots through v_readlane) WITHOUT a GPU: the GPU test's harness source (tests/gpu_kernels/wave_blcp_harness.hip) compiled by g++
against the fiber runtime of tests/kernel_emu/fake_wave_include, fed the same random contact-shaped problems and held to the same
conditions as tests/test_gpu_wave_blcp.py -- every register variant (8 ... 40 rows) and the lane kernels' instantiation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_wave_blcp import check_group_independence, enumerate_solution, kkt_violation, make_problems

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_emu")


def _lib():
    subprocess.check_call(["make", "-s", "-C", DIR, "libdart_wave_blcp_emu.so"])
    L = C.CDLL(os.path.join(DIR, "libdart_wave_blcp_emu.so"))
    for name, ct in (("wave_blcp_run_f64", C.c_double), ("wave_blcp_run_f32", C.c_float)):
        f = getattr(L, name)
        p = C.POINTER(ct)
        f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, p, p, p, p, p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                      C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        f.restype = C.c_int
    return L


@pytest.mark.parametrize("real,mp,ext", [("f64", 8, 0), ("f64", 12, 0), ("f64", 16, 0), ("f64", 24, 0), ("f64", 32, 0), ("f64", 40, 0), ("f64", 16, 1),
                                         ("f64", 24, 1), ("f32", 8, 0), ("f32", 16, 0), ("f32", 40, 0), ("f32", 16, 1),
                                         ("f64", 16, 4), ("f64", 16, 7), ("f32", 16, 4)])   # ext >= 4: sp_blcp4_t, four problems per wave
@pytest.mark.parametrize("zero_bounds,rank_deficient", [(0, False), (1, False), (0, True)])
def test_wave_solver_on_the_host_returns_the_lcp_solution(real, mp, ext, zero_bounds, rank_deficient):
    L = _lib()
    rng = np.random.RandomState(1000 * mp + 10 * ext + 2 * zero_bounds + int(rank_deficient))   # the GPU test's problems
    n = 48
    A, b, lo, hi, m, pin, U, full = make_problems(rng, n, mp, bool(zero_bounds), rank_deficient)
    dt, ct, fn = (np.float64, C.c_double, L.wave_blcp_run_f64) if real == "f64" else (np.float32, C.c_float, L.wave_blcp_run_f32)
    arrs = [np.ascontiguousarray(a, dtype=dt) for a in (A, b, lo, hi)]
    x = np.zeros((n, mp), dtype=dt)
    F = np.zeros(n, np.uint64); Uio = U.copy(); ok = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = fn(n, mp, ext, mp, *[P(a, ct) for a in arrs], P(x, ct), P(m, C.c_int), P(pin, C.c_uint64), P(F, C.c_uint64), P(Uio, C.c_uint64),
            P(ok, C.c_int), P(it, C.c_int), 200, zero_bounds, 0)
    assert rc == 0
    assert ok.all() if not rank_deficient else ok.mean() >= 0.95, (np.where(ok == 0)[0][:5], m[ok == 0][:5])
    tol = (1e-9 if not rank_deficient else 1e-6) if real == "f64" else (2e-4 if not rank_deficient else np.inf)
    worst = 0.0
    for p in np.where(ok == 1)[0]:
        k = int(m[p])
        xs = x[p, :k].astype(np.float64)
        assert np.all(np.isfinite(xs))
        worst = max(worst, kkt_violation(full[p], b[p, :k], arrs[2][p, :k].astype(np.float64), arrs[3][p, :k].astype(np.float64), xs,
                                         1e-9 if real == "f64" else 1e-5))
        assert np.all(x[p, k:] == 0)             # rows beyond m are left alone (their operands were NaN)
    assert worst < tol, worst
    if ext:
        assert it[ok == 1].max() < 200 and it.min() >= 0 and (it > 0).any()
    for p in [p for p in range(n) if m[p] <= 6 and ok[p] and np.isfinite(tol)][:12]:
        k = int(m[p])
        ref = enumerate_solution(full[p], b[p, :k], lo[p, :k], hi[p, :k])
        assert np.abs(ref - x[p, :k]).max() < (1e-8 if real == "f64" else 5e-3) * (1 if not rank_deficient else 1e3) * (1 + np.abs(ref).max()), p


def test_four_problem_solver_on_the_host_is_independent_of_group_and_neighbours():
    check_group_independence(_lib(), n=48)
