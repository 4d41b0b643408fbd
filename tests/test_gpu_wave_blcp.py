"""The wave-cooperative boxed-LCP solver (dart_env_amd/csrc/wave_blcp.hpp) on its own: random contact-shaped problems through
tests/gpu_kernels/wave_blcp_harness.hip, every register variant and both instantiations (the tree kernel's and the lane kernels'),
checked against the complementarity conditions and -- small problems -- against the unique solution found by enumerating active sets.

    w = A x - b,   lo <= x <= hi,   lo < x < hi => w = 0,   x = lo => w >= 0,   x = hi => w <= 0        (A symmetric positive definite)
"""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "gpu_kernels", "libwave_blcp_harness.so")
INF = np.inf


def _lib():
    if not os.path.exists(LIB):     # normally built by __graft_entry__.build(); a 7 s hipcc run otherwise
        import subprocess
        root = os.path.dirname(HERE)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-Wno-unused-value", "-I" + os.path.join(root, "dart_env_amd", "csrc"), "-I" + os.path.join(root, "include"),
                               os.path.join(HERE, "gpu_kernels", "wave_blcp_harness.hip"), "-o", LIB])
    L = C.CDLL(LIB)
    for name, ct in (("wave_blcp_run_f64", C.c_double), ("wave_blcp_run_f32", C.c_float)):
        f = getattr(L, name)
        p = C.POINTER(ct)
        f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, p, p, p, p, p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                      C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        f.restype = C.c_int
    return L


def test_harness_library_is_built_and_exports_its_entry_points():
    """(no GPU needed) __graft_entry__.build() compiles the harness next to the product library"""
    L = _lib()
    assert L.wave_blcp_run_f64 and L.wave_blcp_run_f32


def make_problems(rng, n, mcap, zero_bounds, rank_deficient):
    """contact-shaped rows: unilateral [0, inf), upper (-inf, 0], friction boxes [-h, h] (not with zero_bounds), pinned 0"""
    tri = mcap * (mcap + 1) // 2
    A = np.full((n, tri), np.nan); b = np.full((n, mcap), np.nan); lo = np.full((n, mcap), np.nan); hi = np.full((n, mcap), np.nan)
    m = rng.randint(1, mcap + 1, n).astype(np.int32)
    m[:3] = [1, mcap, mcap]
    pin = np.zeros(n, np.uint64); U = np.zeros(n, np.uint64)
    full = []
    for p in range(n):
        k = int(m[p])
        cols = max(1, k // 2) if rank_deficient else k + 2
        G = rng.normal(size=(k, cols))
        Ap = G @ G.T / cols + (1e-3 if rank_deficient else 1e-2) * np.eye(k)
        full.append(Ap)
        for i in range(k):
            for j in range(i + 1):
                A[p, i * (i + 1) // 2 + j] = Ap[i, j]
        b[p, :k] = rng.normal(size=k) * rng.choice([0.1, 1.0, 30.0])
        kinds = rng.choice(3 if zero_bounds else 4, size=k, p=[.5, .2, .3] if zero_bounds else [.4, .15, .15, .3])
        for i, kd in enumerate(kinds):
            if kd == 0: lo[p, i], hi[p, i] = 0.0, INF
            elif kd == 1: lo[p, i], hi[p, i] = -INF, 0.0; U[p] |= np.uint64(1 << i)
            elif kd == 2: lo[p, i], hi[p, i] = 0.0, 0.0; pin[p] |= np.uint64(1 << i)
            else:
                h = abs(rng.normal()) * rng.choice([0.05, 1.0]); lo[p, i], hi[p, i] = -h, h
    return A, b, lo, hi, m, pin, U, full


def kkt_violation(Ap, b, lo, hi, x, rel=1e-9):
    w = Ap @ x - b
    scale = 1.0 + np.abs(b).max()
    v = 0.0
    for i in range(len(x)):
        if lo[i] == hi[i]:
            v = max(v, abs(x[i] - lo[i])); continue
        v = max(v, lo[i] - x[i], x[i] - hi[i])
        eps = rel * (1 + abs(x[i]))
        at_lo, at_hi = x[i] <= lo[i] + eps, x[i] >= hi[i] - eps
        if at_lo: v = max(v, -w[i] / scale)
        elif at_hi: v = max(v, w[i] / scale)
        else: v = max(v, abs(w[i]) / scale)
    return v


def enumerate_solution(Ap, b, lo, hi):
    """the unique solution by trying every active set (rows: free / at lo / at hi)"""
    k = len(b)
    choices = []
    for i in range(k):
        c = ["f"] if lo[i] != hi[i] else []
        if np.isfinite(lo[i]): c.append("l")
        if np.isfinite(hi[i]) and hi[i] != lo[i]: c.append("h")
        choices.append(c)
    for S in itertools.product(*choices):
        x = np.array([0.0 if s == "f" else (lo[i] if s == "l" else hi[i]) for i, s in enumerate(S)])
        fr = [i for i, s in enumerate(S) if s == "f"]
        if fr:
            bd = [i for i in range(k) if i not in fr]
            x[fr] = np.linalg.solve(Ap[np.ix_(fr, fr)], b[fr] - Ap[np.ix_(fr, bd)] @ x[bd])
        if kkt_violation(Ap, b, lo, hi, x) < 1e-10:
            return x
    raise AssertionError("no active set is feasible")


@pytest.mark.gpu
@pytest.mark.parametrize("real,mp,ext", [("f64", 8, 0), ("f64", 12, 0), ("f64", 16, 0), ("f64", 24, 0), ("f64", 32, 0), ("f64", 40, 0), ("f64", 16, 1), ("f64", 24, 1),
                                         ("f32", 8, 0), ("f32", 12, 0), ("f32", 16, 0), ("f32", 40, 0), ("f32", 16, 1), ("f32", 24, 1),
                                         ("f64", 16, 4), ("f64", 16, 6), ("f32", 16, 4)])   # ext >= 4: sp_blcp4_t, four problems per wave
@pytest.mark.parametrize("zero_bounds,rank_deficient", [(0, False), (1, False), (0, True)])
def test_wave_solver_returns_the_lcp_solution(real, mp, ext, zero_bounds, rank_deficient):
    L = _lib()
    rng = np.random.RandomState(1000 * mp + 10 * ext + 2 * zero_bounds + int(rank_deficient))
    n = 192
    A, b, lo, hi, m, pin, U, full = make_problems(rng, n, mp, bool(zero_bounds), rank_deficient)
    dt, ct, fn = (np.float64, C.c_double, L.wave_blcp_run_f64) if real == "f64" else (np.float32, C.c_float, L.wave_blcp_run_f32)
    arrs = [np.ascontiguousarray(a, dtype=dt) for a in (A, b, lo, hi)]
    x = np.zeros((n, mp), dtype=dt)
    F = np.zeros(n, np.uint64); Uio = U.copy(); ok = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = fn(n, mp, ext, mp, *[P(a, ct) for a in arrs], P(x, ct), P(m, C.c_int), P(pin, C.c_uint64), P(F, C.c_uint64), P(Uio, C.c_uint64),
            P(ok, C.c_int), P(it, C.c_int), 200, zero_bounds, 0)
    assert rc == 0
    # Well-conditioned problems all converge.  With a rank-deficient J (Delassus matrix = low rank + cfm, condition ~1e4) block pivoting
    # with its single-pivot rule can cycle past the cap on a large problem -- the tail the tree kernel's PGS safety net and the lane
    # kernels' clamp exist for: a few such problems may come back unfinished, everything that is reported solved must be solved.
    assert ok.all() if not rank_deficient else ok.mean() >= 0.97, (np.where(ok == 0)[0][:5], m[ok == 0][:5])
    tol = (1e-9 if not rank_deficient else 1e-6) if real == "f64" else (2e-4 if not rank_deficient else np.inf)   # fp32 cannot resolve cond 1e4
    worst = 0.0
    for p in np.where(ok == 1)[0]:
        k = int(m[p])
        xs = x[p, :k].astype(np.float64)
        assert np.all(np.isfinite(xs))
        # (the problem the solver saw: operands rounded to its precision)
        worst = max(worst, kkt_violation(full[p], b[p, :k], arrs[2][p, :k].astype(np.float64), arrs[3][p, :k].astype(np.float64), xs,
                                         1e-9 if real == "f64" else 1e-5))
        assert np.all(x[p, k:] == 0)             # rows beyond m are left alone (their operands were NaN)
    assert worst < tol, worst
    if ext:
        assert it[ok == 1].max() < 200 and it.min() >= 0 and (it > 0).any()     # the lane kernels' instantiation reports its iterations
    small = [p for p in range(n) if m[p] <= 6 and ok[p] and np.isfinite(tol)][:40]
    for p in small:
        k = int(m[p])
        ref = enumerate_solution(full[p], b[p, :k], lo[p, :k], hi[p, :k])
        assert np.abs(ref - x[p, :k]).max() < (1e-8 if real == "f64" else 5e-3) * (1 if not rank_deficient else 1e3) * (1 + np.abs(ref).max()), p


@pytest.mark.gpu
def test_wave_solver_at_its_cap_keeps_the_last_iterate_inside_the_box():
    """keep_last (the lane kernels' callers): with the iteration cap at 1 an unfinished solve still leaves a point inside the box;
    without it the output is not touched."""
    L = _lib()
    rng = np.random.RandomState(7)
    n, mp = 64, 16
    A, b, lo, hi, m, pin, U, full = make_problems(rng, n, mp, False, False)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for keep in (1, 0):
        x = np.full((n, mp), 123.0)
        F = np.zeros(n, np.uint64); Uio = U.copy(); ok = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        assert L.wave_blcp_run_f64(n, mp, 1, mp, P(A, C.c_double), P(b, C.c_double), P(lo, C.c_double), P(hi, C.c_double), P(x, C.c_double),
                                   P(m, C.c_int), P(pin, C.c_uint64), P(F, C.c_uint64), P(Uio, C.c_uint64), P(ok, C.c_int), P(it, C.c_int), 1, 0, keep) == 0
        unfinished = np.where(ok == 0)[0]
        assert len(unfinished) > 5
        for p in unfinished:
            k = int(m[p])
            if keep:
                assert np.all(x[p, :k] >= lo[p, :k]) and np.all(x[p, :k] <= hi[p, :k])
            else:
                assert np.all(x[p, :k] == 123.0)


def check_group_independence(L, n=96):
    """sp_blcp4_t (four problems per wave, one per row of 16 lanes): a problem's solution, final sets and convergence flag are BITWISE the same
    whichever group it sits in and whoever its three neighbours are -- what keeps an env's trajectory independent of the batch around it
    (dart_env_amd/csrc/planar_kernel.hpp: wave_constraints4) -- and they agree with the one-problem-per-wave solver on the same problems."""
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for real, zb in (("f64", 0), ("f64", 1), ("f32", 0)):
        rng = np.random.RandomState(99 + zb)
        A, b, lo, hi, m, pin, U, full = make_problems(rng, n, 16, bool(zb), False)
        dt, ct, fn = (np.float64, C.c_double, L.wave_blcp_run_f64) if real == "f64" else (np.float32, C.c_float, L.wave_blcp_run_f32)
        arrs = [np.ascontiguousarray(a, dtype=dt) for a in (A, b, lo, hi)]

        def run(ext, order=None):
            o = np.arange(n) if order is None else order
            aa = [np.ascontiguousarray(a[o]) for a in arrs]
            x = np.zeros((n, 16), dtype=dt)
            mm = np.ascontiguousarray(m[o]); pp = np.ascontiguousarray(pin[o]); F = np.zeros(n, np.uint64); Uio = np.ascontiguousarray(U[o])
            ok = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
            assert fn(n, 16, ext, 16, *[P(a, ct) for a in aa], P(x, ct), P(mm, C.c_int), P(pp, C.c_uint64), P(F, C.c_uint64), P(Uio, C.c_uint64),
                      P(ok, C.c_int), P(it, C.c_int), 200, zb, 1) == 0
            inv = np.empty(n, np.int64); inv[o] = np.arange(n)
            return x[inv], F[inv], Uio[inv], ok[inv]
        base = run(4)
        assert base[3].all()
        for rot in (1, 2, 3):
            r = run(4 + rot)
            assert all(np.array_equal(a, c) for a, c in zip(base, r)), rot
        shuffled = run(4, np.random.RandomState(5).permutation(n))       # other neighbours, other groups
        assert all(np.array_equal(a, c) for a, c in zip(base, shuffled))
        one = run(1)                                                     # the one-problem solver: same sets, same solution up to rounding
        assert np.array_equal(one[1], base[1]) and np.array_equal(one[2], base[2])
        assert np.abs(one[0].astype(np.float64) - base[0]).max() < (1e-10 if real == "f64" else 1e-3) * (1 + np.abs(base[0]).max())


@pytest.mark.gpu
def test_four_problem_solver_is_independent_of_group_and_neighbours():
    check_group_independence(_lib())
