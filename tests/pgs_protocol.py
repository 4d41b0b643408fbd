"""Fixed-sweep PGS parity protocol -- TEST INFRASTRUCTURE.

`north_star` names an iterative PGS LCP solver.  It is the kernels' `DART_CFG_SOLVER = 1` mode (fixed sweep counts); the oracle has the
same iteration behind `oracle_set_solver(PGS, K1, K2)` (oracle/dart_oracle.c::pgs_sweeps: Gauss-Seidel over the rows in the order
{n, t1, t2} per contact, then joint limits, then joint friction; stage 1 = the rows without a friction index, stage 2 = all rows with
the friction bounds of stage 1).  A Gauss-Seidel iterate after a FIXED number of sweeps depends on the row order, so "the device
PGS = the oracle PGS at K sweeps" is a statement about both the arithmetic and the order -- checked here far from convergence
(K = 30: the LCP residual is still ~1e-2) and near it (K = 80), against the oracle with the SAME K, never against the pivoting solve.
"""
import numpy as np

from dart_env_amd import stepper as st
from tests.batch_oracle import OracleBatch
from tests.oracle_lib import OracleWorld


def pgs_rollout(make_stepper, card, n, steps, K, seed=1, act_scale=1.0):
    """Steps `n` envs through `steps` env-steps with K PGS sweeps per stage on both sides; resets follow the oracle's done flags with
    identical noise.  Returns per-step max |dq| / |q| differences and the number of done-flag mismatches."""
    g = make_stepper(card, n)
    g.configure(st.CFG_SOLVER, st.SOLVER_PGS)
    g.configure(st.CFG_ITERS_STAGE1, K); g.configure(st.CFG_ITERS_STAGE2, K)
    ora = OracleBatch(card, n, solver=OracleWorld.PGS, k1=K, k2=K)
    rng = np.random.RandomState(seed)
    qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
    g.reset(None, qn, vn); ora.reset(None, qn, vn)
    edq, eq, mism = [], [], 0
    for t in range(steps):
        a = (rng.uniform(-1, 1, (n, card.act_dim)) * act_scale).astype(np.float32)
        _, _, d, _ = g.step(a)
        _, _, do, _ = ora.step(a)
        qg, dqg = g.get_state(); qo, dqo = ora.state()
        edq.append(float(np.abs(dqg - dqo).max())); eq.append(float(np.abs(qg - qo).max()))
        do = np.asarray(do, bool)
        mism += int((np.asarray(d).astype(bool) != do).sum())
        if do.any():
            qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
            g.reset(do.astype(np.uint8), qn, vn, want_obs=False); ora.reset(do, qn, vn)
    g.close()
    return {"dq": edq, "q": eq, "done_mismatches": mism}


def exact_vs_pgs_gap(card, n, steps, K, seed=1):
    """How far K sweeps are from the exact solve, on the ORACLE: max |dq| difference after one env-step from identical states, worst
    over `steps` env-steps of a PGS-K rollout.  Shows that the parity check compares iterates, not limits."""
    pgs = OracleBatch(card, n, solver=OracleWorld.PGS, k1=K, k2=K)
    exact = OracleBatch(card, n)
    rng = np.random.RandomState(seed)
    qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
    pgs.reset(None, qn, vn); exact.reset(None, qn, vn)
    worst = 0.0
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        for we, wp in zip(exact.worlds, pgs.worlds):
            we.set_state(wp.q, wp.dq)
        _, _, do, _ = pgs.step(a)
        exact.step(a)
        worst = max(worst, float(np.abs(pgs.state()[1] - exact.state()[1]).max()))
        do = np.asarray(do, bool)
        if do.any():
            qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
            pgs.reset(do, qn, vn); exact.reset(do, qn, vn)
    return worst
