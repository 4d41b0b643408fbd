"""Diagnostic (GPU box): which side of a fp64 GPU / oracle mismatch in DartWalker3dSPD-v1 is the inexact one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper, CFG_STATS
from tests.batch_oracle import OracleBatch
card = card_for("DartWalker3dSPD-v1")
n, nd, na = 48, card.ndofs, card.act_dim
rng = np.random.RandomState(21)
gpu = HipStepper(card, n, precision=64); gpu.configure(CFG_STATS, 1)
ora = OracleBatch(card, n)
qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
gpu.reset(None, qn, vn); ora.reset(None, qn, vn)
for t in range(60):
    a = rng.uniform(-1.2, 1.2, (n, na)).astype(np.float32)
    gpu.step(a); oo, ro, do, to = ora.step(a)
    h1, h2 = gpu.solver_stats()
    qg, dqg = gpu.get_state(); qo, dqo = ora.state()
    e = np.abs(dqg - dqo).max(axis=1)
    res = np.array([w.last_lcp()[4] for w in ora.worlds])
    print(t, "max err %.2e env %d" % (e.max(), e.argmax()), "gpu fallbacks", int(h2[0]), "iters>=31:", int(h1[31]), "oracle last-step residual max %.2e" % res.max())
    if e.max() > 1e-5:
        gpu.set_state(qo, dqo)
    if do.any():
        qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
        gpu.reset(do.astype(np.uint8), qn, vn, want_obs=False); ora.reset(do, qn, vn)
