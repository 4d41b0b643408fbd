"""Chaotic amplification in DartHalfCheetah-v1: two copies of the fp64 ORACLE started 1e-15 apart, same random actions.
Background for tests/test_gpu_long_parity.py (the half cheetah is compared with the oracle over 100 env-steps, not 1 000).
    python tests/diag/diag_cheetah_chaos.py"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from tests.batch_oracle import OracleBatch
card = card_for("DartHalfCheetah-v1"); n = 64; nd, na = card.ndofs, card.act_dim
rng = np.random.RandomState(0)
a_ = OracleBatch(card, n); b_ = OracleBatch(card, n)
qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
a_.reset(None, qn, vn); b_.reset(None, qn + 1e-15 * rng.standard_normal((n, nd)), vn)
for t in range(1, 401):
    a = rng.uniform(-1, 1, (n, na)).astype(np.float32)
    a_.step(a); b_.step(a)
    if t in (1, 10, 50, 100, 200, 300, 400):
        qa, _ = a_.state(); qb, _ = b_.state()
        print(t, "rms dq between two oracle copies started 1e-15 apart: %.2e" % np.sqrt(np.mean((qa - qb) ** 2)))
