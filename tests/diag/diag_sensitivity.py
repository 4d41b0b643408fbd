"""How much per-step rounding noise can a rollout take before the north star's untrimmed RMS bound (1e-4) breaks?

Two fp64 oracle batches get the same actions and resets; the second one has its state multiplied by (1 + eps * N(0,1))
after every env step (a stand-in for arithmetic of relative accuracy eps).  Resets follow the first batch's done flags.
Prints the untrimmed RMS of q and dq differences over all env-steps, and the number of done-flag mismatches.
    python -m tests.diag.diag_sensitivity DartHopper-v1 256 300
"""
import sys
import numpy as np
from dart_env_amd.model_card import card_for
from tests.batch_oracle import OracleBatch


def run(env_id, n, steps, eps, act_scale=1.0, seed=0, where="both"):
    card = card_for(env_id)
    nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(seed)
    prng = np.random.RandomState(seed + 99)
    A, B = OracleBatch(card, n), OracleBatch(card, n)
    qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
    A.reset(None, qn, vn); B.reset(None, qn, vn)
    sq = sdq = 0.0; cnt = 0; mism = 0; worst = 0.0
    for t in range(steps):
        a = (rng.uniform(-1, 1, (n, na)) * act_scale).astype(np.float32)
        _, _, dA, _ = A.step(a); _, _, dB, _ = B.step(a)
        qa, dqa = A.state(); qb, dqb = B.state()
        e, ed = qb - qa, dqb - dqa
        sq += (e ** 2).sum(); sdq += (ed ** 2).sum(); cnt += e.size
        worst = max(worst, np.abs(ed).max())
        mism += int((dA != dB).sum())
        # inject noise into B
        for i, w in enumerate(B.worlds):
            q, dq = qb[i].copy(), dqb[i].copy()
            if where in ("both", "q"): q *= 1 + eps * prng.standard_normal(nd)
            if where in ("both", "dq"): dq *= 1 + eps * prng.standard_normal(nd)
            w.set_state(q, dq)
        if dA.any():
            qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
            A.reset(dA, qn, vn); B.reset(dA, qn, vn)
    return np.sqrt(sq / cnt), np.sqrt(sdq / cnt), mism, worst


if __name__ == "__main__":
    env_id = sys.argv[1]; n = int(sys.argv[2]); steps = int(sys.argv[3])
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    for eps in (1e-7, 1e-9, 1e-11, 1e-13):
        print(env_id, "eps", eps, "rms_q %.3e rms_dq %.3e done-mismatch %d max|ddq| %.2e" % run(env_id, n, steps, eps, scale), flush=True)
