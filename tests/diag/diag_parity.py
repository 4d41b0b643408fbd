"""Diagnostic (GPU box): per-step fp32-vs-oracle error breakdown."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper
from tests.batch_oracle import OracleBatch
np.set_printoptions(precision=3, linewidth=200)
env_id = sys.argv[1] if len(sys.argv) > 1 else "DartHopper-v1"
n, steps = 256, 40
card = card_for(env_id); nd, na = card.ndofs, card.act_dim
rng = np.random.RandomState(0)
gpu = HipStepper(card, n, precision=32); ora = OracleBatch(card, n)
qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
gpu.reset(None, qn, vn); ora.reset(None, qn, vn)
for t in range(steps):
    a = rng.uniform(-1, 1, (n, na)).astype(np.float32)
    og, rg, dg, tg = gpu.step(a); oo, ro, do, to = ora.step(a)
    qg, dqg = gpu.get_state(); qo, dqo = ora.state()
    edq = np.abs(dqg - dqo); eq = np.abs(qg - qo)
    worst = np.unravel_index(edq.argmax(), edq.shape)
    print(t, "rms_q %.2e rms_dq %.2e max_q %.2e max_dq %.2e" % (np.sqrt((eq**2).mean()), np.sqrt((edq**2).mean()), eq.max(), edq.max()),
          "worst env,dof", worst, "dq_o", dqo[worst], "n(edq>1e-3)", int((edq.max(1) > 1e-3).sum()), "done mism", int((dg != do).sum()))
    if do.any():
        qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
        gpu.reset(do.astype(np.uint8), qn, vn, want_obs=False); ora.reset(do, qn, vn)
