"""Diagnostic (GPU box): replay one env-step as 4 single substeps, fp32 vs fp64 kernel vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper
from tests.oracle_lib import OracleWorld
np.set_printoptions(precision=7, linewidth=220, suppress=True)
card = card_for("DartHopper-v1"); card.frame_skip = 1
q0 = np.array([-0.018121, -0.008582, -0.077262, 0.053724, -0.277822, 0.471387])
dq0 = np.array([-1.240102, -0.279882, -7.803305, 0., -17.268038, 16.1093])
a = np.array([[0.175952, -0.525151, 0.349856]], dtype=np.float32)
g32 = HipStepper(card, 1, precision=32); g64 = HipStepper(card, 1, precision=64); w = OracleWorld(card)
for s in (g32, g64): s.set_state(q0[None], dq0[None])
w.set_state(q0, dq0)
for f in range(4):
    g32.step(a); g64.step(a); w.env_step(a[0].astype(np.float64))
    print("sub", f)
    print("  oracle q", w.q, "dq", w.dq, "lam", w.last_lcp()[0])
    print("  gpu64  q", g64.get_state()[0][0], "dq", g64.get_state()[1][0])
    print("  gpu32  q", g32.get_state()[0][0], "dq", g32.get_state()[1][0])
    # re-sync fp32 to the fp64 state so each substep is judged on its own
    q, dq = g64.get_state(); g32.set_state(q, dq)
