import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from dart_env_amd.model_card import card_for
from dart_env_amd import stepper as st
card = card_for(os.environ.get("ENV_ID", "DartHumanWalker-v1"))
if len(sys.argv) > 1: card.contact_cfm = float(sys.argv[1])
n = 4096
s = st.HipStepper(card, n, precision=int(os.environ.get("PREC", "64")))
s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_STATS, 1)
s.reset(None, None, None, want_obs=False)
rng = np.random.RandomState(0)
t0 = time.time()
for t in range(20):
    s.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32) * (0.3 if t % 2 else 1.0))
s.sync(); dt = time.time() - t0
h1, h2 = s.solver_stats()
names=["kin+dyn(lane0)","mass rows","chol+qdd","contacts/rows","J rows","W fwd","A=WWt","BPP stage1","BPP stage2","dv update"]
tot=h2[8:18].sum()
print("phase cycles %:", {nm: round(100*float(c)/float(tot),1) for nm,c in zip(names,h2[8:18])})
print("contact_cfm", card.contact_cfm, "iters hist", h1.tolist(), "fallbacks", int(h2[0]), "solves", int(h2[1]), "time/step ms", dt / 20 * 1e3)
q, dq = s.get_state(); print("finite", np.isfinite(q).all(), "max|dq|", np.abs(dq).max())
