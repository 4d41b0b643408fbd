"""Dump the states of a short auto-reset rollout (for bitwise A/B of two library builds: DART_STEPPER_LIB=... python tools/gpu/ab_states.py out.npz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
env_id = os.environ.get("ENV_ID", "DartWalker2d-v1"); n = int(os.environ.get("N", "65536")); steps = int(os.environ.get("STEPS", "16"))
card = card_for(env_id)
s = st.HipStepper(card, n, precision=int(os.environ.get("PREC", "64")))
s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 9)
if os.environ.get("FORCE_SLOW") == "1":
    s.configure(st.CFG_DEBUG_FORCE_FALLBACK, 1)
s.reset(None, None, None, want_obs=False)
rng = np.random.RandomState(1)
qs, dqs = [], []
for t in range(steps):
    a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
    s.step(a)
    q, dq = s.get_state(); qs.append(q); dqs.append(dq)
np.savez(sys.argv[1], q=np.array(qs), dq=np.array(dqs))
print("saved", sys.argv[1], "static", s.query(st.Q_STATIC_KERNEL))
