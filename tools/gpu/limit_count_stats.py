#!/usr/bin/env python3
"""How many joints are at a limit at once, at scale (round 6: the limit slots of planar_kernel.hpp).  65 536 envs, random actions, auto-reset;
after every env-step the states come back and the joints at / beyond a limit are counted per env (a sample of the substeps: one in frame_skip).
    python tools/gpu/limit_count_stats.py [env-id] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
env_id = sys.argv[1] if len(sys.argv) > 1 else "DartWalker2d-v1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
card = card_for(env_id); n = 65536; nd = card.ndofs
lo = np.array([card.lower[d] for d in range(nd)]); hi = np.array([card.upper[d] for d in range(nd)]); lim = np.array([bool(card.limited[d]) for d in range(nd)])
g = st.HipStepper(card, n, precision=64)
g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_SEED, 3)
g.reset(None, None, None, want_obs=False)
rng = np.random.RandomState(0)
hist = np.zeros(int(lim.sum()) + 1, dtype=np.int64)
for mode, k in (("random actions U[-1,1)", steps), ("constant actions (+1 on every actuator)", 60), ("constant actions (-1)", 60)):
    hist[:] = 0
    for t in range(k):
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32) if mode.startswith("random") else np.full((n, card.act_dim), 1.0 if "+1" in mode else -1.0, np.float32)
        g.step(a)
        q, _ = g.get_state()
        at = ((q <= lo) | (q >= hi)) & lim
        hist += np.bincount(at.sum(axis=1), minlength=len(hist))
    print("%s, %s: joints at a limit per env, share of %d env-step samples: %s" % (env_id, mode, hist.sum(), " ".join("%d:%.2e" % (i, h / hist.sum()) for i, h in enumerate(hist))), flush=True)
g.close()
