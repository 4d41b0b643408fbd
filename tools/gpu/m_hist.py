import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dart_env_amd.model_card import card_for
from dart_env_amd import stepper as st
for env_id in ("DartHumanWalker-v1", "DartWalker3d-v1"):
    card = card_for(env_id); n = 4096
    s = st.HipStepper(card, n, precision=64)
    s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_STATS, 1)
    s.reset(None, None, None, want_obs=False)
    rng = np.random.RandomState(0); hist = np.zeros(72, int); hc = np.zeros(24, int)
    for t in range(40):
        s.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32))
        if t % 4 == 3:
            D = s.debug_dump()
            m = D[:, 0].astype(int); ncp = D[:, 1].astype(int)
            for v in m: hist[min(v, 71)] += 1
            for v in ncp: hc[min(v, 23)] += 1
    print(env_id, "rows m of the last LCP of an env-step (10 samples x 4096 envs):")
    print("  m histogram:", {i: int(c) for i, c in enumerate(hist) if c})
    print("  contact points:", {i: int(c) for i, c in enumerate(hc) if c})
    s.close()
