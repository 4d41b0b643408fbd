"""A physics-only user model (tests/pogo_env.py: no termination inside the library) under random torques: the hoppers fall over and stay
on the floor, with more capsules touching than the register tiers of their lane kernel hold -- what does a batched world step cost
then?  (python tools/gpu/pogo_probe.py [num_envs]; DART_STEPPER_LIB picks an A/B library.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.pogo_env import PogoEnv     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for prec in (64, 32):
    env = PogoEnv(num_envs=n, precision=prec)
    env.seed(0); env.reset()
    rng = np.random.RandomState(0)
    taus = [np.concatenate([np.zeros((n, 3)), rng.uniform(-1, 1, (n, 3)) * [40, 30, 15]], axis=1).astype(np.float32) for _ in range(8)]
    out = []
    for block in range(8):
        t0 = time.perf_counter()
        for t in range(40):
            env.do_simulation(taus[t % 8], 4)
        dt = (time.perf_counter() - t0) / 40
        q = env.robot_skeleton.q
        out.append("%d-%d: %.0f us (mean height %.2f)" % (40 * block, 40 * block + 40, dt * 1e6, float(np.mean(q[:, 1]))))
    print("pogo x%d fp%d per do_simulation(frame_skip 4), host-synchronous:" % (n, prec), "; ".join(out))
    env.close()
