"""First-launch vs later-launch repeatability of the step kernels (run on the GPU box): the same state and actions, stepped
repeatedly in one process, must give bitwise identical observations, states and contact reports every time.
    python tools/gpu/determinism.py [env ids ...]"""
import sys; sys.path.insert(0, ".")
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper, CFG_CONTACT_REPORT, StepperError
envs = sys.argv[1:] or ["DartHopper-v1", "DartWalker2d-v1", "DartHalfCheetah-v1", "DartCartPole-v1", "DartDoubleInvertedPendulumEnv-v1"]
total_bad = 0
for env_id in envs:
    card = card_for(env_id); n = 256; nd, na = card.ndofs, card.act_dim
    for prec in (64, 32):
        for report in (0, 1):
            rng = np.random.RandomState(5)
            q0 = rng.uniform(-0.3, 0.3, (n, nd)); dq0 = rng.uniform(-2, 2, (n, nd))
            if card.ground_y > -1e9: q0[:, 1] = rng.uniform(-0.65, -0.3, n)
            acts = rng.uniform(-1, 1, (5, n, na)).astype(np.float32)
            g = HipStepper(card, n, precision=prec)
            if report:
                try: g.configure(CFG_CONTACT_REPORT, 1)
                except StepperError: g.close(); continue
            ref = None; bad = 0
            for rep in range(4):
                g.set_state(q0, dq0)
                outs = []
                for t in range(5):
                    ob, r, d, tr = g.step(acts[t]); outs.append(ob.copy())
                    if report:
                        cnt, bod, pt, fc = g.contacts(); outs += [cnt.copy(), bod.copy(), fc.copy()]
                outs += list(g.get_state())
                if ref is None: ref = outs
                elif not all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, outs)): bad += 1
            g.close(); total_bad += bad
            print(env_id, "f%d" % prec, "report" if report else "lean  ", "launches that differ from the first:", bad, "of 3")
print("TOTAL", total_bad)
