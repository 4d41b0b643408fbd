"""Launch-to-launch repeatability of the lane-per-env kernels: the same states and actions stepped again in the same process must
give bitwise identical observations, states and contact reports.  (Round 2 found register tiers of 12+ LCP rows inlined into
the step kernel giving a first launch that differed from the later ones on gfx950 -- round 5: spill copies ahead of an EXEC restore,
tools/exec_prologue_lint.py; round 6 deleted those tiers -- and this test keeps watching every lane kernel.)"""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for

pytestmark = pytest.mark.gpu

ENVS = ["DartHopper-v1", "DartWalker2d-v1", "DartHalfCheetah-v1", "DartSnake7Link-v1", "DartCartPole-v1", "DartReacher-v1", "DartReacher3d-v1",
        "DartDoubleInvertedPendulumEnv-v1"]


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("report", [False, True])
@pytest.mark.parametrize("env_id", ENVS)
def test_same_inputs_same_bits(env_id, report, precision):
    from dart_env_amd.stepper import HipStepper, CFG_CONTACT_REPORT, StepperError
    card = card_for(env_id); n = 256; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(5)
    q0 = rng.uniform(-0.3, 0.3, (n, nd)); dq0 = rng.uniform(-2, 2, (n, nd))
    if card.ground_y > -1e9:
        q0[:, 1] = rng.uniform(-0.65, -0.3, n)          # from above the floor to lying in it: every tier and the fallback solver
    acts = rng.uniform(-1, 1, (5, n, na)).astype(np.float32)
    g = HipStepper(card, n, precision=precision)
    if report:
        try:
            g.configure(CFG_CONTACT_REPORT, 1)
        except StepperError:
            g.close(); pytest.skip("no contact report on this kernel")
    first = None
    for rep in range(4):
        g.set_state(q0, dq0)
        outs = []
        for t in range(5):
            ob, r, d, tr = g.step(acts[t]); outs += [ob.copy(), r.copy(), d.copy()]
            if report:
                cnt, bod, pt, fc = g.contacts(); outs += [cnt.copy(), bod.copy(), fc.copy()]
        outs += list(g.get_state())
        if first is None:
            first = outs
        else:
            for k, (a, b) in enumerate(zip(first, outs)):
                assert np.array_equal(a, b, equal_nan=True), (rep, k)
    g.close()
