"""How often does the fp32 arithmetic of the planar kernels take a contact / limit event on another substep than fp64?

The kernels' device code compiled for the host (tests/emu_lib.py) in fp32 against the same code in fp64, both fed the oracle's
episode resets, compared per env-step: an env-step is *flipped* when the set of touching capsules at its last world step
differs between the two precisions.  Reported per 10^6 substeps, with the untrimmed RMS state error those flips cause.
    python -m tests.diag.diag_event_flips DartWalker2d-v1 1024 500
"""
import sys

import numpy as np

from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
from tests.parity_protocol import make_reference


def run(env_id, n, steps):
    card = card_for(env_id)
    acts, ref = make_reference(card, n, steps)
    g32, g64 = EmuStepper(card, n, precision=32), EmuStepper(card, n, precision=64)
    for g in (g32, g64):
        g.configure(st.CFG_AUTORESET, 0); g.configure(st.CFG_SEED, 0); g.configure(st.CFG_ENV_OFFSET, 0)
        g.enable_contact_report(True)
        g.reset(None, None, None, want_obs=False)
    flips = first_flips = 0
    diverged = np.zeros(n, bool)
    sq = 0.0; cnt = 0
    for t in range(steps):
        g32.step(acts[t]); g64.step(acts[t])
        c32, b32, _, _ = g32.contacts(); c64, b64, _, _ = g64.contacts()
        same = (c32 == c64) & np.all(b32[:, :, 0] == b64[:, :, 0], axis=1)
        flips += int((~same).sum())
        first_flips += int((~same & ~diverged).sum())
        diverged |= ~same
        q32, dq32 = g32.get_state(); q64, dq64 = g64.get_state()
        e = dq32 - dq64
        sq += float((e ** 2).sum()); cnt += e.size
        m = ref["done"][t]
        if m.any():
            g32.reset(m, None, None, want_obs=False); g64.reset(m, None, None, want_obs=False)
            diverged &= ~m.astype(bool)
    sub = n * steps * card.frame_skip
    print("%s: %d envs x %d env-steps (%d substeps): env-steps with a different touching set %d = %.0f per 1e6 substeps; first departures "
          "(an env leaving the fp64 event sequence) %d = %.1f per 1e6 substeps; untrimmed RMS dq error over all env-steps %.2e"
          % (env_id, n, steps, sub, flips, flips / sub * 1e6, first_flips, first_flips / sub * 1e6, np.sqrt(sq / cnt)))


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
