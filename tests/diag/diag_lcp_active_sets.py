"""Why does a lane need more than one factorisation in the pivoting LCP loop?  (research tool; see tests/diag/emu_trace.cpp)

For every run of blcp_bpp in a rollout of the host-built planar kernel: the active set it started from, the one it ended with and
the number of solves.  Prints, per stage, the share of runs by number of solves and -- for the runs that needed more than one --
which kind of row the start guess had wrong (contact normal / friction / joint limit; guessed free but bound, or bound but free).

    g++ ... -o /tmp/libdart_planar_emu_trace.so tests/diag/emu_trace.cpp      (command in that file)
    DART_EMU_LIB=/tmp/libdart_planar_emu_trace.so python tests/diag/diag_lcp_active_sets.py DartWalker2d-v1
"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dart_env_amd import stepper as st          # noqa: E402
from dart_env_amd.model_card import card_for    # noqa: E402
from tests.emu_lib import EmuStepper            # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "DartWalker2d-v1"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
card = card_for(env_id)
n = 512
g = EmuStepper(card, n, precision=64)
L = g.L
L.emu_trace_size.restype = C.c_int64
g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_SEED, 0)
g.reset()
rng = np.random.RandomState(0)
for t in range(60):
    g.step((rng.uniform(-1, 1, (n, card.act_dim)) * scale).astype(np.float32))
L.emu_trace_clear()
for t in range(120):
    g.step((rng.uniform(-1, 1, (n, card.act_dim)) * scale).astype(np.float32))
k = L.emu_trace_size()
rec = np.zeros((k, 8), dtype=np.uint32)
L.emu_trace_get(rec.ctypes.data_as(C.POINTER(C.c_uint32)))
for zb, name in ((1, "stage 1 (frictionless; rows: contact normals, then limits)"), (0, "stage 2 (rows: normal, friction per slot, then limits)")):
    r = rec[rec[:, 1] == zb]
    if not len(r):
        continue
    M = int(r[0, 0]); its = r[:, 7] + 1      # `it` counts completed non-final iterations
    print("%s: %d runs, M = %d; solves: %s" % (name, len(r), M, {int(i): round(float((its == i).mean()), 4) for i in np.unique(its)[:6]}))
    multi = r[its > 1]
    nslots = (M - (6 if "Walker2d" in env_id or "Cheetah" in env_id else 3)) // (1 if zb else 2)
    kinds = collections.Counter()
    for F0, F1 in zip(multi[:, 2], multi[:, 4]):
        diff = int(F0) ^ int(F1)
        for i in range(M):
            if (diff >> i) & 1:
                if zb:
                    kind = "normal" if i < nslots else "limit"
                else:
                    kind = ("normal" if i % 2 == 0 else "friction") if i < 2 * nslots else "limit"
                kinds[(kind, "guessed free, ends on a bound" if (F0 >> i) & 1 else "guessed bound, ends free")] += 1
    tot = max(1, sum(kinds.values()))
    for (kind, what), c in kinds.most_common():
        print("    %-9s %-30s %5.1f %% of the wrong guesses" % (kind, what, 100.0 * c / tot))
    nwrong = np.array([bin(int(a) ^ int(b)).count("1") for a, b in zip(multi[:, 2], multi[:, 4])])
    print("    rows guessed wrong per multi-solve run:", {int(i): round(float((nwrong == i).mean()), 3) for i in np.unique(nwrong)[:6]})
