"""What would binning envs to lanes buy the lane kernels?  (result: profiles/r06_lane_binning.txt -- a measured negative.)

A wave of a lane kernel iterates its pivoting loops until its slowest lane has converged, so an env-step costs the wave the MAXIMUM over
its 64 lanes of every (substep, stage) solve count.  This tool replays a random-action rollout on the host build of the kernels (the
tracing build of tests/diag/emu_trace.cpp, one record per run of the pivoting loop), and re-assigns the envs to 64-lane groups before
every env-step by a sort key: the identity (today), two oracles that know the step's own solve counts in advance (upper bounds, not
realisable), and keys a kernel could really compute -- the previous env-step's solve counts and LCP signature.

    g++ -O2 -fPIC -std=c++17 -march=native -ffp-contract=fast -Wno-unused-value -Wno-attributes -Itests/kernel_emu/fake_include \
        -Idart_env_amd/csrc -shared -o /tmp/libdart_planar_emu_trace.so tests/diag/emu_trace.cpp
    N=4096 T=40 python tools/diag_lane_binning.py DartHopper-v1 [--patterns]

--patterns: also list the start sets that needed more than one solve (first traced env-step): which rows moved."""
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DART_EMU_LIB"] = os.environ.get("TRLIB", "/tmp/libdart_planar_emu_trace.so")
from dart_env_amd import stepper as st          # noqa: E402
from dart_env_amd.model_card import card_for    # noqa: E402
from tests.emu_lib import EmuStepper            # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    env_id = args[0] if args else "DartHopper-v1"
    card = card_for(env_id)
    n, T = int(os.environ.get("N", "4096")), int(os.environ.get("T", "40"))
    g = EmuStepper(card, n, precision=64)
    L = g.L
    L.emu_trace_size.restype = C.c_int64
    g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_SEED, 0)
    g.reset()
    rng = np.random.RandomState(0)
    for t in range(60):
        g.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32))
    fs = card.frame_skip
    S = np.zeros((T, n, fs, 2), dtype=np.int32)       # solves of (step, env, substep, stage); 0 = the stage did not run
    rec0 = None
    for t in range(T):
        L.emu_trace_clear()
        g.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32))
        k = L.emu_trace_size()
        rec = np.zeros((k, 10), dtype=np.uint32)     # M, zero_bounds, F0, U0, F1, U1, pin, iters, env, substep
        L.emu_trace_get10(rec.ctypes.data_as(C.POINTER(C.c_uint32)))
        S[t, rec[:, 8].astype(int), rec[:, 9].astype(int), 1 - rec[:, 1].astype(int)] = rec[:, 7].astype(int) + 1
        if t == 0:
            rec0 = rec

    def wave_total(S_t, perm):
        return S_t[perm].reshape(n // 64, 64, fs, 2).max(axis=1).sum() / (n // 64)

    keys = {
        "identity": lambda t: np.arange(n),
        "oracle_total": lambda t: np.argsort(S[t].sum(axis=(1, 2)), kind="stable"),
        "oracle_lex": lambda t: np.lexsort(S[t].reshape(n, -1).T[::-1]),
        "prev_total": lambda t: np.argsort(S[t - 1].sum(axis=(1, 2)), kind="stable"),
        "prev_sig_end": lambda t: np.lexsort((S[t - 1][:, fs - 1, 1], S[t - 1][:, fs - 1, 0])),
        "prev_extra+sig": lambda t: np.lexsort(((S[t - 1] > 1).sum(axis=(1, 2)), S[t - 1][:, fs - 1, 1] > 0, S[t - 1][:, fs - 1, 0] > 0)),
    }
    print("%s, %d envs x %d env-steps: wave solves per env-step by lane assignment" % (env_id, n, T - 1))
    for name, key in keys.items():
        print("  %-16s %.2f" % (name, sum(wave_total(S[t], key(t)) for t in range(1, T)) / (T - 1)))
    print("  %-16s %.2f" % ("lane mean", S[1:].sum() / ((T - 1) * n)))
    if "--patterns" in sys.argv:
        for zb in (1, 0):
            r = rec0[rec0[:, 1] == zb]
            print("stage %d: %d problems, %.3f need more than one solve" % (1 if zb else 2, len(r), (r[:, 7] > 0).mean()))
            c = collections.Counter((int(M), int(pin), int(F0), int(U0), int(F1), int(U1), int(it)) for M, _, F0, U0, F1, U1, pin, it, _, _ in r if it > 0)
            for (M, pin, F0, U0, F1, U1, it), v in c.most_common(16):
                print("   M=%d pin=%s  F0=%s U0=%s -> F1=%s U1=%s  solves %d  %5.2f %% of the stage's problems" %
                      (M, bin(pin), bin(F0), bin(U0), bin(F1), bin(U1), it + 1, 100.0 * v / len(r)))


if __name__ == "__main__":
    main()
