#!/usr/bin/env python3
"""Which REGISTER does the kernel read before writing it?  (follow-up of tools/gpu/first_launch_probe.py: with every vector /
accumulator register set to a fixed value before each rollout the results are repeatable; with leftovers in them they are not.)

One process.  test(S): three identical rollouts, before each one every register outside S is set to 0 and every register in S to a fresh
random word; if the three digests differ, some register in S is read uninitialised AND its value matters.  Bisection over v8..v255 and
a0..a255 lists the culprits one by one (each found register is pinned to 0 afterwards).

    DART_STEPPER_LIB=abtest/lib_ctab.so python tools/gpu/first_launch_bisect.py [--env DartHalfCheetah-v1 --prec 32 --report]"""
import argparse, ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper, CFG_CONTACT_REPORT

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="DartHalfCheetah-v1"); ap.add_argument("--prec", type=int, default=32); ap.add_argument("--report", action="store_true")
ap.add_argument("--n", type=int, default=256); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--max-found", type=int, default=6)
a = ap.parse_args()
H = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "gpu_kernels", "libpoison_harness.so"))
H.poison_regs_subset.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
card = card_for(a.env); n = a.n; nd, na = card.ndofs, card.act_dim
rng = np.random.RandomState(5)
q0 = rng.uniform(-0.3, 0.3, (n, nd)); dq0 = rng.uniform(-2, 2, (n, nd))
if card.ground_y > -1e9:
    q0[:, 1] = rng.uniform(-0.65, -0.3, n)
acts = rng.uniform(-1, 1, (a.steps, n, na)).astype(np.float32)
g = HipStepper(card, n, precision=a.prec)
if a.report:
    g.configure(CFG_CONTACT_REPORT, 1)
prng = np.random.RandomState(99)


def masks(S):
    mv = (C.c_uint32 * 8)(); ma = (C.c_uint32 * 8)()
    for f, i in S:
        (mv if f == "v" else ma)[i // 32] |= 1 << (i % 32)
    return mv, ma


def rollout(S, pattern):
    g.set_state(q0, dq0)
    mv, ma = masks(S)
    assert H.poison_regs_subset(C.c_uint32(pattern), C.c_uint32(0), mv, ma) == 0
    h = hashlib.sha1()
    for t in range(a.steps):
        ob, r, d, tr = g.step(acts[t]); h.update(ob.tobytes())
    q, dq = g.get_state(); h.update(q.tobytes()); h.update(dq.tobytes())
    return h.hexdigest()[:10]


def varies(S):
    ds = {rollout(S, int(prng.randint(1, 2**31 - 1)) | 1) for _ in range(3)}
    return len(ds) > 1, ds


allregs = [("v", i) for i in range(8, 256)] + [("a", i) for i in range(256)]
base = {rollout([], 0) for _ in range(3)}
print("all registers 0 before each rollout: digests", base, flush=True)
v_all, ds = varies(allregs)
print("all registers random before each rollout: varies =", v_all, ds, flush=True)
found = []
cand = list(allregs)
while v_all and len(found) < a.max_found:
    S = [r for r in cand if r not in found]
    ok, _ = varies(S)
    if not ok:
        break
    while len(S) > 1:
        half = S[:len(S) // 2]
        ok, _ = varies(half)
        S = half if ok else S[len(S) // 2:]
    ok, ds = varies(S)
    print("culprit:", "%s%d" % S[0], "varies alone =", ok, sorted(ds), flush=True)
    if not ok:
        print("(not reproducible alone: an interaction of several registers; stopping)"); break
    found.append(S[0])
print("registers read before written (value-dependent):", ["%s%d" % r for r in found], flush=True)
g.close()
