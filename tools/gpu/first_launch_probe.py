#!/usr/bin/env python3
"""First launch != later launches: ONE case per fresh process, with the three kinds of storage a kernel can read
before writing -- scratch, LDS, vector / accumulator registers -- poisoned on demand before the first and / or the later rollouts.

    python tools/gpu/first_launch_probe.py --env DartHalfCheetah-v1 --prec 32 [--report] [--n 256]
           [--poison none|scratch|lds|regs|all] [--pattern 0x7fc00000] [--when first|later|both|combined]
           [--phys hopper|walker2d|halfcheetah|snake7link|cartpole|double_pendulum|reacher2d|reacher3d]   a physics-only card of that model
                                                                                                          (envs.DartEnv on a user's .skel) instead of --env
    [--autoreset philox|mt|mt-split]   (round 6) resets inside the rollout: the step kernel's own epilogue (Philox noise, or the env's MT19937 stream:
                                       mt19937_draw.hpp), or the two launches behind it (mt-split); a TimeLimit of 2 steps on top of the task's own
                                       terminations, a fresh handle with the same seeds for every rollout
    --when combined (round 6): the first two rollouts undisturbed, every later one behind a fresh poisoning -- one process answers both
    "first launch == later launches" and "nothing read before it is written"
    DART_STEPPER_LIB=abtest/lib_ctab.so ...   the build with compile-time ancestor tables that failed in round 4

Prints one line: the digest of each of four identical rollouts (same state, same actions).  Reading it:
  * later rollouts agree with each other but not with the first, no poison      -> the symptom;
  * poisoning X before the LATER rollouts changes them (or makes them agree with a first one that was poisoned the same way)
    -> the kernel reads X before writing it; if no X does, the difference is not uninitialised storage."""
import argparse, ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import card_for
from dart_env_amd.stepper import HipStepper, CFG_CONTACT_REPORT, CFG_AUTORESET, CFG_SEED, CFG_HOST_DMA

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="DartHalfCheetah-v1"); ap.add_argument("--prec", type=int, default=32); ap.add_argument("--report", action="store_true")
ap.add_argument("--n", type=int, default=256); ap.add_argument("--poison", default="none"); ap.add_argument("--pattern", default="0x7fc00000")
ap.add_argument("--when", default="later"); ap.add_argument("--reps", type=int, default=4); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--phys", default=""); ap.add_argument("--autoreset", default="", choices=["", "philox", "mt", "mt-split"])
ap.add_argument("--fresh", action="store_true", help="a new handle for every rollout: tasks that carry more than (q, dq) from step to step -- "
                "DartWalker3dSPD-v1 feeds the previous step's constraint forces into its controller, walker3d_spd.py:40-55 -- start equal only then")
a = ap.parse_args()
H = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "gpu_kernels", "libpoison_harness.so"))
_prng = __import__("random").Random(12345)


def next_pattern():       # "random": a fresh 32-bit word for every poisoning -- leftovers that differ from launch to launch, under control
    return _prng.getrandbits(32) | 1 if a.pattern == "random" else int(a.pattern, 0)


def poison():
    kinds = ("scratch", "lds", "regs") if a.poison == "all" else (a.poison,)
    for k in kinds:
        if k != "none":
            rc = getattr(H, "poison_" + k)(C.c_uint32(next_pattern()))
            assert rc == 0, (k, rc)


if a.phys:
    from dart_env_amd.model_card import build_card, load_model
    card = build_card(load_model(a.phys), None); card.frame_skip = 4
    a.env = "phys:" + a.phys
else:
    card = card_for(a.env)
n = a.n; nd, na = card.ndofs, card.act_dim
rng = np.random.RandomState(5)
q0 = rng.uniform(-0.3, 0.3, (n, nd)); dq0 = rng.uniform(-2, 2, (n, nd))
if card.ground_y > -1e9:
    q0[:, 1] = rng.uniform(-0.65, -0.3, n)
acts = rng.uniform(-1, 1, (a.steps, n, na)).astype(np.float32)
if a.phys:
    acts *= 20.0                      # generalized forces, not normalised actions
    acts[:, :, :min(3, na - 1)] = 0   # (nothing pushes the root / the cart)
if a.autoreset:
    card.max_episode_steps = 2
    a.fresh = True                    # (episode counters and generator positions are part of the start)


def make():
    g = HipStepper(card, n, precision=a.prec)
    if a.report:
        g.configure(CFG_CONTACT_REPORT, 1)
    if a.autoreset:
        g.configure(CFG_AUTORESET, 1); g.configure(CFG_SEED, 7)
        if a.autoreset != "philox":
            from dart_env_amd import seeding
            keys, klen = seeding.mt_keys(list(range(11, 11 + n)))
            g.seed_mt19937(keys, klen)
            g.configure(CFG_HOST_DMA, 3 | (8 if a.autoreset == "mt-split" else 0))
    return g


g = make()
digests, outs_all = [], []
for rep in range(a.reps):
    if a.fresh and rep > 0:
        g.close()
        g = make()
    g.set_state(q0, dq0)
    if (rep == 0 and a.when in ("first", "both")) or (rep > 0 and a.when in ("later", "both")) or (rep > 1 and a.when == "combined"):
        poison()
    outs = []
    for t in range(a.steps):
        ob, r, d, tr = g.step(acts[t]); outs += [ob.copy(), r.copy(), d.copy()]
    outs += list(g.get_state())
    if a.autoreset:
        outs.append(g.snapshot())     # counters, generator words and positions
    h = hashlib.sha1()
    for o in outs:
        h.update(np.ascontiguousarray(o).tobytes())
    digests.append(h.hexdigest()[:10]); outs_all.append(outs)
q_first, q_later = outs_all[0][-2], outs_all[-1][-2]
ndiff = int((~((q_first == q_later) | (np.isnan(q_first) & np.isnan(q_later)))).any(axis=1).sum())
print("%-22s f%d %-6s%s lib=%-18s poison=%-7s when=%-5s pattern=%s | digests %s | envs whose final q differs first vs last: %d of %d" % (
    a.env, a.prec, "report" if a.report else "lean", (" autoreset=" + a.autoreset) if a.autoreset else "", os.path.basename(os.environ.get("DART_STEPPER_LIB", "in-tree")), a.poison, a.when, a.pattern,
    " ".join(digests), ndiff, n), flush=True)
g.close()
