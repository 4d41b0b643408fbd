#!/usr/bin/env python3
"""How far does each DART-semantic assumption move a trajectory?  (DESIGN.md section 2 table.)

The physics of `World.step()` is pinned by nothing from DART (no pydart2 anywhere), so every assumption the restatement makes is
a knob.  This script flips one knob at a time on the fp64 oracle and reports the RMS difference of q and dq against the default
setting after 1, 10 and 100 env-steps, over the envs that are still inside their first episode in BOTH runs (same reset noise,
same actions; small actions 0.05 U[-1,1) keep the episodes long enough to reach 100 steps).  A future capture from real DART
(tools/capture_dart_golden.py) that disagrees with the oracle should be compared with these signatures first.

  A1  body inertia from the first shape: ignoring the shape's local transform (default) vs honouring it   (model compiler)
  A3  impulse pass on M (default, DART 6) vs on M + dt D + dt^2 K                                          (card.impulse_inertia)
  A5  capsule contact point: ODE sphere-sphere midpoint (default) vs on the capsule surface               (oracle only)
  A7  friction bounds fixed from the frictionless solve (default, ODE's driver) vs iterated to a consistent pyramid (oracle only)
  A9  ContactConstraint constants MAX_ERV 1e-3 / CFM 1e-5 (default) vs the joint-limit values 10 / 1e-9   (card.max_erv, contact_cfm)

Needs /root/reference for A1 (re-parses the .skel); runs on the CPU only.   python tools/knob_sensitivity.py [envs]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dart_env_amd.model_card import TASKS, build_card, card_for  # noqa: E402
from dart_env_amd.skel import parse_skel  # noqa: E402
from tests.oracle_lib import OracleWorld  # noqa: E402

REF = os.environ.get("DART_REFERENCE", "/root/reference")
ASSETS = {"DartHopper-v1": "hopper_capsule.skel", "DartWalker2d-v1": "walker2d.skel", "DartHumanWalker-v1": "kima/kima_human_edited.skel"}
SNAPS = (1, 10, 30, 100)


def card_a1(env_id):
    task = TASKS[env_id]
    path = os.path.join(REF, "gym/envs/dart/assets", ASSETS[env_id])
    if not os.path.exists(path):
        return None
    m = parse_skel(path, dt=0.002, inertia_ignores_shape_transform=False)
    for s in m.shapes:
        s.collidable = True
    return build_card(m, task)


def variants(env_id):
    base = card_for(env_id)
    out = [("default", base, {})]
    c = card_a1(env_id)
    if c is not None:
        out.append(("A1 inertia honours the shape transform", c, {}))
    c = card_for(env_id); c.impulse_inertia = 1
    out.append(("A3 impulses on M + dt D + dt^2 K", c, {}))
    out.append(("A5 contact point on the capsule surface", card_for(env_id), {5: 1}))
    out.append(("A7 friction bounds iterated to consistency", card_for(env_id), {7: 1}))
    c = card_for(env_id); c.max_erv = 10.0; c.contact_cfm = 1e-9
    out.append(("A9 contact MAX_ERV 10, CFM 1e-9", c, {}))
    return out


def run(card, assume, n, qn, vn, acts):
    steps = acts.shape[0]
    q = np.full((steps, n, card.ndofs), np.nan); dq = np.full((steps, n, card.ndofs), np.nan)
    for i in range(n):
        w = OracleWorld(card)
        for k, v in assume.items():
            w.set_assumption(k, v)
        w.reset()
        q0, d0 = w.get_state()
        w.set_state(q0 + qn[i], d0 + vn[i]); w.env_after_reset()
        for t in range(steps):
            _, _, done = w.env_step(acts[t, i].astype(np.float64))
            q[t, i], dq[t, i] = w.get_state()
            if done:
                break      # first episode only: later rows stay NaN
    return q, dq


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for env_id in ("DartHopper-v1", "DartWalker2d-v1", "DartHumanWalker-v1"):
        vs = variants(env_id)
        base = vs[0][1]
        rng = np.random.RandomState(7)
        ne = n if env_id != "DartHumanWalker-v1" else max(8, n // 4)
        qn = rng.uniform(-1, 1, (ne, base.ndofs)) * base.reset_noise
        vn = rng.uniform(-1, 1, (ne, base.ndofs)) * base.reset_noise_vel
        acts = (0.05 * rng.uniform(-1, 1, (max(SNAPS), ne, base.act_dim))).astype(np.float32)
        q0, d0 = run(base, {}, ne, qn, vn, acts)
        print("%s  (%d envs, actions 0.05 U[-1,1), RMS over the envs alive in both runs)" % (env_id, ne))
        print("| knob flipped | " + " | ".join("step %d: RMS dq (q) / RMS d(dq) [alive]" % s for s in SNAPS) + " |")
        print("|---|" + "---|" * len(SNAPS))
        for name, card, assume in vs[1:]:
            q1, d1 = run(card, assume, ne, qn, vn, acts)
            cells = []
            for s_ in SNAPS:
                a = np.isfinite(q0[s_ - 1]).all(1) & np.isfinite(q1[s_ - 1]).all(1)
                if not a.any():
                    cells.append("- [0]")
                    continue
                eq = np.sqrt(np.mean((q1[s_ - 1][a] - q0[s_ - 1][a]) ** 2)); ev = np.sqrt(np.mean((d1[s_ - 1][a] - d0[s_ - 1][a]) ** 2))
                cells.append("%.1e / %.1e [%d]" % (eq, ev, int(a.sum())))
            print("| %s | %s |" % (name, " | ".join(cells)))
        print()


if __name__ == "__main__":
    main()
