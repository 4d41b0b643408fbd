#!/usr/bin/env python3
"""Dry run of the real-DART capture pipeline WITHOUT DART (test infrastructure, build container only).

tools/capture_dart_golden.py is the script that, on a machine with pydart2, turns row (c) of SURVEY.md section 8 from "parity
unpinned" into pinned.  No such machine exists for this build, so the script itself could rot unnoticed.  This driver runs it -- the
very same `main()` -- against the stub `pydart2` of tests/golden/make_golden.py (the reference's own Python on top of this repo's fp64
oracle): the fixtures it writes must satisfy tests/test_dart_real_fixtures.py trivially, and with one DART-semantics knob of SURVEY.md
Appendix C flipped IN THE CAPTURED WORLD (--knob) the same test must fail -- which is what it will do the day a real capture
disagrees with the oracle.

    python tests/golden/capture_with_stub.py --out DIR [--knob A3|A9] [--envs ...] [--steps K] [--seeds ...]

Needs /root/reference (imports the reference's gym package); never runs on the GPU box."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--knob", default="", choices=["", "A3", "A9"])
    ap.add_argument("--envs", nargs="*", default=["DartHopper-v1"])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--seeds", type=int, nargs="*", default=[0])
    args = ap.parse_args()
    import make_golden as mg
    if args.knob:
        real = mg.OracleWorld

        def knobbed(card):          # the "DART" being captured differs from the oracle's default in exactly one assumption
            if args.knob == "A3":
                card.impulse_inertia = 1 - int(card.impulse_inertia)          # impulse pass on M  <->  on M + dt D + dt^2 K
            else:
                card.max_erv, card.contact_cfm = 10.0, 1e-9                   # ContactConstraint constants = the joint-limit values
            return real(card)
        mg.OracleWorld = knobbed
    mg.install_stubs()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import capture_dart_golden as cap
    cap.main(["--out", args.out, "--steps", str(args.steps), "--seeds"] + [str(s) for s in args.seeds] + ["--envs"] + args.envs)


if __name__ == "__main__":
    main()
