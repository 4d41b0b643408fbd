#!/usr/bin/env python3
"""Compile the reference's .skel assets into this package's own model cards.

Run HERE (the only place /root/reference exists):
    python tools/compile_models.py [/root/reference]
Writes dart_env_amd/models/<name>.json -- plain numbers in our own format
(masses, frames, axes, limits, shape primitives), nothing of the XML text.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dart_env_amd.skel import parse_skel  # noqa: E402

ASSETS = {
    "hopper": "hopper_capsule.skel",        # hopper.py:13
    "walker2d": "walker2d.skel",            # walker2d.py:12
    "walker3d": "walker3d_waist.skel",      # walker3d.py:18
    "humanwalker": "kima/kima_human_edited.skel",  # human_walker.py
    "cartpole": "cartpole.skel",            # cart_pole.py:9 (dt 0.02)
    "halfcheetah": "half_cheetah.skel",     # half_cheetah.py:18 (dt 0.01)
    "cartpole_swingup": "cartpole_swingup.skel",        # cartpole_swingup.py:11 (dt 0.01)
    "double_pendulum": "inverted_double_pendulum.skel",  # inverted_double_pendulum.py:14 (dt 0.01)
    "snake7link": "snake_7link.skel",                    # snake_7link.py:18
    "reacher2d": "reacher2d.skel",                       # reacher2d.py:10 (dt 0.01)
    "reacher3d": "reacher.skel",                         # reacher.py:10
    "dog": "dog.skel",                                   # dog.py:14 (free root joint)
}
DT = {"reacher2d": 0.01, "cartpole": 0.02, "halfcheetah": 0.01, "cartpole_swingup": 0.01, "double_pendulum": 0.01}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dart_env_amd", "models")
    os.makedirs(out, exist_ok=True)
    for name, rel in ASSETS.items():
        card = parse_skel(os.path.join(ref, "gym/envs/dart/assets", rel), dt=DT.get(name, 0.002))
        with open(os.path.join(out, name + ".json"), "w") as f:
            f.write(card.to_json())
        print("%-12s ndofs=%2d bodies=%2d mass=%.8f ground_y=%g" % (name, card.ndofs, card.nbodies, card.total_mass, card.ground_y))


if __name__ == "__main__":
    main()
