#!/usr/bin/env python3
"""Record golden rollouts from the REAL reference stack (gym + pydart2 + DART) -- run this wherever pydart2 imports.

SURVEY.md 8(c), last row: the oracle of this repo is "parity unpinned" because DART / pydart2 exist neither in the
reference tree nor in the build image.  On a machine that has them:

    PYTHONPATH=/path/to/dart-env python tools/capture_dart_golden.py [--out tests/golden/dart_real] [--steps 1000]

writes one `<env-tag>_seed<k>.npz` per env id with exactly the fields `tests/golden/make_golden.py` produces
(obs0, actions (float32), obs, reward, done, truncated, q, dq, reset_obs) plus the versions of pydart2 / numpy used.
Commit the files: `tests/test_dart_real_fixtures.py` then checks the oracle (CPU) and the HIP kernels (`-m gpu`) against
them, which upgrades row (c) from "unpinned" to pinned and settles the knobs of SURVEY.md Appendix C.  Nothing here is
imported by the product.  This script is the build's own code: it calls the reference only through `gym.make`."""
import argparse
import os
import sys

import numpy as np

ENVS = {"DartHopper-v1": "hopper", "DartWalker2d-v1": "walker2d", "DartWalker3d-v1": "walker3d",
        "DartHumanWalker-v1": "humanwalker", "DartHalfCheetah-v1": "halfcheetah", "DartCartPole-v1": "cartpole",
        "DartCartPoleSwingUp-v1": "swingup", "DartDoubleInvertedPendulumEnv-v1": "doublependulum",
        "DartSnake7Link-v1": "snake", "DartReacher-v1": "reacher2d", "DartReacher3d-v1": "reacher3d",
        "DartWalker3dSPD-v1": "walker3dspd", "DartDog-v1": "dog"}


def rollout(gym, env_id, seed, steps, act_scale):
    env = gym.make(env_id)
    env.seed(seed)
    env.action_space.seed(seed + 1000)
    rec = dict(actions=[], obs=[], reward=[], done=[], truncated=[], q=[], dq=[], reset_obs=[], ncontacts=[])
    rec["obs0"] = env.reset()
    sv0 = env.unwrapped.state_vector()
    rec["q0"], rec["dq0"] = sv0[:len(sv0) // 2], sv0[len(sv0) // 2:]
    for _ in range(steps):
        a32 = (env.action_space.sample() * act_scale).astype(np.float32)
        ob, r, d, info = env.step(a32.astype(np.float64))      # float64 view of the float32 sample (numpy-version independent)
        sv = env.unwrapped.state_vector()
        n = len(sv) // 2
        rec["actions"].append(a32); rec["obs"].append(ob); rec["reward"].append(r); rec["done"].append(d)
        rec["truncated"].append(bool(info.get("TimeLimit.truncated", False)))
        rec["q"].append(sv[:n]); rec["dq"].append(sv[n:])
        try:
            rec["ncontacts"].append(len(env.unwrapped.dart_world.collision_result.contacts))
        except Exception:
            rec["ncontacts"].append(-1)
        rec["reset_obs"].append(env.reset() if d else np.full_like(ob, np.nan))
    out = {k: np.asarray(v) for k, v in rec.items()}
    env.close()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "dart_real"))
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--seeds", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--envs", nargs="*", default=list(ENVS))
    args = ap.parse_args(argv)
    try:
        import pydart2
    except ImportError:
        sys.exit("pydart2 is not importable here: this script needs the real DART stack (see the docstring)")
    import gym
    os.makedirs(args.out, exist_ok=True)
    for env_id in args.envs:
        for seed in args.seeds:
            for scale, tag in ((1.0, ""), (0.2, "_small")):     # small actions: long episodes, gentle contacts
                d = rollout(gym, env_id, seed, args.steps, scale)
                d["pydart2_version"] = np.array(getattr(pydart2, "__version__", "unknown"))
                d["numpy_version"] = np.array(np.__version__)
                d["env_id"] = np.array(env_id); d["act_scale"] = np.array(scale)
                path = os.path.join(args.out, "%s_seed%d%s.npz" % (ENVS[env_id], seed, tag))
                np.savez_compressed(path, **d)
                print("wrote", path, "episodes:", int(d["done"].sum()))


if __name__ == "__main__":
    main()
