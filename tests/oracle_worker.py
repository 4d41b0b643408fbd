"""One process of the all-cores CPU baseline (TEST INFRASTRUCTURE; started by bench.py's cpu_baseline leg only):
   python -m tests.oracle_worker <env_id> <all_bodies_collide 0|1> <env_offset> <n_envs> <steps> <act_dim>
rolls envs [env_offset, env_offset + n_envs) of the bench sample (same Philox streams and the same action tensor as the
single-core leg) on the fp64 oracle and prints `env_steps seconds`."""
import sys
import time

import numpy as np


def main():
    env_id, allc, off, n, steps, act = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    from dart_env_amd.model_card import card_for
    from tests import oracle_lib as ol
    card = card_for(env_id, all_bodies_collide=bool(allc))
    total = int(sys.argv[7])
    acts = np.random.RandomState(7).uniform(-1, 1, (steps, total, act)).astype(np.float32)[:, off:off + n]
    ol.lib()
    t0 = time.perf_counter()
    ref = ol.rollout(card, acts, seed=0, env_offset=off, solver=0)
    dt = time.perf_counter() - t0
    print(int(ref["env_steps"]), dt, flush=True)


if __name__ == "__main__":
    main()
