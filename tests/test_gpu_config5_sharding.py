"""BASELINE.json config 5 on the HIP path, with the one GPU a test box has: DartHopper-v1, 524 288 envs sharded as 8 x 65 536.

north_star: "Envs shard trivially across the 8 GPUs of one node (independent batches ...)".  The sharding rule is the one of
`SyncVectorEnv.seed` (reference gym/vector/sync_vector_env.py:50-58: env i is seeded s + i): rank g owns the global envs
[g n, (g + 1) n) and its handle is configured with DART_CFG_ENV_OFFSET = g n, so the Philox reset streams are keyed by GLOBAL env
index.  Claim (DESIGN.md sections 4, 7): the union of the 8 shards IS the one big batch, bit for bit -- states, observations,
rewards, done flags, episode and TimeLimit counters -- including across on-device auto-resets.  Here the 8 shards are 8 handles
on cuda:0 stepping the action slices bench.py's rank g would see; what a real 8-GPU node adds is only that they run
concurrently on 8 devices (bench.py --gpus 8, the driver's SCALE run)."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for

pytestmark = pytest.mark.gpu

SHARDS, PER = 8, 65536


@pytest.mark.parametrize("precision", [64, 32])
def test_union_of_eight_shards_equals_one_big_batch_bitwise(precision):
    from dart_env_amd import stepper as st
    card = card_for("DartHopper-v1")
    n_all, steps, seed = SHARDS * PER, 12, 5
    rng = np.random.RandomState(77)
    acts = rng.uniform(-1, 1, (steps, n_all, card.act_dim)).astype(np.float32)

    def run(n, offset):
        s = st.HipStepper(card, n, precision=precision)
        s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, seed); s.configure(st.CFG_ENV_OFFSET, offset)
        obs0 = s.reset(None, None, None)
        outs = [obs0.copy()]
        for t in range(steps):
            o, r, d, tr = s.step(acts[t, offset:offset + n])
            outs += [o.copy(), r.copy(), d.copy(), tr.copy()]
        q, dq = s.get_state(); el, ep = s.counters()
        s.close()
        return outs + [q, dq, el, ep]

    big = run(n_all, 0)
    parts = [run(PER, g * PER) for g in range(SHARDS)]
    for k, whole in enumerate(big):
        union = np.concatenate([p[k] for p in parts], axis=0)
        assert union.shape == whole.shape and union.dtype == whole.dtype
        assert np.array_equal(union, whole, equal_nan=True), "output %d differs between the shards and the single batch" % k
    done_total = sum(int(big[3 + 4 * t].sum()) for t in range(steps))
    ep = big[-1]
    assert done_total > n_all and ep.max() >= 3      # the window spans several on-device resets per env
    # shards really are different envs: shard 1's first env is not shard 0's (different Philox stream)
    assert not np.array_equal(parts[0][0][0], parts[1][0][0])
