"""GPU tests through the C ABI: (1) the committed golden fixtures (reference Python task logic on oracle physics),
(2) the gym.vector surface on the device, (3) size-independent properties at BASELINE's full batch (65 536)."""
import os

import numpy as np
import pytest

import dart_env_amd
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
from dart_env_amd.envs import DartHopperEnv, DartWalker2dEnv
from dart_env_amd.wrappers import TimeLimit

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IDS = {"hopper": "DartHopper-v1", "walker2d": "DartWalker2d-v1"}
CLS = {"hopper": DartHopperEnv, "walker2d": DartWalker2dEnv}


@pytest.mark.parametrize("tag", ["hopper", "walker2d"])
def test_golden_single_env_fp64_kernel(tag):
    """fp64 kernel vs the reference-generated fixture: obs/reward to float32 rounding, done flags exact, full q/dq."""
    d = np.load(os.path.join(G, "%s_single_seed0.npz" % tag))
    env = TimeLimit(CLS[tag](precision=64), max_episode_steps=1000)
    env.seed(0)
    assert np.allclose(env.reset(), d["obs0"], atol=1e-7)
    for t in range(300):
        ob, r, done, info = env.step(d["actions"][t])
        assert done == bool(d["done"][t]), t
        assert np.allclose(ob, d["obs"][t], rtol=0, atol=5e-6), (t, np.abs(ob - d["obs"][t]).max())
        assert abs(r - d["reward"][t]) < 1e-4
        sv = env.state_vector()
        assert np.allclose(sv, np.concatenate([d["q"][t], d["dq"][t]]), rtol=0, atol=1e-8)
        if done:
            assert np.allclose(env.reset(), d["reset_obs"][t], atol=1e-7)
    env.close()


@pytest.mark.parametrize("tag", ["hopper", "walker2d"])
def test_golden_long_episodes_fp64_kernel(tag):
    """1 100 steps of small-action episodes (40-100 steps each): RMS state error of the fp64 kernel stays < 1e-8."""
    d = np.load(os.path.join(G, "%s_single_seed5_small.npz" % tag))
    env = CLS[tag](precision=64)
    env.seed(5)
    env.reset()
    err = []
    for t in range(len(d["done"])):
        ob, r, done, info = env.step(d["actions"][t])
        assert done == bool(d["done"][t]), t
        err.append(env.state_vector() - np.concatenate([d["q"][t], d["dq"][t]]))
        if done:
            env.reset()
    rms = np.sqrt(np.mean(np.square(err)))
    assert rms < 1e-8, rms
    env.close()


@pytest.mark.parametrize("tag", ["hopper", "walker2d"])
def test_golden_vector_env_product_default_fp64(tag):
    """The product default (`vector.make` = fp64 kernels, device MT19937 resets) against the SyncVectorEnv fixture of the reference's
    Python: EVERY step, done flags exact (so the per-env reset streams never desynchronise), obs to float32 rounding, rewards 1e-4."""
    d = np.load(os.path.join(G, "%s_vector4_seed3.npz" % tag))
    venv = dart_env_amd.vector.make(IDS[tag], 4)
    assert venv.precision == 64
    venv.seed(3)
    ob = venv.reset()
    assert ob.dtype == np.float32 and np.allclose(ob, d["obs0"], atol=1e-6)
    assert d["done"].any()                               # the fixture does exercise SyncVectorEnv's auto-reset
    for t in range(len(d["done"])):
        ob, r, done, infos = venv.step(d["actions"][t])
        assert ob.dtype == np.float32 and r.dtype == np.float64 and done.dtype == np.bool_
        assert np.array_equal(done, d["done"][t]), t
        assert np.allclose(ob, d["obs"][t], rtol=0, atol=5e-6), (t, np.abs(ob - d["obs"][t]).max())
        assert np.allclose(r, d["reward"][t], rtol=0, atol=1e-4), (t, np.abs(r - d["reward"][t]).max())
    venv.close()


@pytest.mark.parametrize("tag", ["hopper", "walker2d"])
def test_golden_vector_env_fp32_fast_mode(tag):
    """The explicit fast mode (`precision=32`) on the same fixture, held to what fp32 delivers: until the first done flag that fp32
    rounding flips (which desynchronises that env's reset stream) everything matches to 2e-2, and that is at least 60 steps."""
    d = np.load(os.path.join(G, "%s_vector4_seed3.npz" % tag))
    venv = dart_env_amd.vector.make(IDS[tag], 4, precision=32)
    assert venv.precision == 32
    venv.seed(3)
    ob = venv.reset()
    assert ob.dtype == np.float32 and np.allclose(ob, d["obs0"], atol=1e-6)
    agree = 0
    for t in range(len(d["done"])):
        ob, r, done, infos = venv.step(d["actions"][t])
        if not np.array_equal(done, d["done"][t]):
            break  # an fp32 done flip desynchronises the RNG streams; everything before must match
        agree += 1
        assert np.allclose(ob, d["obs"][t], rtol=0, atol=2e-2) and np.allclose(r, d["reward"][t], atol=2e-2)
    assert agree >= 60, agree
    venv.close()


def test_time_limit_on_device():
    d = np.load(os.path.join(G, "hopper_single_seed2_limit20.npz"))
    card = card_for("DartHopper-v1")
    card.max_episode_steps = 20
    s = st.HipStepper(card, 1, precision=64)
    from dart_env_amd import seeding
    rng, _ = seeding.np_random(2)
    s.reset(None, rng.uniform(-.005, .005, (1, 6)), rng.uniform(-.005, .005, (1, 6)))
    for t in range(len(d["done"])):
        ob, r, done, trunc = s.step(d["actions"][t][None])
        assert bool(done[0]) == bool(d["done"][t]) and bool(trunc[0]) == bool(d["truncated"][t])
        if done[0]:
            s.reset(None, rng.uniform(-.005, .005, (1, 6)), rng.uniform(-.005, .005, (1, 6)))
    s.close()


# ---------------------------------------------------------------- size-independent properties at N = 65 536
N_FULL = 65536


def _run(n, steps, block=64, precision=32, env_id="DartHopper-v1", seed=9, x_shift=0.0, solver=None):
    card = card_for(env_id)
    s = st.HipStepper(card, n, precision=precision)
    s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, seed); s.configure(st.CFG_BLOCK_THREADS, block)
    s.reset(None, None, None, want_obs=False)
    if x_shift:
        q, dq = s.get_state(); q[:, 0] += x_shift; s.set_state(q, dq)
    rng = np.random.RandomState(1)
    outs = []
    for t in range(steps):
        a = rng.uniform(-1, 1, (N_FULL, card.act_dim)).astype(np.float32)[:n]
        outs.append(s.step(a))
    q, dq = s.get_state()
    el, ep = s.counters()
    s.close()
    return outs, q, dq, el, ep


@pytest.mark.parametrize("precision", [64, 32])
def test_full_batch_determinism_and_batch_independence(precision):
    """Run-to-run bitwise determinism at N=65 536, and env i's trajectory is independent of the batch it sits in
    (wave-level votes only end loops early, they never change a lane's result) and of the workgroup width.  precision=64 is the
    product default -- the kernel bench.py times; 32 the fast mode."""
    import functools
    _run = functools.partial(globals()["_run"], precision=precision)
    o1, q1, dq1, el1, ep1 = _run(N_FULL, 12)
    o2, q2, dq2, el2, ep2 = _run(N_FULL, 12)
    assert np.array_equal(q1, q2) and np.array_equal(dq1, dq2) and np.array_equal(ep1, ep2)
    for a, b in zip(o1, o2):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    o3, q3, dq3, el3, ep3 = _run(1000, 12)              # ragged: not a multiple of 64
    assert np.array_equal(q1[:1000], q3) and np.array_equal(dq1[:1000], dq3) and np.array_equal(ep1[:1000], ep3)
    o4, q4, dq4, el4, ep4 = _run(1000, 12, block=32)
    assert np.array_equal(q3, q4) and np.array_equal(dq3, dq4)
    assert np.isfinite(q1).all() and np.isfinite(dq1).all()
    assert ep1.max() > 1 and ep1.min() >= 1              # episodes end and restart on device


@pytest.mark.parametrize("precision", [64, 32])
def test_full_batch_outputs_are_consistent(precision):
    """obs/reward/done at N=65 536 obey the task definition: obs = [height, q[2:], clip(dq)], done envs restart (64 = the product
    default and the kernel bench.py times, 32 = the fast mode)."""
    card = card_for("DartHopper-v1")
    s = st.HipStepper(card, N_FULL, precision=precision)
    s.configure(st.CFG_SEED, 3)
    s.reset(None, None, None, want_obs=False)
    rng = np.random.RandomState(2)
    for t in range(6):
        a = rng.uniform(-1.5, 1.5, (N_FULL, 3)).astype(np.float32)
        q0, _ = s.get_state()
        ob, r, done, trunc = s.step(a)
        q, dq = s.get_state()                            # no auto-reset configured: state is the post-step state
        assert np.allclose(ob[:, 0], 1.25 + q[:, 1], atol=1e-5)
        assert np.allclose(ob[:, 1:5], q[:, 2:], atol=1e-5)
        assert np.allclose(ob[:, 5:], np.clip(dq, -10, 10), atol=1e-4)
        pen = 0.75 * ((card.lower[4] - q[:, 4]) > -0.05) + 0.75 * ((card.upper[4] - q[:, 4]) < 0.05)
        rew = (q[:, 0] - q0[:, 0]) / 0.008 + 1.0 - 1e-3 * np.sum(a.astype(np.float64) ** 2, axis=1) - pen
        assert np.allclose(r, rew, atol=2e-2 if precision == 32 else 2e-4)   # fp32: x-difference / 0.008 in single precision; the reward leaves the device as float32
        ok = np.isfinite(q).all(1) & np.isfinite(dq).all(1) & (np.abs(q[:, 2:]) < 100).all(1) & \
            (np.abs(dq) < 100).all(1) & (ob[:, 0] > .7) & (ob[:, 0] < 1.8) & (np.abs(q[:, 2]) < .2)
        border = (np.abs(ob[:, 0] - .7) < 1e-5) | (np.abs(np.abs(q[:, 2]) - .2) < 1e-5)
        assert np.array_equal(done[~border], ~ok[~border])
        if done.any():
            s.reset(done.astype(np.uint8), None, None, want_obs=False)
    s.close()


def test_translation_invariance_in_x():
    """Dynamics do not depend on the root x coordinate: shifting every env by +3 m changes only q[0]."""
    oa, qa, dqa, _, _ = _run(4096, 5, precision=64)
    ob, qb, dqb, _, _ = _run(4096, 5, precision=64, x_shift=3.0)
    same = np.ones(4096, dtype=bool)
    for (o1, r1, d1, t1), (o2, r2, d2, t2) in zip(oa, ob):
        same &= d1 == d2
    assert same.mean() > 0.999
    # envs that auto-reset lose the shift, the others keep it exactly
    moved = np.abs((qb[:, 0] - qa[:, 0]) - 3.0) < 1e-9
    assert (moved | (np.abs(qb[:, 0] - qa[:, 0]) < 1e-9)).all()
    assert np.allclose(qa[:, 1:], qb[:, 1:], atol=1e-9) and np.allclose(dqa, dqb, atol=1e-8)


def test_device_philox_autoreset_matches_oracle_rollout():
    """On-device auto-reset (Philox) against the oracle's rollout with the same counter-based streams."""
    from tests import oracle_lib as ol
    card = card_for("DartHopper-v1")
    n, steps = 512, 30
    acts = np.random.RandomState(4).uniform(-1, 1, (steps, n, 3)).astype(np.float32)
    ref = ol.rollout(card, acts, seed=11, env_offset=1000)
    s = st.HipStepper(card, n, precision=64)
    s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 11); s.configure(st.CFG_ENV_OFFSET, 1000)
    s.reset(None, None, None, want_obs=False)
    for t in range(steps):
        s.step(acts[t])
    q, dq = s.get_state()
    el, ep = s.counters()
    assert np.array_equal(ep, ref["episode"]) and np.array_equal(el, ref["elapsed"])
    assert np.abs(q - ref["q"]).max() < 1e-7 and np.abs(dq - ref["dq"]).max() < 1e-5
    s.close()


def test_async_misuse_errors_on_device():
    venv = dart_env_amd.vector.make("DartHopper-v1", 8, noise="philox")
    venv.reset()
    with pytest.raises(st.NoAsyncCallError):
        venv.step_wait()
    venv.step_async(np.zeros((8, 3), dtype=np.float32))
    with pytest.raises(st.AlreadyPendingCallError):
        venv.step_async(np.zeros((8, 3), dtype=np.float32))
    venv.step_wait()
    venv.close()


def test_pgs_solver_converges_to_pivoting_solver():
    """north star names PGS: with enough sweeps the device PGS reproduces the exact (pivoting) solve."""
    card = card_for("DartHopper-v1")
    n = 2048
    res = {}
    for solver, iters in ((st.SOLVER_BPP, 24), (st.SOLVER_PGS, 3000)):
        s = st.HipStepper(card, n, precision=64)
        s.configure(st.CFG_SOLVER, solver); s.configure(st.CFG_ITERS_STAGE1, iters); s.configure(st.CFG_ITERS_STAGE2, iters)
        s.configure(st.CFG_SEED, 2)
        s.reset(None, None, None, want_obs=False)
        a = np.random.RandomState(0).uniform(-.3, .3, (n, 3)).astype(np.float32)
        for t in range(3):
            s.step(a)
        res[solver] = s.get_state()
        s.close()
    dq_err = np.abs(res[0][1] - res[1][1])
    assert np.percentile(dq_err, 99) < 1e-6 and dq_err.max() < 1e-2


@pytest.mark.parametrize("n", [1, 63, 65, 130])
def test_ragged_batch_sizes(n):
    """Batch sizes that do not fill a wavefront: tail lanes shadow the last env and must not write anything."""
    outs, q, dq, el, ep = _run(n, 6, precision=64)
    ref, qr, dqr, elr, epr = _run(256, 6, precision=64)
    assert np.array_equal(q, qr[:n]) and np.array_equal(dq, dqr[:n]) and np.array_equal(ep, epr[:n])


def test_non_finite_actions_terminate_instead_of_poisoning_neighbours():
    """done = not isfinite(state) (hopper.py:60): a NaN/inf action ends that env only."""
    card = card_for("DartHopper-v1")
    s = st.HipStepper(card, 128, precision=32)
    s.configure(st.CFG_SEED, 1)
    s.reset(None, None, None, want_obs=False)
    a = np.zeros((128, 3), dtype=np.float32)
    a[5, 0] = np.nan
    a[70, 2] = np.inf          # clamps to +1 like any large action (hopper.py:25-30)
    ob, r, done, trunc = s.step(a)
    assert done[5] and not done[6] and not done[4]
    others = np.arange(128) != 5
    assert np.isfinite(ob[others]).all()
    assert r[70] == -np.inf            # control cost uses the unclamped action (hopper.py:55), as in the reference
    assert np.isfinite(r[others & (np.arange(128) != 70)]).all()
    q, dq = s.get_state()
    assert np.isfinite(q[70]).all()
    s.close()


def test_device_gaussian_resets_follow_numpys_legacy_stream():
    """DartDoubleInvertedPendulumEnv-v1 (inverted_double_pendulum.py:50-51): qpos noise uniform, qvel noise `np_random.randn(ndofs) * .1` --
    numpy's legacy polar Gaussian incl. the deviate it caches ACROSS calls.  Round 4 draws it on the device (csrc/mt19937_kernels.hpp,
    MT_EXTRA_GAUSS_VEL; log = dartk::log_cr).  Against numpy's RandomState through 80 masked resets of 256 envs: the positions and the
    uniform consumption bit-exact, every velocity within a few ulps, and >= 99.5 % of them bit-exact (the host libm's log is not correctly
    rounded in 0.08 % of the draws; a device log of the usual 1-ulp quality would miss 2.7 %)."""
    from dart_env_amd import seeding
    card = card_for("DartDoubleInvertedPendulumEnv-v1")
    n, nd = 256, card.ndofs
    s = st.HipStepper(card, n, precision=64)
    keys, klen = seeding.mt_keys(list(range(40, 40 + n)))
    s.seed_mt19937(keys, klen)
    rngs = [seeding.np_random(sd)[0] for sd in range(40, 40 + n)]
    r, rv = card.reset_noise, card.reset_noise_vel
    q0 = np.array([card.init_pos[d] for d in range(nd)]); v0 = np.array([card.init_vel[d] for d in range(nd)])
    rs = np.random.RandomState(1)
    total = exact = 0
    for it in range(80):
        mask = (rs.rand(n) < 0.6).astype(np.uint8) if it else None
        s.reset(mask, None, None, want_obs=False)
        q, dq = s.get_state()
        for i in range(n):
            if mask is None or mask[i]:
                eq = q0 + rngs[i].uniform(low=-r, high=r, size=nd)
                ev = v0 + rngs[i].randn(nd) * rv
                assert np.array_equal(q[i], eq), (it, i)                       # uniforms: bit-exact, and the stream position with them
                assert np.all(np.abs(dq[i] - ev) <= 4 * np.spacing(np.abs(ev))), (it, i, dq[i] - ev)   # (1 ulp in the log -> <= a few ulps after sqrt, two products and the sum)
                total += nd; exact += int((dq[i] == ev).sum())
    assert exact >= 0.995 * total, (exact, total)
    # a snapshot carries the cached deviate: resume draws the same Gaussians
    snap = s.snapshot()
    s.reset(None, None, None, want_obs=False); a = s.get_state()[1].copy()
    s.restore(snap)
    s.reset(None, None, None, want_obs=False); b = s.get_state()[1]
    assert np.array_equal(a, b)
    s.close()


def test_device_mt19937_bank_is_bit_exact_with_numpy_streams():
    """dart_seed_mt19937 + device draws == seeding.np_random(seed).uniform(...) for qpos then qvel, through 60 resets
    of every env (crosses the 624-word regeneration several times), for 1- and 2-word keys."""
    from dart_env_amd import seeding
    card = card_for("DartHopper-v1")
    n = 300
    seeds = list(range(7, 7 + n))
    s = st.HipStepper(card, n, precision=64)
    keys = np.zeros((n, 2), dtype=np.uint32); klen = np.zeros(n, dtype=np.int32)
    rngs = []
    for i, sd in enumerate(seeds):
        w = seeding.int_list_from_bigint(seeding.hash_seed(sd))
        if i == 5:
            w = [w[0]]                      # exercise the single-word key path with a hand-made generator
            r = np.random.RandomState(); r.seed(w)
        else:
            r, _ = seeding.np_random(sd)
        klen[i] = len(w); keys[i, :len(w)] = w
        rngs.append(r)
    s.seed_mt19937(keys, klen)
    rs = np.random.RandomState(0)
    for it in range(60):
        mask = (rs.rand(n) < 0.7).astype(np.uint8) if it else None
        s.reset(mask, None, None, want_obs=False)
        q, dq = s.get_state()
        for i in range(n):
            if mask is None or mask[i]:
                eq = rngs[i].uniform(low=-.005, high=.005, size=6); ev = rngs[i].uniform(low=-.005, high=.005, size=6)
                assert np.array_equal(q[i], eq) and np.array_equal(dq[i], ev), (it, i, q[i] - eq, dq[i] - ev)
    s.close()


@pytest.mark.parametrize("env_id,precision", [("DartHopper-v1", 64), ("DartHopper-v1", 32), ("DartWalker2d-v1", 64), ("DartHalfCheetah-v1", 64),
                                              ("DartSnake7Link-v1", 64)])
def test_mt19937_reset_in_the_step_kernel_equals_the_two_launch_path(env_id, precision):
    """Round 6: a lane kernel draws a finished env's reset noise from the env's MT19937 stream in its own epilogue (mt19937_draw.hpp)
    instead of two more launches behind it (mt_draw_kernel + the masked reset kernel; DART_CFG_HOST_DMA bit 3 keeps those for this A/B).
    Same streams, same order, same roundings: observations, rewards, done flags, states and the generators' positions agree bitwise over
    a rollout in which every env resets several times -- and the fused draws ARE numpy's (hopper: checked against RandomState directly)."""
    from dart_env_amd import seeding
    card = card_for(env_id)
    n, T = 640, 150 if env_id != "DartHalfCheetah-v1" else 60
    card.max_episode_steps = 23          # TimeLimit truncations on top of the task's own terminations: several resets per env
    keys, klen = seeding.mt_keys(list(range(11, 11 + n)))
    acts = np.random.RandomState(4).uniform(-1, 1, (T, n, card.act_dim)).astype(np.float32)
    outs = []
    for dma in (3, 3 | 8):
        g = st.HipStepper(card, n, precision=precision)
        g.seed_mt19937(keys, klen)
        g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_HOST_DMA, dma)
        g.reset(None, None, None, want_obs=False)
        rec = []
        for t in range(T):
            o, r, d, tr = g.step(acts[t])
            rec.append((o.copy(), r.copy(), d.copy(), tr.copy()))
        q, dq = g.get_state()
        el, ep = g.counters()
        outs.append((rec, q, dq, el, g.snapshot()))
        g.close()
    (ra, qa, dqa, ela, sa), (rb, qb, dqb, elb, sb) = outs
    resets = sum(int(x[2].sum()) for x in ra)
    assert resets >= 2 * n, resets       # (the half cheetah never terminates on its own: the 23-step TimeLimit twice in 60 steps)
    for t in range(T):
        for k in range(4):
            assert np.array_equal(ra[t][k], rb[t][k]), (t, k)
    assert np.array_equal(qa, qb) and np.array_equal(dqa, dqb) and np.array_equal(ela, elb)
    assert np.array_equal(sa, sb)        # the whole checkpoint: state, counters, MT19937 words and positions
    if env_id == "DartHopper-v1" and precision == 64:
        # env 0's post-reset states are numpy's: replay its stream
        r0, _ = seeding.np_random(11)
        g = st.HipStepper(card, n, precision=64)
        g.seed_mt19937(keys, klen); g.configure(st.CFG_AUTORESET, 1)
        g.reset(None, None, None, want_obs=False)
        q, dq = g.get_state()
        assert np.array_equal(q[0], card_init(card)[0] + r0.uniform(-.005, .005, 6)) and np.array_equal(dq[0], card_init(card)[1] + r0.uniform(-.005, .005, 6))
        for t in range(60):
            o, r, d, tr = g.step(acts[t])
            if d[0]:
                q, dq = g.get_state()
                eq = card_init(card)[0] + r0.uniform(-.005, .005, 6); ev = card_init(card)[1] + r0.uniform(-.005, .005, 6)
                assert np.array_equal(q[0], eq) and np.array_equal(dq[0], ev), t
        g.close()


def card_init(card):
    nd = card.ndofs
    return np.array(card.init_pos[:nd]), np.array(card.init_vel[:nd])


@pytest.mark.parametrize("tag", ["hopper", "walker2d"])
def test_vector_env_device_mt19937_matches_reference_fixture(tag):
    """Default noise mode: generators in HBM, resets inside dart_step -- still the reference's SyncVectorEnv stream."""
    d = np.load(os.path.join(G, "%s_vector4_seed3.npz" % tag))
    venv = dart_env_amd.vector.make(IDS[tag], 4, precision=64)
    assert venv.env.noise == "mt19937" and venv.env.device_noise
    venv.seed(3)
    ob = venv.reset()
    assert np.allclose(ob, d["obs0"], atol=1e-7)
    for t in range(len(d["done"])):
        ob, r, done, infos = venv.step(d["actions"][t])
        assert np.array_equal(done, d["done"][t]), t
        assert np.allclose(ob, d["obs"][t], rtol=0, atol=5e-6) and np.allclose(r, d["reward"][t], atol=1e-4)
    venv.close()


def test_device_resident_step_with_mt19937_resets_equals_host_buffer_step():
    """dart_step_device (learner-owned HBM buffers, caller's stream) and dart_step (host arrays) run the same three
    launches in MT19937 auto-reset mode: identical states, observations and done flags after 120 steps."""
    import torch
    from dart_env_amd import seeding
    card = card_for("DartWalker2d-v1")
    n = 512
    keys = np.zeros((n, 2), dtype=np.uint32); klen = np.zeros(n, dtype=np.int32)
    for i in range(n):
        w = seeding.int_list_from_bigint(seeding.hash_seed(100 + i))
        klen[i] = len(w); keys[i, :len(w)] = w
    a_host = st.HipStepper(card, n, precision=64); b_dev = st.HipStepper(card, n, precision=64)
    for s in (a_host, b_dev):
        s.configure(st.CFG_AUTORESET, 1); s.seed_mt19937(keys, klen)
    ob_a = a_host.reset(None, None, None)
    d_obs = torch.empty((n, card.obs_dim), dtype=torch.float32, device="cuda")
    d_rew = torch.empty(n, dtype=torch.float32, device="cuda")
    d_done = torch.empty(n, dtype=torch.uint8, device="cuda"); d_tr = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    b_dev.reset_device(0, d_obs.data_ptr(), stream.cuda_stream); stream.synchronize()
    assert np.array_equal(d_obs.cpu().numpy(), ob_a)
    rs = np.random.RandomState(5)
    n_done = 0
    for t in range(120):
        act = rs.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        oa, ra, da, ta = a_host.step(act)
        d_act = torch.from_numpy(act).cuda()
        torch.cuda.current_stream().synchronize()       # the upload ran on torch's stream, the step runs on `stream`
        b_dev.step_device(d_act.data_ptr(), d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_tr.data_ptr(),
                          stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(d_done.cpu().numpy().astype(bool), da) and np.array_equal(d_obs.cpu().numpy(), oa), t
        n_done += int(da.sum())
    assert n_done > n            # every env went through resets on the device
    qa, va = a_host.get_state(); qb, vb = b_dev.get_state()
    assert np.array_equal(qa, qb) and np.array_equal(va, vb)
    a_host.close(); b_dev.close()


def test_device_episode_statistics_match_host_accumulation():
    """DART_CFG_EPISODE_STATS: per-env return / length accumulators in HBM = RecordEpisodeStatistics semantics."""
    from dart_env_amd.wrappers import VectorRecordEpisodeStatistics
    n = 2048
    venv = VectorRecordEpisodeStatistics(dart_env_amd.vector.make("DartHopper-v1", n, noise="philox"))
    assert venv._device
    venv.seed(4)
    venv.reset()
    rs = np.random.RandomState(8)
    acc, cnt = np.zeros(n), np.zeros(n, dtype=np.int64)
    fin = 0; sum_r = 0.0; sum_l = 0
    for t in range(80):
        ob, r, done, infos = venv.step(rs.uniform(-1, 1, (n, 3)).astype(np.float32))
        acc += r; cnt += 1
        for i in done.nonzero()[0]:
            ep = infos[int(i)]["episode"]
            assert ep["r"] == pytest.approx(acc[i], rel=1e-6, abs=1e-4) and ep["l"] == cnt[i]
            sum_r += acc[i]; sum_l += cnt[i]; fin += 1
            acc[i] = 0; cnt[i] = 0
        assert "episode" not in infos[int((~done).nonzero()[0][0])]
    tr, tl, tc = venv.totals(clear=True)
    assert tc == fin and tl == sum_l and tr == pytest.approx(sum_r, rel=1e-6)
    assert venv.totals()[2] == 0 and fin > n
    venv.close()


def test_rollout_buffer_device_resident_equals_host_stepping():
    """RolloutBuffer: dart_step_device writes every slot of the (T, n, ...) HBM tensors; same trajectory as stepping
    through host arrays with the same policy."""
    import torch
    from dart_env_amd.distributed import RolloutBuffer
    n, T = 512, 40
    policy = lambda ob: torch.tanh(ob[:, :3] * 3.0 - ob[:, 5:8])
    venv = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox", precision=64)
    venv.seed(2)
    buf = RolloutBuffer(venv, T)
    assert buf.on_device and buf.obs.is_cuda
    buf.collect(policy); torch.cuda.synchronize()
    ref = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox", precision=64)
    ref.seed(2)
    ob = ref.reset()
    assert np.array_equal(buf.obs[0].cpu().numpy(), ob)
    for t in range(T):
        a = policy(torch.from_numpy(ob)).numpy()
        assert np.allclose(buf.actions[t].cpu().numpy(), a, atol=1e-6)
        ob, r, d, infos = ref.step(buf.actions[t].cpu().numpy())
        assert np.array_equal(buf.dones[t].cpu().numpy().astype(bool), d), t
        assert np.array_equal(buf.obs[t + 1].cpu().numpy(), ob) and np.allclose(buf.rewards[t].cpu().numpy(), r, atol=1e-5)
    assert buf.dones.sum().item() > 0
    g = buf.gather()
    assert g["obs"].shape == (1, T + 1, n, 11)
    last = g["obs"][0, T].clone()
    buf.collect(policy); torch.cuda.synchronize()   # a second window continues from the last observation
    assert torch.equal(buf.obs[0], last)
    venv.close(); ref.close()


def test_rollout_buffer_and_episode_statistics_against_the_oracle():
    """SURVEY 8(f)-4 against the ORACLE, not against the kernel itself: the device-resident RolloutBuffer (dart_step_device writing
    slot t of HBM tensors, policy on the GPU) and the device episode accumulators (DART_CFG_EPISODE_STATS) reproduce what the same
    deterministic policy gives on oracle worlds stepped through host arrays, with RecordEpisodeStatistics' bookkeeping
    (reference gym/wrappers/record_episode_statistics.py:22-34: return += reward, length += 1, latched into info['episode']
    {'r', 'l'} on done, then restarted) done in numpy on the oracle's rewards / done flags."""
    import torch
    from dart_env_amd.distributed import RolloutBuffer
    from tests.fake_stepper import OracleStepper
    n, T = 256, 40
    policy = lambda ob: torch.tanh(ob[:, :3] * 3.0 - ob[:, 5:8])
    venv = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox", precision=64)
    venv.seed(2)
    venv.env._stepper.configure(st.CFG_EPISODE_STATS, 1)
    buf = RolloutBuffer(venv, T)
    assert buf.on_device
    buf.collect(policy); torch.cuda.synchronize()
    last_r, last_l, totals = venv.env._stepper.episode_stats()
    oenv = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox", precision=64, stepper_factory=OracleStepper)
    oenv.seed(2)
    obuf = RolloutBuffer(oenv, T)
    assert not obuf.on_device
    obuf.collect(policy)
    g = {k: getattr(buf, k).cpu().numpy() for k in ("obs", "actions", "rewards", "dones", "truncated")}
    o = {k: getattr(obuf, k).numpy() for k in ("obs", "actions", "rewards", "dones", "truncated")}
    assert np.array_equal(g["dones"], o["dones"]) and np.array_equal(g["truncated"], o["truncated"])
    assert np.abs(g["obs"] - o["obs"]).max() < 1e-4 and np.abs(g["actions"] - o["actions"]).max() < 1e-4
    assert np.abs(g["rewards"] - o["rewards"]).max() < 1e-3
    # RecordEpisodeStatistics on the oracle's trajectory
    acc, cnt = np.zeros(n), np.zeros(n, dtype=np.int64)
    ref_r, ref_l, fin, sum_r, sum_l = np.zeros(n), np.zeros(n, dtype=np.int64), 0, 0.0, 0
    for t in range(T):
        acc += o["rewards"][t]; cnt += 1
        d = o["dones"][t].astype(bool)
        ref_r[d] = acc[d]; ref_l[d] = cnt[d]
        fin += int(d.sum()); sum_r += float(acc[d].sum()); sum_l += int(cnt[d].sum())
        acc[d] = 0; cnt[d] = 0
    assert fin > 50
    assert np.array_equal(last_l, ref_l) and np.allclose(last_r, ref_r, rtol=1e-5, atol=1e-3)
    assert totals[2] == fin and totals[1] == sum_l and totals[0] == pytest.approx(sum_r, rel=1e-5)
    venv.close(); oenv.close()


def test_step_outputs_in_registered_blocks_keep_copy_semantics():
    """dart_step_async_to: a step's outputs land in a page-locked block the caller sees directly (no staging memcpy).  The arrays a
    step returned must stay intact while the caller holds them, however many steps follow (gym.vector's copy=True,
    sync_vector_env.py:83), blocks must be reused once dropped, and the values must be those of the staging path bit for bit."""
    card = card_for("DartHopper-v1")
    n = 512
    acts = np.random.RandomState(3).uniform(-1, 1, (12, n, 3)).astype(np.float32)

    def run(pool):
        s = st.HipStepper(card, n, precision=64)
        s.output_pool = pool          # (False: round 2's staging path -- an attribute, the host layer reads no environment switches)
        s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 4)
        s.reset(None, None, None, want_obs=False)
        kept, copies = [], []
        for t in range(12):
            out = s.step(acts[t])
            kept.append(out); copies.append(tuple(np.array(x, copy=True) for x in out))
            if t >= 6:
                kept.pop(0)            # from here on older results are dropped: their blocks become reusable
        for out, cp in zip(kept, copies[-len(kept):]):
            assert all(np.array_equal(a, b) for a, b in zip(out, cp))     # held results were never overwritten
        nblocks = len(s.__dict__.get("_blocks", []))
        s.close()
        return copies, nblocks

    a, nb_a = run(True)
    b, nb_b = run(False)
    assert nb_b == 0 and 1 <= nb_a <= st.HipStepper._POOL_SETS
    for x, y in zip(a, b):
        assert all(np.array_equal(p, q) and p.dtype == q.dtype and p.shape == q.shape for p, q in zip(x, y))
    assert a[0][1].dtype == np.float64 and a[0][2].dtype == np.bool_ and a[0][0].dtype == np.float32


def test_vector_env_copy_false_returns_views_with_the_same_values():
    """copy=False (sync_vector_env.py:83 semantics): observations are views of the pinned staging buffer."""
    a = np.random.RandomState(1).uniform(-1, 1, (20, 256, 3)).astype(np.float32)
    va = dart_env_amd.vector.make("DartHopper-v1", 256, noise="philox"); vb = dart_env_amd.vector.make("DartHopper-v1", 256, noise="philox", copy=False)
    va.seed(5); vb.seed(5)
    assert np.array_equal(va.reset(), vb.reset())
    prev = prev_a = None
    for t in range(20):
        oa, ra, da, ia = va.step(a[t]); ob, rb, db, ib = vb.step(a[t])
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
        assert not ob.flags.owndata
        if prev is not None:
            assert prev is ob or np.shares_memory(prev, ob)        # copy=False: the same pinned buffer every step
            assert not np.shares_memory(prev_a, oa)                # copy=True: a result the caller still holds is never written again
        prev, prev_a = ob, oa
    va.close(); vb.close()


# ---------------------------------------------------------------- the same properties at BASELINE configs 3 and 4
@pytest.mark.parametrize("env_id,n_full,steps", [("DartWalker2d-v1", 65536, 10), ("DartHumanWalker-v1", 16384, 4),
                                                 ("DartHalfCheetah-v1", 65536, 6), ("DartSnake7Link-v1", 65536, 12),
                                                 ("DartCartPole-v1", 65536, 40), ("DartDoubleInvertedPendulumEnv-v1", 65536, 12),
                                                 ("DartReacher-v1", 65536, 55)])
def test_other_configs_full_batch_determinism_and_batch_independence(env_id, n_full, steps):
    """BASELINE.json configs 3 (DartWalker2d-v1 @ 65 536) and 4 (DartHumanWalker-v1 @ 16 384) at full size, product precision:
    bitwise run-to-run determinism, every env's trajectory independent of the batch around it (ragged sub-batch), finite
    states, and episodes that end and restart on the device.  The same for the lane kernels round 2 added (half cheetah, snake,
    cart-pole family, 2-D reacher)."""
    def run(n):
        card = card_for(env_id)
        s = st.HipStepper(card, n, precision=64)
        if env_id in ("DartReacher-v1", "DartDoubleInvertedPendulumEnv-v1"):
            # reset_model also draws the reach target (reacher2d.py:53-59) / Gaussian velocities (inverted_double_pendulum.py:50-51):
            # on the device that is the MT19937 bank's job, the in-kernel Philox reset would leave the target stale / draw the wrong
            # distribution and the ABI refuses it (test_philox_autoreset_is_refused_...)
            from dart_env_amd import seeding
            s.seed_mt19937(*seeding.mt_keys(list(range(9, 9 + n))))
        s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 9)
        s.reset(None, None, None, want_obs=False)
        rng = np.random.RandomState(1)
        outs = []
        for t in range(steps):
            a = rng.uniform(-1, 1, (n_full, card.act_dim)).astype(np.float32)[:n]
            outs.append(s.step(a))
        q, dq = s.get_state(); el, ep = s.counters()
        s.close()
        return outs, q, dq, el, ep
    o1, q1, dq1, el1, ep1 = run(n_full)
    o2, q2, dq2, el2, ep2 = run(n_full)
    assert np.array_equal(q1, q2) and np.array_equal(dq1, dq2) and np.array_equal(ep1, ep2) and np.array_equal(el1, el2)
    for a, b in zip(o1, o2):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    o3, q3, dq3, el3, ep3 = run(1000)                    # ragged: not a multiple of 64
    assert np.array_equal(q1[:1000], q3) and np.array_equal(dq1[:1000], dq3) and np.array_equal(ep1[:1000], ep3)
    assert np.isfinite(q1).all() and np.isfinite(dq1).all()
    # (the episode counter keys the Philox reset streams; MT19937-bank resets -- the reacher here -- do not advance it)
    assert env_id in ("DartReacher-v1", "DartDoubleInvertedPendulumEnv-v1") or (ep1.min() >= 1 and (ep1.max() > 1 or env_id in ("DartHumanWalker-v1", "DartHalfCheetah-v1", "DartSnake7Link-v1")))
    if env_id == "DartReacher-v1":
        assert sum(int(o[3].sum()) for o in o1) >= n_full          # every env ran into its 50-step TimeLimit


@pytest.mark.parametrize("env_id", ["DartReacher-v1", "DartReacher3d-v1", "DartCartPoleSwingUp-v1", "DartDoubleInvertedPendulumEnv-v1"])
def test_philox_autoreset_is_refused_where_reset_model_draws_more_than_noise(env_id):
    """reacher2d.py:47-60 / reacher.py:44-54 resample the reach target and cartpole_swingup.py:38-46 draws the +-pi offset in
    reset_model: the in-kernel Philox auto-reset re-noises q / dq only, so stepping with it would run degenerate episodes.  The
    ABI says so (DART_E_UNSUPPORTED) instead of doing it; with the MT19937 bank seeded the same call works."""
    from dart_env_amd import seeding
    card = card_for(env_id)
    s = st.HipStepper(card, 64, precision=64)
    s.configure(st.CFG_AUTORESET, 1)
    a = np.zeros((64, card.act_dim), dtype=np.float32)
    with pytest.raises(st.StepperError) as e:
        s.step(a)
    assert e.value.code == st.E_UNSUPPORTED and "MT19937" in str(e.value)
    s.seed_mt19937(*seeding.mt_keys(list(range(64))))
    s.reset(None, None, None, want_obs=False)
    s.step(a)
    s.close()


def test_walker2d_full_batch_outputs_are_consistent():
    """obs / reward / done of DartWalker2d-v1 at N = 65 536 obey walker2d.py:43-74: obs = [height, q[2:], clip(dq)],
    reward dx/dt + 1 - 1e-3 sum a^2, done outside 0.8 < h < 2.0, |ang| < 1.0."""
    card = card_for("DartWalker2d-v1")
    s = st.HipStepper(card, N_FULL, precision=64)
    s.configure(st.CFG_SEED, 3)
    s.reset(None, None, None, want_obs=False)
    rng = np.random.RandomState(2)
    n_done = 0
    for t in range(8):
        a = rng.uniform(-1.5, 1.5, (N_FULL, 6)).astype(np.float32)
        q0, _ = s.get_state()
        ob, r, done, trunc = s.step(a)
        q, dq = s.get_state()
        assert np.allclose(ob[:, 0], 1.25 + q[:, 1], atol=1e-5) and np.allclose(ob[:, 1:8], q[:, 2:], atol=1e-5)
        assert np.allclose(ob[:, 8:], np.clip(dq, -10, 10), atol=1e-4)
        rew = (q[:, 0] - q0[:, 0]) / 0.008 + 1.0 - 1e-3 * np.sum(a.astype(np.float64) ** 2, axis=1)
        assert np.allclose(r, rew, atol=1e-5)
        ok = np.isfinite(q).all(1) & np.isfinite(dq).all(1) & (np.abs(q[:, 2:]) < 100).all(1) & (np.abs(dq) < 100).all(1) & \
            (ob[:, 0] > .8) & (ob[:, 0] < 2.0) & (np.abs(q[:, 2]) < 1.0)
        border = (np.abs(ob[:, 0] - .8) < 1e-6) | (np.abs(np.abs(q[:, 2]) - 1.0) < 1e-6)
        assert np.array_equal(done[~border], ~ok[~border])
        n_done += int(done.sum())
        if done.any():
            s.reset(done.astype(np.uint8), None, None, want_obs=False)
    assert n_done > 100
    s.close()


def test_humanwalker_full_batch_outputs_are_consistent():
    """DartHumanWalker-v1 at N = 16 384 (BASELINE config 4): observation = [q[1:], clip(dq), two foot-contact flags in {0, 1}]
    (human_walker.py:140-149), a done env earns reward 0 (:127-128), episodes end under random actions."""
    card = card_for("DartHumanWalker-v1")
    n = 16384
    s = st.HipStepper(card, n, precision=64)
    s.configure(st.CFG_SEED, 5)
    s.reset(None, None, None, want_obs=False)
    rng = np.random.RandomState(3)
    n_done = touched = 0
    for t in range(6):
        a = rng.uniform(-1, 1, (n, 23)).astype(np.float32)
        ob, r, done, trunc = s.step(a)
        q, dq = s.get_state()
        # a tumbling humanoid under full-scale random torques can explode inside one env-step (impulses act on M alone, A3): such
        # an env must report done (human_walker.py:121-125: the state test is part of it), like the reference would
        fin = np.isfinite(ob).all(axis=1) & np.isfinite(q).all(axis=1) & np.isfinite(dq).all(axis=1)
        assert ob.shape == (n, 59) and done[~fin].all() and (~fin).sum() < 0.01 * n
        assert np.allclose(ob[fin, :28], q[fin, 1:], atol=1e-5) and np.allclose(ob[fin, 28:57], np.clip(dq[fin], -10, 10), atol=1e-4)
        assert np.isin(ob[:, 57:], (0.0, 1.0)).all()
        assert np.all(r[done] == 0.0) and np.all(np.abs(r[~done]) < 100.0)
        ob = np.where(np.isfinite(ob), ob, 0.0)
        touched += int(ob[:, 57:].sum()); n_done += int(done.sum())
        if done.any():
            s.reset(done.astype(np.uint8), None, None, want_obs=False)
    assert touched > n and n_done > 0
    s.close()
