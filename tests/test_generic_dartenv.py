"""The generic DartEnv base class (reference gym/envs/dart/dart_env.py:25-215) for user-defined tasks: a .skel file written
for this test, a subclass written like the reference's env classes, checked against the same task on oracle worlds."""
import numpy as np
import pytest

from tests.fake_stepper import OracleStepper
from tests.inchworm_env import InchwormEnv, SKEL, reference_rollout


def test_generic_dartenv_host_layer_cpu():
    env = InchwormEnv(num_envs=3, stepper_factory=OracleStepper)
    assert env.robot_skeleton.ndofs == 5 and env.dt == pytest.approx(0.01)
    assert env.action_space.shape == (2,) and env.observation_space.shape == (9,)
    assert [b.name for b in env.robot_skeleton.bodynodes] == ["carrier_x", "carrier_y", "mid", "front", "rear"]
    assert np.array_equal(env.robot_skeleton.q_lower[3:], [-1.1, -0.9]) and np.isinf(env.robot_skeleton.q_lower[0])
    assert env.seed(11) == [11, 12, 13]
    rng = np.random.RandomState(0)
    actions = rng.uniform(-1.5, 1.5, (30, 3, 2))
    obs = [env.reset()]
    rew, done = [], []
    for t in range(30):
        o, r, d, info = env.step(actions[t])
        obs.append(o); rew.append(r); done.append(d)
    ro, rr, rd = reference_rollout(env.card, [11, 12, 13], actions)
    assert np.allclose(np.stack(obs), ro, atol=1e-12) and np.allclose(np.stack(rew), rr, atol=1e-9) and np.array_equal(np.stack(done), rd)
    assert np.allclose(env.state_vector()[:, :5], env.robot_skeleton.q)
    with pytest.raises(ValueError):
        env.do_simulation(np.zeros((3, 5)), 3)          # not a multiple of frame_skip
    env.set_state_vector(np.zeros((3, 10)))
    assert np.all(env.state_vector() == 0)
    env.close()
    with pytest.raises(IOError):
        from dart_env_amd.envs.dart_env import DartEnv
        DartEnv("/nonexistent/model.skel", 4, 3, np.array([[1.0], [-1.0]]), stepper_factory=OracleStepper)
    with pytest.raises(NotImplementedError):
        from dart_env_amd.envs.dart_env import DartEnv
        DartEnv(SKEL, 4, 3, np.array([[1.0], [-1.0]]), obs_type="image", stepper_factory=OracleStepper)


@pytest.mark.gpu
def test_generic_dartenv_on_gpu_matches_oracle_worlds():
    n = 32
    env = InchwormEnv(num_envs=n, precision=64)
    env.enable_contact_report()
    env.seed(5)
    rng = np.random.RandomState(1)
    T = 60
    actions = rng.uniform(-1.5, 1.5, (T, n, 2))
    obs = [env.reset()]
    rew, done = [], []
    touched = 0
    for t in range(T):
        o, r, d, info = env.step(actions[t])
        obs.append(o); rew.append(r); done.append(d)
        touched += int((env.contacts()[0] > 0).sum())
    ro, rr, rd = reference_rollout(env.card, [5 + i for i in range(n)], actions)
    assert np.abs(np.stack(obs) - ro).max() < 1e-7 and np.abs(np.stack(rew) - rr).max() < 1e-5 and np.array_equal(np.stack(done), rd)
    assert touched > n * T // 4                       # the worm lies on the floor for most of the rollout
    com = env.robot_skeleton.bodynode("front").com()
    assert com.shape == (n, 3) and np.all(com[:, 1] > 0)
    env.close()
    # torch tensors resident in HBM: same physics, no host round trip
    import torch
    ea, eb = InchwormEnv(num_envs=n, precision=32), InchwormEnv(num_envs=n, precision=32)
    for e in (ea, eb):
        e.seed(5); e.reset()
    for t in range(8):
        tau = np.zeros((n, 5), dtype=np.float32); tau[:, 3:] = np.clip(actions[t], -1, 1) * [6.0, 4.0]
        ea.do_simulation(tau, 5)
        sdev = eb.do_simulation(torch.from_numpy(tau).cuda(), 10 if t == 0 else 5)
        if t == 0:
            ea.do_simulation(tau, 5)
    torch.cuda.synchronize()
    assert np.array_equal(sdev.cpu().numpy(), ea.state_vector().astype(np.float32)) and np.array_equal(eb.state_vector(), ea.state_vector())
    ea.close(); eb.close()
    env32 = InchwormEnv(num_envs=n, precision=32)
    env32.seed(5)
    o32 = [env32.reset()]
    for t in range(10):
        o32.append(env32.step(actions[t])[0])
    assert np.abs(np.stack(o32) - ro[:11]).max() < 2e-3
    env32.close()


def test_user_skel_with_a_compiled_topology_is_accepted_by_the_lane_kernel_code():
    """SURVEY 8(f)-1 "model compiler generality": a physics-only card (DART_TASK_NONE -- what envs.DartEnv builds from a user's
    .skel) whose tree is the hopper chain goes through `make_planar` (PhysTopo<HopperAllTopo>), here in the host build of the lane
    kernels: states, [q, dq] observations and zero rewards against oracle worlds over 150 env-steps with every capsule on the floor."""
    from dart_env_amd.model_card import build_card
    from dart_env_amd.skel import parse_skel
    from tests.batch_oracle import OracleBatch
    from tests.emu_lib import EmuStepper
    from tests.pogo_env import SKEL as POGO, SCALE
    card = build_card(parse_skel(POGO), None)
    card.frame_skip = 4
    n = 48
    g = EmuStepper(card, n, precision=64)          # raises if no lane kernel takes the card
    assert not g.is_static and card.obs_dim == 12 and card.act_dim == 6
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(0)
    qn = rng.uniform(-.01, .01, (n, 6)); vn = rng.uniform(-.01, .01, (n, 6))
    ora.reset(None, qn, vn)
    assert np.abs(g.reset(None, qn, vn) - ora.obs()).max() < 1e-6
    for t in range(150):
        a = np.zeros((n, 6), np.float32); a[:, 3:] = rng.uniform(-1, 1, (n, 3)) * SCALE
        o, r, d, tr = g.step(a); oo, ro, do, _ = ora.step(a)
        assert np.abs(o - oo).max() < 1e-5 and not r.any() and not ro.any() and not d.any() and not do.any()
    qg, dqg = g.get_state(); qo, dqo = ora.state()
    assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-8 and qo[:, 1].min() < -0.6     # they all ended up on the floor
    # the walker tree (the reference's walker2d.skel as a physics-only card): PhysTopo<Walker2dAllTopo>, up to five of its seven
    # capsules on the floor -- the second register tier and, beyond it, the fallback solver
    from dart_env_amd.model_card import load_model
    wcard = build_card(load_model("walker2d"), None)
    wcard.frame_skip = 4
    n = 24
    g = EmuStepper(wcard, n, precision=64)
    ora = OracleBatch(wcard, n)
    qn = rng.uniform(-.01, .01, (n, 9)); vn = rng.uniform(-.01, .01, (n, 9))
    ora.reset(None, qn, vn); g.reset(None, qn, vn)
    most = 0
    for t in range(160):
        a = np.zeros((n, 9), np.float32); a[:, 3:] = rng.uniform(-1, 1, (n, 6)) * [100, 100, 20, 100, 100, 20]
        g.step(a); ora.step(a)
        most = max(most, max(len(w.last_contacts()) for w in ora.worlds))
    qg, dqg = g.get_state(); qo, dqo = ora.state()
    assert most >= 4 and np.abs(qg - qo).max() < 1e-8 and np.abs(dqg - dqo).max() < 1e-7, (most, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
    # a model of another shape is declined by every lane kernel (and lands on the tree kernel in the product)
    with pytest.raises(RuntimeError):
        EmuStepper(build_card(parse_skel(SKEL), None), 4, precision=64)


@pytest.mark.gpu
def test_user_task_on_a_matching_skel_runs_one_env_per_lane():
    """The pogo task (tests/pogo_env.py: a subclass written like the reference's hopper.py) on the GPU: the library picks the
    hopper-chain register kernel for the user's .skel (DART_Q_LANE_KERNEL), the same task forced onto the tree kernel and the task
    on oracle worlds give the same rollout, and the lane kernel is the faster of the two by a wide margin."""
    import time
    from dart_env_amd import stepper as st
    from tests.pogo_env import PogoEnv, reference_rollout as pogo_reference
    n, T = 32, 50
    rng = np.random.RandomState(2)
    actions = rng.uniform(-1.5, 1.5, (T, n, 3))
    outs = {}
    for generic in (False, True):
        env = PogoEnv(num_envs=n, precision=64, generic_kernel=generic)
        assert env._stepper.query(st.Q_LANE_KERNEL) == (0 if generic else 1)
        env.seed(7)
        obs = [env.reset()]; rew, done = [], []
        for t in range(T):
            o, r, d, info = env.step(actions[t])
            obs.append(o); rew.append(r); done.append(d)
        outs[generic] = (np.stack(obs), np.stack(rew), np.stack(done))
        card = env.card
        env.close()
    ro, rr, rd = pogo_reference(card, [7 + i for i in range(n)], actions)
    for generic in (False, True):
        o, r, d = outs[generic]
        assert np.abs(o - ro).max() < 1e-7 and np.abs(r - rr).max() < 1e-5 and np.array_equal(d, rd), generic
    assert rd.any()
    big = {}
    for generic in (False, True):
        env = PogoEnv(num_envs=16384, precision=64, generic_kernel=generic)
        env.seed(0); env.reset()
        tau = np.zeros((16384, 6), dtype=np.float32)
        env.do_simulation(tau, 4)
        t0 = time.perf_counter()
        for _ in range(10):
            env.do_simulation(tau, 4)
        big[generic] = (time.perf_counter() - t0) / 10
        env.close()
    assert big[False] < big[True]


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pogo", "walker-tree"])
@pytest.mark.parametrize("precision,tq,tdq", [(64, 1e-7, 1e-5), (32, 2e-3, 5e-2)])
def test_fallen_user_model_stays_in_the_register_tiers(precision, tq, tdq, model):
    """A user model has no termination inside the library: the pogo hoppers fall over under random torques and stay on the floor with
    every capsule touching.  The physics-only lane variants size their register tiers for that (PhysTopo in planar_kernel.hpp: all four
    capsules of the hopper chain), the rollout stays on the oracle's, bitwise repeatable, and a batched step of fallen hoppers stays
    well under the 2.2-2.4 ms it took when every lane waited its turn in the single-lane fallback solver."""
    import time
    from dart_env_amd import stepper as st
    from tests.batch_oracle import OracleBatch
    from tests.pogo_env import PogoEnv
    if model == "pogo":
        env = PogoEnv(num_envs=4, precision=precision)
        card = env.card
        env.close()
        scale = [40.0, 30.0, 15.0]
    else:   # the reference's walker2d.skel as a physics-only card: seven capsules, a second tier of 4 (fp64: 3) slots as a real call,
        from dart_env_amd.model_card import build_card, load_model     # the wave-served fallback beyond it
        card = build_card(load_model("walker2d"), None)
        card.frame_skip = 4
        scale = [100.0, 100.0, 20.0, 100.0, 100.0, 20.0]
    nd, na = card.ndofs, len(scale)
    n, T = 128, 160
    rng = np.random.RandomState(5)
    qn = rng.uniform(-.05, .05, (n, nd)); vn = rng.uniform(-.05, .05, (n, nd))
    taus = (np.concatenate([np.zeros((T, n, 3)), rng.uniform(-1, 1, (T, n, na)) * scale], axis=2)).astype(np.float32)
    finals = []
    for rep in range(2):
        gpu = st.HipStepper(card, n, precision=precision)
        assert gpu.query(st.Q_LANE_KERNEL) == 1
        gpu.reset(None, qn, vn, want_obs=False)
        if rep == 0:
            ora = OracleBatch(card, n)
            ora.reset(None, qn, vn)
        most = 0
        for t in range(T):
            gpu.step(taus[t])
            if rep == 0:
                ora.step(taus[t])
                most = max(most, max(len(w.last_contacts()) for w in ora.worlds[:32]))
                qg, dqg = gpu.get_state(); qo, dqo = ora.state()
                # fp32: one env-step from the oracle's state; whether a resting capsule counts as touching in this step is decided
                # within rounding, so a few envs differ by an impact (dq by ~1, q by dt x that) -- the bulk is held, every position within 0.05
                eq, edq = np.abs(qg - qo), np.abs(dqg - dqo)
                eq, edq = (eq.max(), edq.max()) if precision == 64 else (np.percentile(eq, 99), np.percentile(edq, 99))
                assert eq < tq and edq < tdq and np.abs(qg - qo).max() < 0.05, (t, eq, edq, np.abs(qg - qo).max())
                if precision == 32:
                    gpu.set_state(qo, dqo)
        if rep == 0:
            assert most >= (3 if model == "pogo" else 4), most      # more capsules on the floor than the base topology's tiers hold
        finals.append(np.concatenate(gpu.get_state(), axis=1))
        gpu.close()
    if precision == 64:
        assert np.array_equal(finals[0], finals[1])      # bitwise repeatable (the fp32 run is re-synchronised to the oracle, rep 1 is not)
    # cost: 16 384 hoppers, standing (first steps) against fallen (after 150 steps of random torques)
    big = st.HipStepper(card, 16384, precision=precision)
    big.reset(None, None, None, want_obs=False)
    tau = np.concatenate([np.zeros((16384, 3)), np.random.RandomState(0).uniform(-1, 1, (16384, na)) * scale], axis=1).astype(np.float32)
    def timed(k):
        t0 = time.perf_counter()
        for _ in range(k):
            big.step(tau)
        return (time.perf_counter() - t0) / k
    timed(3)
    standing = timed(10)
    timed(150)
    fallen = timed(10)
    print("%s x16384 fp%d: standing %.0f us, fallen %.0f us per env-step (host-synchronous)" % (model, precision, standing * 1e6, fallen * 1e6))
    big.close()
    # (a fallen walker lies on five to seven capsules, beyond its tiers: every lane is served by the wave in turn -- bounded, not fast)
    assert fallen < (1.2e-3 if model == "pogo" else 40e-3)


PHYS_SHAPES = [("halfcheetah", [120, 90, 60, 120, 60, 30], 3, True), ("snake7link", [20] * 6, 3, False), ("cartpole", [5.0], 1, False),
               ("cartpole_swingup", [5.0], 1, False), ("double_pendulum", [3.0, 3.0], 1, False), ("reacher2d", [2.0, 2.0], 0, False),
               ("reacher3d", [3.0] * 5, 0, False)]


@pytest.mark.parametrize("name,scale,nroot,contacts", PHYS_SHAPES, ids=[s[0] for s in PHYS_SHAPES])
def test_physics_only_card_of_every_compiled_shape_runs_on_a_lane_kernel(name, scale, nroot, contacts):
    """a user's .skel (envs.DartEnv builds a DART_TASK_NONE card from it: action = generalized forces, obs = [q, dq],
    reward 0, never done -- reference dart_env.py:28-175) whose tree is one of the compiled shapes -- the half-cheetah tree with its eight
    capsules, the seven-link chain in the horizontal plane, cart + one / two links, the two-link arm, the five-link 3-D chain -- is served
    one env per lane (round 4: hopper chain and walker tree only; everything else fell to the tree kernel, 80x slower).  Here on the host
    build of the lane kernels against oracle worlds, 100 env-steps under random forces on every actuated dof."""
    from dart_env_amd.model_card import build_card, load_model
    from tests.batch_oracle import OracleBatch
    from tests.emu_lib import EmuStepper
    m = load_model(name)
    if not contacts:
        for s in m.shapes:
            s.collidable = False
    card = build_card(m, None)
    card.frame_skip = 4
    n, nd = 12, card.ndofs
    g = EmuStepper(card, n, precision=64)          # raises if no lane kernel takes the card
    assert card.obs_dim == 2 * nd and card.act_dim == nd
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(0)
    qn = rng.uniform(-.02, .02, (n, nd)); vn = rng.uniform(-.02, .02, (n, nd))
    ora.reset(None, qn, vn)
    assert np.abs(g.reset(None, qn, vn) - ora.obs()).max() < 1e-6
    most = 0
    for t in range(100):
        a = np.zeros((n, nd), np.float32); a[:, nroot:] = rng.uniform(-1, 1, (n, nd - nroot)) * (scale if len(scale) == nd - nroot else scale[0])
        o, r, d, tr = g.step(a); oo, ro, do, _ = ora.step(a)
        assert not r.any() and not d.any() and np.abs(o - oo).max() < 1e-4, t
        most = max(most, max(len(w.last_contacts()) for w in ora.worlds))
    qg, dqg = g.get_state(); qo, dqo = ora.state()
    assert np.abs(qg - qo).max() < 1e-8 and np.abs(dqg - dqo).max() < 1e-7, (np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
    assert most >= (3 if contacts else 0)          # the cheetah tree ends up on three or more capsules: second register tier
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,nroot,contacts", PHYS_SHAPES, ids=[s[0] for s in PHYS_SHAPES])
def test_physics_only_card_of_every_compiled_shape_reports_a_lane_kernel_on_the_gpu(name, scale, nroot, contacts):
    """... and on the device: DART_Q_LANE_KERNEL = 1, the same card forced onto the tree kernel (generic_kernel) and oracle worlds agree."""
    from dart_env_amd import stepper as st
    from dart_env_amd.model_card import build_card, load_model
    from tests.batch_oracle import OracleBatch
    m = load_model(name)
    if not contacts:
        for s in m.shapes:
            s.collidable = False
    card = build_card(m, None); card.frame_skip = 4
    tcard = build_card(m, None); tcard.frame_skip = 4; tcard.generic_kernel = 1
    n, nd = 64, card.ndofs
    lane = st.HipStepper(card, n, precision=64); tree = st.HipStepper(tcard, n, precision=64)
    assert lane.query(st.Q_LANE_KERNEL) == 1 and tree.query(st.Q_LANE_KERNEL) == 0
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(1)
    qn = rng.uniform(-.02, .02, (n, nd)); vn = rng.uniform(-.02, .02, (n, nd))
    for s_ in (lane, tree):
        s_.reset(None, qn, vn, want_obs=False)
    ora.reset(None, qn, vn)
    for t in range(40):
        a = np.zeros((n, nd), np.float32); a[:, nroot:] = rng.uniform(-1, 1, (n, nd - nroot)) * (scale if len(scale) == nd - nroot else scale[0])
        lane.step(a); tree.step(a); ora.step(a)
    ql, dql = lane.get_state(); qt, dqt = tree.get_state(); qo, dqo = ora.state()
    assert np.abs(ql - qo).max() < 1e-7 and np.abs(dql - dqo).max() < 1e-6, (np.abs(ql - qo).max(), np.abs(dql - dqo).max())
    assert np.abs(qt - qo).max() < 1e-7 and np.abs(dqt - dqo).max() < 1e-6, (np.abs(qt - qo).max(), np.abs(dqt - dqo).max())
    lane.close(); tree.close()
