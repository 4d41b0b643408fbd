"""The planar DEVICE kernels, compiled for the host (tests/emu_lib.py), against the fp64 oracle -- runs without a GPU.

This is the arithmetic of dart_env_amd/csrc/planar_kernel.hpp itself (world-aligned planar CRBA, LDL^T, pivoting LCP, task
epilogue, Philox auto-reset), lane by lane; what it cannot show is anything wave-level (votes, occupancy) or the fp32 hardware
approximations -- the `-m gpu` tests hold the real build to the same oracle."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
from tests.parity_protocol import make_reference, run_host_api


@pytest.mark.parametrize("impulse_inertia", [0, 1])
@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_fp64_kernel_code_meets_the_north_star_bound_on_cpu(env_id, impulse_inertia):
    """both settings of the A3 knob (card.impulse_inertia: 0 = DART 6's impulse pass on M, the default and the baked kernels;
    1 = on M + dt D + dt^2 K, served by the runtime-parameter kernel)"""
    card = card_for(env_id)
    card.impulse_inertia = impulse_inertia
    acts, ref = make_reference(card, 128, 300)
    g = EmuStepper(card, 128, precision=64)
    assert g.is_static == (impulse_inertia == 0)
    s = run_host_api(g, acts, ref)
    assert s["done_flag_mismatches"] == 0 and s["episodes"] > 100
    assert s["q"] < 1e-9 and s["dq"] < 1e-8, (s["q"], s["dq"])          # bound to beat: 1e-4


def test_generic_and_baked_kernels_agree():
    """runtime-parameter kernel (any card of the topology) = compile-time specialisation, bit for bit in fp64"""
    card = card_for("DartWalker2d-v1")
    acts, ref = make_reference(card, 64, 60)
    a = EmuStepper(card, 64, precision=64, allow_static=True); b = EmuStepper(card, 64, precision=64, allow_static=False)
    assert a.is_static and not b.is_static
    run_host_api(a, acts, ref); run_host_api(b, acts, ref)
    qa, dqa = a.get_state(); qb, dqb = b.get_state()
    assert np.abs(qa - qb).max() < 1e-12 and np.abs(dqa - dqb).max() < 1e-10


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
@pytest.mark.parametrize("force_fallback", [False, True])
def test_default_cards_collide_every_capsule_small_torques(env_id, force_fallback):
    """Small torques: the robots sink to the floor and knees / shins touch before `done` -- the tier-1 contact slots, and with
    force_fallback the single-lane loop solver that serves an env with more contacts than the tiers hold."""
    from dart_env_amd import stepper as st
    card = card_for(env_id)
    assert all(card.shape_collidable[s] for s in range(card.nshapes))
    acts, ref = make_reference(card, 96, 300)
    acts *= 0.1
    from tests import oracle_lib as ol
    ref = ol.rollout_trace(card, acts, ref["snap_steps"])
    g = EmuStepper(card, 96, precision=64)
    if force_fallback:
        g.force_slow(True)
    s = run_host_api(g, acts, ref)
    assert s["done_flag_mismatches"] == 0
    assert s["q"] < 1e-9 and s["dq"] < 1e-8, (s["q"], s["dq"])


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_robot_lying_on_the_floor_takes_the_fallback_solver(env_id):
    """Every capsule on the ground at once (more contacts than any register tier): the natural trigger of slow_constraints."""
    from tests.oracle_lib import OracleWorld
    card = card_for(env_id)
    n = 4
    g = EmuStepper(card, n, precision=64)
    nd = card.ndofs
    q = np.zeros((n, nd)); dq = np.zeros((n, nd))
    for i in range(n):
        q[i, 1] = -1.17 + 0.01 * i          # root height: the capsules just reach into the floor
        q[i, 2] = 1.5 + 0.02 * i            # lying on its side
        q[i, 3:] = 0.05 * (i + 1) * (-1.0) ** np.arange(nd - 3) * 0.2
    g.set_state(q, dq)
    worlds = [OracleWorld(card) for _ in range(n)]
    for w, qi, dqi in zip(worlds, q, dq):
        w.set_state(qi, dqi)
    most = 0
    for t in range(40):
        a = np.zeros((n, card.act_dim), dtype=np.float32)
        g.step(a)
        for w in worlds:
            w.env_step(a[0].astype(np.float64))
            most = max(most, len(w.last_contacts()))
        qg, dqg = g.get_state()
        qo = np.stack([w.q for w in worlds]); dqo = np.stack([w.dq for w in worlds])
        assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-7, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
    assert most >= 3, most       # more touching capsules than the register tiers of either model hold


@pytest.mark.parametrize("env_id,body", [("DartHopper-v1", 3), ("DartWalker2d-v1", 5), ("DartHalfCheetah-v1", 6)])
@pytest.mark.parametrize("force_fallback", [False, True])
def test_contact_report_constraint_forces_and_external_force(env_id, body, force_fallback):
    """What walker2d.py:38-41 reads (collision_result.contacts: body, point, force), skel.constraint_forces() and the
    perturbation branch of do_simulation (add_ext_force, dart_env.py:159-172) on the planar register kernels' code."""
    from tests.batch_oracle import OracleBatch
    card = card_for(env_id); n = 16; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(4)
    g = EmuStepper(card, n, precision=64); o = OracleBatch(card, n)
    if force_fallback:
        g.force_slow(True)
    g.enable_contact_report(True)
    F = rng.uniform(-30, 30, (n, 3)); F[:, 2] = 0
    g.set_ext_force(body, F)
    for i, w in enumerate(o.worlds):
        w.set_ext_force(body, F[i])
    qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
    g.reset(None, qn, vn); o.reset(None, qn, vn)
    seen = 0
    for t in range(50):
        a = (rng.uniform(-1, 1, (n, na)) * 0.3).astype(np.float32)
        g.step(a); _, _, do, _ = o.step(a)
        cnt, bod, pt, fc = g.contacts(); cf = g.constraint_forces()
        qg, dqg = g.get_state(); qo, dqo = o.state()
        assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-7
        for i, w in enumerate(o.worlds):
            rep = w.contact_report(); k = len(rep)
            assert cnt[i] == k
            assert np.abs(cf[i] - w.constraint_forces()).max() < 1e-6
            if k:
                seen += k
                assert np.array_equal(bod[i, :k, 0], rep[:, 0].astype(np.int32)) and np.all(bod[i, :k, 1] == -1)
                assert np.abs(pt[i, :k] - rep[:, 2:5]).max() < 1e-9 and np.abs(fc[i, :k] - rep[:, 5:8]).max() < 1e-6
        if do.any():
            qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
            g.reset(do.astype(np.uint8), qn, vn, want_obs=False); o.reset(do, qn, vn)
    assert seen > 100


def test_half_cheetah_on_the_register_kernel():
    """half_cheetah.skel: welded head folded into the torso link, joint springs, eight capsules (tiers of 2 / 4 contact slots)"""
    card = card_for("DartHalfCheetah-v1")
    # 150 env-steps: a cheetah thrashing under random torques is chaotic (two copies of the ORACLE 1e-15 apart are 7e-7 apart after
    # 200 env-steps), the kernel-vs-oracle difference follows the same curve: 3e-14 / 5e-12 / 7e-11 / 1e-9 at 50 / 100 / 150 / 200
    acts, ref = make_reference(card, 64, 150)
    s = run_host_api(EmuStepper(card, 64, precision=64), acts, ref)
    assert s["done_flag_mismatches"] == 0 and s["q"] < 1e-9 and s["dq"] < 1e-8, (s["q"], s["dq"])


def test_pivoting_cap_is_not_reached_on_the_hard_cheetah_lcps():
    """Regression (round 3): with the impulse pass on M (A3) the Delassus matrices are stiffer, and 2 of 4 096 half-cheetah envs ran
    into the pivoting solver's old iteration cap of 24 within 10 env-steps (warm-started sets, three touching capsules) -- the
    kernel then kept the clipped iterate and left the oracle's trajectory by O(1).  The cap is 200 now (it costs nothing until it is
    needed: the loop ends on the wave's vote); every env of this sample stays on the oracle's trajectory."""
    card = card_for("DartHalfCheetah-v1")
    acts, ref = make_reference(card, 4096, 12)
    s = run_host_api(EmuStepper(card, 4096, precision=64), acts, ref)
    assert s["done_flag_mismatches"] == 0 and all(v["envs_beyond_1e-4"] == 0 for v in s["by_step"].values())
    assert s["q"] < 1e-12 and s["dq"] < 1e-10, (s["q"], s["dq"])


@pytest.mark.parametrize("env_id,qnoise,vnoise", [("DartCartPole-v1", 0.01, 0.01), ("DartCartPoleSwingUp-v1", 0.1, 0.01),
                                                   ("DartDoubleInvertedPendulumEnv-v1", 0.1, 0.1)])
def test_cart_family_on_the_lane_kernel(env_id, qnoise, vnoise):
    """csrc/cart_kernel.hpp: slider + one or two hinged poles, joint limits through the pivoting LCP, tasks 5 / 7 / 8 -- state,
    observation, reward and done flags against the oracle with resets on the oracle's flags"""
    from tests.batch_oracle import OracleBatch
    card = card_for(env_id); n = 64; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(3)
    g = EmuStepper(card, n, precision=64); o = OracleBatch(card, n)
    qn = rng.uniform(-qnoise, qnoise, (n, nd)); vn = rng.uniform(-vnoise, vnoise, (n, nd))
    og = g.reset(None, qn, vn); o.reset(None, qn, vn)
    assert np.allclose(og, o.obs(), atol=1e-6)
    worst = dict(q=0.0, dq=0.0, obs=0.0, rew=0.0); dones = 0
    for t in range(150):
        a = rng.uniform(-1.5, 1.5, (n, na)).astype(np.float32)
        ob, r, d, _ = g.step(a); oo, ro, do, _ = o.step(a)
        assert np.array_equal(d.astype(bool), do.astype(bool)), t
        qg, dqg = g.get_state(); qo, dqo = o.state()
        for key, x, y in (("q", qg, qo), ("dq", dqg, dqo), ("obs", ob, oo), ("rew", r, ro)):
            worst[key] = max(worst[key], float(np.abs(np.asarray(x, dtype=np.float64) - y).max()))
        dones += int(do.sum())
        if do.any():
            qn = rng.uniform(-qnoise, qnoise, (n, nd)); vn = rng.uniform(-vnoise, vnoise, (n, nd))
            g.reset(do.astype(np.uint8), qn, vn, want_obs=False); o.reset(do, qn, vn)
    assert worst["q"] < 1e-9 and worst["dq"] < 1e-8 and worst["obs"] < 1e-5 and worst["rew"] < 5e-6, worst       # obs and reward cross the ABI as float32
    if env_id != "DartCartPoleSwingUp-v1":
        assert dones > 0


def test_snake_on_the_register_kernel():
    """snake_7link.skel on SnakeTopo: motion in the horizontal x-z plane (rotations about +y = clockwise in (x, z)), no contact
    rows, six limit rows, the fluid force on every body before every world step, deviation cost and |q[2]| >= 1.5 ending"""
    from tests.batch_oracle import OracleBatch
    card = card_for("DartSnake7Link-v1"); n = 64; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(3)
    g = EmuStepper(card, n, precision=64); o = OracleBatch(card, n)
    qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
    og = g.reset(None, qn, vn); o.reset(None, qn, vn)
    assert np.allclose(og, o.obs(), atol=1e-6)
    dones = 0
    for t in range(200):
        a = rng.uniform(-1.5, 1.5, (n, na)).astype(np.float32)
        ob, r, d, _ = g.step(a); oo, ro, do, _ = o.step(a)
        assert np.array_equal(d.astype(bool), do.astype(bool)), t
        qg, dqg = g.get_state(); qo, dqo = o.state()
        assert np.abs(qg - qo).max() < 1e-10 and np.abs(dqg - dqo).max() < 1e-9, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
        assert np.allclose(ob, oo, atol=1e-5) and np.allclose(r, ro, atol=5e-6)
        dones += int(do.sum())
        if do.any():
            qn = rng.uniform(-.005, .005, (n, nd)); vn = rng.uniform(-.005, .005, (n, nd))
            g.reset(do.astype(np.uint8), qn, vn, want_obs=False); o.reset(do, qn, vn)
    assert dones > 10


def test_reacher2d_on_the_arm_lane_kernel():
    """csrc/arm_kernel.hpp: two links on +y hinges in the horizontal plane, fixed base, welded finger tip folded into the last link,
    one joint-limit and one Coulomb-friction row per dof in the LCP, per-env reach targets as task state, TimeLimit resets"""
    from tests.batch_oracle import OracleBatch
    card = card_for("DartReacher-v1"); n, nd = 64, card.ndofs
    rng = np.random.RandomState(4)
    g = EmuStepper(card, n, precision=64); o = OracleBatch(card, n)
    tg = np.zeros((n, 3)); tg[:, 0] = rng.uniform(-.2, .2, n); tg[:, 1] = 0.01; tg[:, 2] = rng.uniform(-.2, .2, n)
    g.set_task_state(None, tg)
    for i, w in enumerate(o.worlds):
        w.set_task_state(tg[i])
    qn = rng.uniform(-.01, .01, (n, nd)); vn = np.zeros((n, nd)); vn[n // 2:] = rng.uniform(-.005, .005, (n // 2, nd))
    og = g.reset(None, qn, vn); o.reset(None, qn, vn)
    assert np.allclose(og, o.obs(), atol=1e-6)
    for t in range(49):
        a = rng.uniform(-1.3, 1.3, (n, 2)).astype(np.float32)
        a[: n // 2] = 0.0002 * np.sign(a[: n // 2])           # 0.04 Nm < 0.05 Nm of Coulomb friction: that half stays put
        ob, r, d, tr = g.step(a); oo, ro, do, to = o.step(a)
        qg, dqg = g.get_state(); qo, dqo = o.state()
        assert np.abs(qg - qo).max() < 1e-12 and np.abs(dqg - dqo).max() < 1e-11, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
        assert np.allclose(ob, oo, atol=1e-5) and np.allclose(r, ro, atol=5e-6) and not d.any() and not do.any()
    q, dq = g.get_state()
    assert np.abs(dq[: n // 2]).max() < 1e-9 and np.abs(q[: n // 2] - qn[: n // 2]).max() < 1e-9      # stuck by friction
    assert np.abs(dq[n // 2:]).max() > 0.1
    ob, r, d, tr = g.step(np.zeros((n, 2), dtype=np.float32)); oo, ro, do, to = o.step(np.zeros((n, 2), dtype=np.float32))
    assert d.all() and tr.all() and do.all()                  # step 50: the TimeLimit ends every episode


def test_reacher3d_on_the_chain_lane_kernel():
    """csrc/chain3d_kernel.hpp: reacher.skel's universal - revolute - universal joints as five revolute links in 3-D, one env per
    lane: free motion, joint limits through the LCP, reward / done from the tip-target distance BEFORE the step (reacher.py:23-33)"""
    from tests.batch_oracle import OracleBatch
    card = card_for("DartReacher3d-v1"); n, nd = 64, card.ndofs
    rng = np.random.RandomState(5)
    g = EmuStepper(card, n, precision=64); o = OracleBatch(card, n)
    tg = rng.uniform(-.3, .3, (n, 3))
    g.set_task_state(None, tg)
    for i, w in enumerate(o.worlds):
        w.set_task_state(tg[i])
    qn = rng.uniform(-.01, .01, (n, nd)); vn = rng.uniform(-.01, .01, (n, nd))
    og = g.reset(None, qn, vn); o.reset(None, qn, vn)
    assert np.allclose(og, o.obs(), atol=1e-6)
    tip = og[:, -3:].astype(np.float64) + tg                     # obs ends with tip - target
    tg[:16] = tip[:16] + 0.01                                    # these envs start within 0.1 of their target: done at once
    g.set_task_state(None, tg)
    for i, w in enumerate(o.worlds):
        w.set_task_state(tg[i])
    q, dq = g.get_state()
    q[32:] = rng.choice([-3.13, 3.13], (32, nd)); dq[32:] = np.sign(q[32:]) * 4.0      # these run into their joint limits
    g.set_state(q, dq)
    for i in range(n):
        o.worlds[i].set_state(q[i], dq[i])
    dones = beyond = 0
    for t in range(60):
        a = rng.uniform(-1.3, 1.3, (n, 5)).astype(np.float32)
        ob, r, d, tr = g.step(a); oo, ro, do, to = o.step(a)
        qg, dqg = g.get_state(); qo, dqo = o.state()
        assert np.array_equal(d.astype(bool), do.astype(bool)), t
        assert np.abs(qg - qo).max() < 1e-12 and np.abs(dqg - dqo).max() < 1e-11, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
        assert np.allclose(ob, oo, atol=1e-5) and np.allclose(r, ro, atol=1e-5)
        dones += int(do.sum()); beyond += int((np.abs(qo) >= 3.14).sum())
        if do.any():
            qn = rng.uniform(-.01, .01, (n, nd)); vn = rng.uniform(-.01, .01, (n, nd))
            g.reset(do.astype(np.uint8), qn, vn, want_obs=False); o.reset(do, qn, vn)
    assert dones >= 16 and beyond > 100
