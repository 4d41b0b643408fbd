"""The lane kernels (planar_kernel.hpp: one env per lane) executed as WHOLE 64-lane wavefronts on the host: the device source on the fiber
runtime of tests/kernel_emu/fake_wave_include (see tests/test_tree_kernel_emu_parity.py).  tests/test_kernel_emu_parity.py runs the same
kernels one lane at a time, which leaves out everything lanes do together on the device: the wave votes that pick a tier and stop the
pivoting loops, the wave-served fallback for an env beyond its register tiers (wave_constraints), the hand-off of lanes that keep
pivoting to the wave solver (blcp_bpp: coop) and their iteration budget.  Here they run as they do on the GPU, against the fp64 oracle."""
import numpy as np
import pytest

from dart_env_amd.model_card import build_card, card_for, load_model
from tests.batch_oracle import OracleBatch
from tests.emu_lib import EmuStepper


def rollout(card, n, T, force_fallback=False, noise=0.1, scale=None, seed=0):
    g = EmuStepper(card, n, precision=64, waves=True)
    if force_fallback:
        g.force_slow(True)
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(seed)
    nd = card.ndofs
    qn = rng.uniform(-noise, noise, (n, nd)); vn = rng.uniform(-noise, noise, (n, nd))
    g.reset(None, qn, vn); ora.reset(None, qn, vn)
    worst, most = [0.0, 0.0], 0
    for t in range(T):
        a = rng.uniform(-1, 1, (n, card.act_dim))
        a = (a if scale is None else np.concatenate([np.zeros((n, 3)), a[:, 3:] * scale], axis=1)).astype(np.float32)
        o, r, d, tr = g.step(a)
        oo, ro, do, to = ora.step(a)
        most = max(most, max(len(w.last_contacts()) for w in ora.worlds))
        qg, dqg = g.get_state(); qo, dqo = ora.state()
        worst = [max(worst[0], np.abs(qg - qo).max()), max(worst[1], np.abs(dqg - dqo).max())]
        assert np.array_equal(d.astype(bool), np.asarray(do, bool)), t
        if np.any(do):
            qn = rng.uniform(-noise, noise, (n, nd)); vn = rng.uniform(-noise, noise, (n, nd))
            g.reset(np.asarray(do, np.uint8), qn, vn, want_obs=False); ora.reset(np.asarray(do, bool), qn, vn)
    g.close()
    return worst, most


@pytest.mark.parametrize("env_id,n,T", [("DartHopper-v1", 128, 30), ("DartWalker2d-v1", 64, 30), ("DartHalfCheetah-v1", 128, 40), ("DartSnake7Link-v1", 64, 20)])
def test_whole_waves_follow_the_oracle(env_id, n, T):
    worst, most = rollout(card_for(env_id), n, T)
    assert worst[0] < 1e-9 and worst[1] < 1e-7, worst


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1", "DartHalfCheetah-v1"])
def test_limit_slot_vote_and_the_all_limits_tier_follow_the_oracle(env_id):
    """Round 6: the small register tier of the Walker2d / half-cheetah kernels carries 4 / 3 compacted joint-limit rows instead of 6 -- slot s =
    the s-th joint that is at a limit -- and a wave in which some lane has more joints at their limits runs the instantiation with one row per
    limited joint (the wave's vote, planar_kernel.hpp: topo_limit_slots); the Hopper's small tier carries the limit rows of its first two
    joints, uncompacted, and a wave in which some lane has the foot joint at a limit runs the tier with all three (topo_limit_prefix).  Two waves here: the first starts with EVERY joint of a few lanes beyond its limits (so that
    wave takes the all-limits tier while those lanes recover), the second with ordinary states (compacted rows, joints entering and leaving
    their limits from substep to substep: slots change owners, the warm sets must not follow them).  Both against the oracle, step by step."""
    card = card_for(env_id)
    n, T, nd = 128, 25, card.ndofs
    g = EmuStepper(card, n, precision=64, waves=True)
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(12)
    qn = rng.uniform(-0.02, 0.02, (n, nd)); vn = rng.uniform(-0.5, 0.5, (n, nd))
    lo = np.array([card.lower[d] for d in range(nd)]); hi = np.array([card.upper[d] for d in range(nd)])
    lim = np.array([bool(card.limited[d]) for d in range(nd)])
    init = np.array(card.init_pos[:nd])
    for e in range(0, 64, 5):            # every limited joint 0.02 rad beyond its upper (even lanes) / lower (odd) limit
        target = (hi + 0.02) if (e // 5) % 2 == 0 else (lo - 0.02)
        qn[e, lim] = (target - init)[lim]
    g.reset(None, qn, vn); ora.reset(None, qn, vn)
    worst = [0.0, 0.0]
    for t in range(T):
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        o, r, d, tr = g.step(a); oo, ro, do, to = ora.step(a)
        qg, dqg = g.get_state(); qo, dqo = ora.state()
        worst = [max(worst[0], np.abs(qg - qo).max()), max(worst[1], np.abs(dqg - dqo).max())]
        assert np.array_equal(d.astype(bool), np.asarray(do, bool)), t
        if np.any(do):
            qr = rng.uniform(-0.02, 0.02, (n, nd)); vr = rng.uniform(-0.5, 0.5, (n, nd))
            g.reset(np.asarray(do, np.uint8), qr, vr, want_obs=False); ora.reset(np.asarray(do, bool), qr, vr)
    g.close()
    assert worst[0] < 1e-9 and worst[1] < 1e-7, worst


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1", "DartHalfCheetah-v1"])
def test_limit_slot_vote_does_not_change_a_lanes_numbers(env_id):
    """Which limit layout a wave runs -- the compacted slots, or one row per limited joint because some lane has more joints at their limits than
    slots -- is the wave's vote; a lane's numbers must not depend on it, bitwise: the all-limits factorisation only adds exact zeros for the
    rows the compacted one leaves out, the rows keep their order, and the warm sets travel in joint layout.  The same 128 envs in two lane
    orders: in one, the envs with every joint beyond its limits share a wave with ordinary ones; in the other they are spread differently."""
    card = card_for(env_id)
    n, T, nd = 128, 20, card.ndofs
    rng = np.random.RandomState(31)
    qn = rng.uniform(-0.02, 0.02, (n, nd)); vn = rng.uniform(-0.5, 0.5, (n, nd))
    lo = np.array([card.lower[d] for d in range(nd)]); hi = np.array([card.upper[d] for d in range(nd)])
    lim = np.array([bool(card.limited[d]) for d in range(nd)]); init = np.array(card.init_pos[:nd])
    for e in range(0, 32, 3):            # eleven envs of the first half-wave: all limited joints beyond a limit
        qn[e, lim] = (((hi + 0.02) if e % 2 == 0 else (lo - 0.02)) - init)[lim]
    acts = rng.uniform(-1, 1, (T, n, card.act_dim)).astype(np.float32)
    out = []
    for order in (np.arange(n), np.random.RandomState(32).permutation(n)):
        g = EmuStepper(card, n, precision=64, waves=True)
        g.reset(None, qn[order], vn[order])
        for t in range(T):
            g.step(acts[t][order])
        q, dq = g.get_state()
        inv = np.empty(n, np.int64); inv[order] = np.arange(n)
        out.append((q[inv], dq[inv]))
        g.close()
    assert np.isfinite(out[0][0]).all() and np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_four_env_wave_solver_keeps_an_env_independent_of_its_wave_mates():
    """Every half cheetah beyond the two register slots is served by wave_constraints4, four envs per pass, one per row of 16 lanes.  Which
    row an env lands in and who shares the pass depends on its wave mates; its trajectory must not -- bitwise: the same 128 envs, shuffled
    across the two waves, give the same states.  (That the mode follows the oracle: test_whole_waves_follow_the_oracle, and below with
    every env on the floor.)"""
    card = card_for("DartHalfCheetah-v1")
    n, T, nd = 128, 30, card.ndofs
    rng = np.random.RandomState(2)
    qn = rng.uniform(-0.1, 0.1, (n, nd)); vn = rng.uniform(-0.1, 0.1, (n, nd))
    qn[:, 1] -= rng.uniform(0.0, 0.25, n)            # some start low: three and more capsules touching from the first steps on
    acts = rng.uniform(-1, 1, (T, n, card.act_dim)).astype(np.float32)
    perm = np.random.RandomState(3).permutation(n)
    out = []
    for order in (np.arange(n), perm):
        g = EmuStepper(card, n, precision=64, waves=True)
        g.reset(None, qn[order], vn[order])
        for t in range(T):
            g.step(acts[t][order])
        q, dq = g.get_state()
        inv = np.empty(n, np.int64); inv[order] = np.arange(n)
        out.append((q[inv], dq[inv]))
        g.close()
    assert np.isfinite(out[0][0]).all() and np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_wave_served_fallback_follows_the_oracle():
    """every touching half cheetah through wave_constraints (rows built by the lanes together, both pivoting stages in the register wave
    solver), as DART_CFG_DEBUG_FORCE_FALLBACK does on the GPU"""
    worst, most = rollout(card_for("DartHalfCheetah-v1"), 64, 25, force_fallback=True)
    assert worst[0] < 1e-9 and worst[1] < 1e-7 and most >= 3, (worst, most)


def test_wave_served_envs_report_their_contacts():
    """The reporting instantiation of wave_constraints4 (EXTRAS: each owner writes its env's contact records and constraint forces from the
    LDS block it was served in): every touching half cheetah through the four-env passes, contacts and J^T lambda / dt against the oracle's."""
    card = card_for("DartHalfCheetah-v1"); n = 64; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(6)
    g = EmuStepper(card, n, precision=64, waves=True); o = OracleBatch(card, n)
    g.force_slow(True)
    g.enable_contact_report(True)
    qn = rng.uniform(-.05, .05, (n, nd)); vn = rng.uniform(-.05, .05, (n, nd))
    g.reset(None, qn, vn); o.reset(None, qn, vn)
    seen = most = 0
    for t in range(25):
        a = rng.uniform(-1, 1, (n, na)).astype(np.float32)
        g.step(a); o.step(a)
        cnt, bod, pt, fc = g.contacts(); cf = g.constraint_forces()
        qg, dqg = g.get_state(); qo, dqo = o.state()
        assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-7, t
        for i, w in enumerate(o.worlds):
            rep = w.contact_report(); k = len(rep)
            assert cnt[i] == k, (t, i)
            assert np.abs(cf[i] - w.constraint_forces()).max() < 1e-6
            most = max(most, k)
            if k:
                seen += k
                assert np.array_equal(bod[i, :k, 0], rep[:, 0].astype(np.int32)) and np.all(bod[i, :k, 1] == -1)
                assert np.abs(pt[i, :k] - rep[:, 2:5]).max() < 1e-9 and np.abs(fc[i, :k] - rep[:, 5:8]).max() < 1e-6
    g.close()
    assert seen > 500 and most >= 3, (seen, most)


def test_fallen_user_models_on_whole_waves():
    """physics-only cards (no termination): the walker tree lies on up to seven capsules -- two register slots, beyond them the wave solvers
    (round 6: no second register tier any more); the pogo hopper keeps all four in its register tier"""
    wcard = build_card(load_model("walker2d"), None)
    wcard.frame_skip = 4
    worst, most = rollout(wcard, 64, 110, noise=0.01, scale=np.array([100, 100, 20, 100, 100, 20.0]), seed=3)
    assert most >= 4 and worst[0] < 1e-7 and worst[1] < 1e-5, (worst, most)      # four capsules: two more than the register slots hold -> served by the wave
    from dart_env_amd.skel import parse_skel
    from tests.pogo_env import SKEL as POGO, SCALE
    pcard = build_card(parse_skel(POGO), None)
    pcard.frame_skip = 4
    worst, most = rollout(pcard, 64, 120, noise=0.01, scale=np.asarray(SCALE, float), seed=4)
    assert most >= 3 and worst[0] < 1e-8 and worst[1] < 1e-6, (worst, most)


def test_a_wave_of_half_cheetahs_lying_on_the_floor_follows_the_oracle_and_its_envs_stay_independent():
    """ADVICE r5 (round 6): EVERY lane of the wave beyond the two register slots -- 16+ passes of the four-env solver per world step, the
    regime in which rounds 3-5's shared iteration budget ran out and left the late envs with a clamped, unconverged iterate that depended
    on their wave mates.  Now a solve's cap is its own (coop_iters): the wave follows the oracle, and the same envs in another lane order
    give bitwise the same states."""
    card = card_for("DartHalfCheetah-v1")
    n, T, nd, na = 64, 12, card.ndofs, card.act_dim
    rng = np.random.RandomState(21)
    q0 = rng.uniform(-0.05, 0.05, (n, nd)); v0 = rng.uniform(-0.2, 0.2, (n, nd))
    kind = np.arange(n) % 3
    q0[:, 1] += np.where(kind == 0, -0.25, -0.45)
    q0[:, 2] += np.where(kind == 2, 1.4, 0.0)
    acts = rng.uniform(-0.3, 0.3, (T, n, na)).astype(np.float32)
    from tests.oracle_lib import OracleWorld
    worlds = [OracleWorld(card) for _ in range(n)]
    for i, w in enumerate(worlds):
        w.set_state(q0[i], v0[i])
    outs, many = [], 0
    for order in (np.arange(n), np.random.RandomState(5).permutation(n)):
        g = EmuStepper(card, n, precision=64, waves=True)
        g.set_state(q0[order], v0[order])
        inv = np.empty(n, np.int64); inv[order] = np.arange(n)
        for t in range(T):
            g.step(acts[t][order])
            if order[0] == 0 and order[1] == 1:      # the index-order pass is the one held to the oracle, step by step
                for i, w in enumerate(worlds):
                    w.env_step(acts[t][i].astype(np.float64))
                many += sum(1 for w in worlds if len(w.last_contacts()) >= 3)
                qg, dqg = g.get_state()
                qo = np.stack([w.get_state()[0] for w in worlds]); dqo = np.stack([w.get_state()[1] for w in worlds])
                assert np.abs(qg - qo).max() < 1e-8 and np.abs(dqg - dqo).max() < 1e-6, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
        q, dq = g.get_state()
        outs.append((q[inv], dq[inv]))
        g.close()
    assert many > 0.8 * T * n, many
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


_POISON_LANE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper, poison_static_lds
assert poison_static_lds(waves=True) > 0
outs = []
for env_id, prec in (("DartHalfCheetah-v1", 32), ("DartHalfCheetah-v1", 64), ("DartWalker2d-v1", 64), ("DartHopper-v1", 64)):
    card = card_for(env_id); n = 64; nd, na = card.ndofs, card.act_dim
    rng = np.random.RandomState(5)
    q0 = rng.uniform(-0.3, 0.3, (n, nd)); dq0 = rng.uniform(-2, 2, (n, nd))
    q0[:, 1] = rng.uniform(-0.65, -0.3, n)      # from above the floor to lying in it: every register tier, the fallback solver, the hand-off
    g = EmuStepper(card, n, precision=prec, waves=True)
    g.set_state(q0, dq0)
    for t in range(3):
        ob, r, d, tr = g.step(rng.uniform(-1, 1, (n, na)).astype(np.float32))
        outs += [np.asarray(ob, np.float64).ravel(), np.asarray(r, np.float64).ravel(), np.asarray(d, np.float64).ravel()]
    outs += [np.asarray(x, np.float64).ravel() for x in g.get_state()]
    g.close()
np.save(sys.argv[2], np.concatenate(outs))
"""


def test_lane_kernel_results_do_not_depend_on_what_lds_held_before(tmp_path):
    """The lane kernels' static LDS (the fallback solver's block, the wave hand-off buffer) filled with a byte pattern before every
    workgroup (DART_EMU_POISON_LDS + emu_lib.poison_static_lds): NaNs / -1 or plausible numbers instead of zeros must not change a bit.
    (Written while looking for the cause of the half cheetah's first-launch difference on gfx950, DESIGN.md section 4.1: it is NOT a read of
    LDS the workgroup has not written.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "poison_lane.py"
    script.write_text(_POISON_LANE_SCRIPT)
    got = {}
    for tag, val in (("zero", "0x00"), ("nan", "0xff"), ("finite", "0x3f")):
        out = tmp_path / (tag + ".npy")
        subprocess.check_call([sys.executable, str(script), root, str(out)], env=dict(os.environ, DART_EMU_POISON_LDS=val), timeout=1500)
        got[tag] = np.load(out)
    assert np.all(np.isfinite(got["zero"]))
    assert np.array_equal(got["zero"], got["nan"]) and np.array_equal(got["zero"], got["finite"])
