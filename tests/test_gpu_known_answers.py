"""Closed forms and invariants checked on the HIP kernels themselves (through the C ABI), with no oracle in between:
what pins the oracle on the CPU (tests/test_oracle_physics.py) also pins the product path directly."""
import numpy as np
import pytest

from dart_env_amd.model_card import build_card, card_for, load_model

pytestmark = pytest.mark.gpu


def test_free_fall_closed_form_until_first_contact_planar_and_tree_kernel():
    """q = dq = 0, zero actions: the foot bottom starts 0.04 m above the floor and semi-implicit Euler gives
    y_k = -g dt^2 k (k + 1) / 2, v_k = -g dt k for the 45 substeps before the first contact."""
    from dart_env_amd.stepper import HipStepper
    g, dt = 9.81, 0.002
    for generic in (False, True):
        for prec, tol in ((64, 1e-12), (32, 2e-6)):
            card = card_for("DartHopper-v1", generic_kernel=generic)
            s = HipStepper(card, 3, precision=prec)
            s.reset(None, np.zeros((3, 6)), np.zeros((3, 6)))
            for j in range(1, 12):                    # 11 env-steps = 44 substeps, all before the contact
                s.step(np.zeros((3, 3), dtype=np.float32))
                k = 4 * j
                q, dq = s.get_state()
                assert np.abs(q[:, 1] + g * dt * dt * k * (k + 1) / 2).max() < tol and np.abs(dq[:, 1] + g * dt * k).max() < tol * 100
                assert np.abs(np.delete(q, 1, axis=1)).max() < tol
            s.step(np.zeros((3, 3), dtype=np.float32))   # substeps 45-48: contact from 46 on -> no longer in free fall
            q, dq = s.get_state()
            assert np.all(dq[:, 1] > -g * dt * 48 + 0.05)
            s.close()


def test_total_mass_and_standing_force_from_the_dynamics_getter():
    from dart_env_amd.stepper import HipStepper
    for env_id, mass in (("DartHopper-v1", 15.26499871), ("DartWalker2d-v1", 22.69800692)):
        s = HipStepper(card_for(env_id), 2, precision=64)
        M, c = s.dynamics()
        assert np.allclose(M[:, 0, 0], mass, atol=1e-8) and np.allclose(M[:, 1, 1], mass, atol=1e-8)
        assert np.allclose(c[:, 1], mass * 9.81, rtol=1e-12)        # hopper: the 149.75 N standing normal force
        assert np.allclose(M, np.swapaxes(M, 1, 2), atol=1e-12) and np.all(np.linalg.eigvalsh(M) > 0)
        s.close()


def test_energy_drift_is_first_order_without_damping_and_contacts():
    """Hopper with damping, limits and floor removed: E = dq^T M dq / 2 + sum m g com_y from the dynamics and body-pose
    getters drifts by O(dt) only (semi-implicit Euler) -- the drift halves with dt."""
    from dart_env_amd.stepper import HipStepper
    model = load_model("hopper")
    model.damping[:] = 0
    model.ground_y = -np.inf
    model.limited[:] = False
    masses = np.array([b.mass for b in model.bodies])
    rng = np.random.RandomState(3)
    n = 8
    q0, v0 = rng.uniform(-.3, .3, (n, 6)), rng.uniform(-1, 1, (n, 6))

    def energy(s):
        M, _ = s.dynamics(True, False)
        q, dq = s.get_state()
        _, _, com = s.body_poses()
        return 0.5 * np.einsum("ni,nij,nj->n", dq, M, dq) + 9.81 * (com[:, :, 1] * masses).sum(axis=1)
    drift = []
    for dt in (0.002, 0.001):
        model.dt = dt
        card = build_card(model, None)
        card.generic_kernel = 1
        s = HipStepper(card, n, precision=64)
        s.set_state(q0, v0)
        e0 = energy(s)
        for _ in range(int(round(0.4 / dt))):
            s.step(np.zeros((n, 6), dtype=np.float32))
        drift.append(np.abs(energy(s) - e0))
        s.close()
    assert np.all(drift[0] < 0.05 * (5.0 + 15.26 * 9.81 * 0.8)) and np.all(drift[1] < 0.7 * drift[0] + 1e-9)


def test_sled_translation_and_batch_invariance_fp32_bitwise():
    """Envs that differ only by a multiple-of-2^-10 x offset produce bitwise identical velocities in fp32 (the root
    translation is factored out of the dynamics), whatever the batch size."""
    from dart_env_amd.stepper import HipStepper
    card = card_for("DartHalfCheetah-v1")
    rng = np.random.RandomState(9)
    a = rng.uniform(-1, 1, (30, 1, card.act_dim)).astype(np.float32)
    outs = []
    for n, off in ((1, 0.0), (7, 16.0), (130, -64.0)):
        s = HipStepper(card, n, precision=32)
        q0 = np.zeros((n, card.ndofs)); q0[:, 0] = off
        s.set_state(q0, np.zeros((n, card.ndofs)))
        for t in range(30):
            s.step(np.repeat(a[t], n, axis=0))
        q, dq = s.get_state()
        assert np.all(q == q[0]) and np.all(dq == dq[0])          # every env of the batch identical
        outs.append((q[0].copy(), dq[0].copy(), off))
        s.close()
    for q, dq, off in outs[1:]:
        assert np.array_equal(dq, outs[0][1]) and np.array_equal(q[1:], outs[0][0][1:]) and abs((q[0] - off) - outs[0][0][0]) < 1e-5


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartHumanWalker-v1", "DartWalker3dSPD-v1", "DartReacher-v1"])
def test_snapshot_restore_resumes_bitwise(env_id):
    """dart_snapshot / dart_restore: after a restore the next steps -- including on-device auto-resets from the MT19937
    bank, TimeLimit truncations, episode statistics and the SPD controller's carried constraint forces -- are bitwise those
    of the uninterrupted run; a snapshot of a different configuration is refused."""
    import dart_env_amd
    from dart_env_amd.stepper import StepperError
    n = 64
    kw = {}
    venv = dart_env_amd.vector.make(env_id, n, **kw)
    venv.seed(7); venv.reset()
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, (70, n, venv.single_action_space.shape[0])).astype(np.float32)
    for t in range(20):
        venv.step(acts[t])
    snap = venv.env.snapshot() if hasattr(venv, "env") else venv._env.snapshot()
    ref = [venv.step(acts[t])[:3] for t in range(20, 70)]
    if env_id != "DartWalker3dSPD-v1":
        assert sum(int(r[2].sum()) for r in ref) > 0                  # episodes ended and were reset on the device
    for t in range(5):
        venv.step(acts[t])                                            # wander off, then come back
    (venv.env if hasattr(venv, "env") else venv._env).restore(snap)
    for t in range(20, 70):
        ob, r, d, _ = venv.step(acts[t])
        assert np.array_equal(ob, ref[t - 20][0]) and np.array_equal(r, ref[t - 20][1]) and np.array_equal(d, ref[t - 20][2]), t
    other = dart_env_amd.vector.make(env_id, n // 2, **kw)
    with pytest.raises(StepperError):
        (other.env if hasattr(other, "env") else other._env).restore(snap)
    other.close()
    # same size in bytes, different identity: another precision of half the batch (fp64 state of n / 2 envs = fp32 state of n) or
    # another Philox shard offset is refused by the snapshot header, not restored silently
    if env_id == "DartHopper-v1":
        from dart_env_amd.stepper import HipStepper, CFG_ENV_OFFSET
        from dart_env_amd.model_card import card_for
        a = HipStepper(card_for(env_id), 32, precision=64); b = HipStepper(card_for(env_id), 32, precision=64)
        b.configure(CFG_ENV_OFFSET, 32)
        sa = a.snapshot()
        b.configure(CFG_ENV_OFFSET, 32)
        with pytest.raises(StepperError):
            b.restore(sa)
        b.configure(CFG_ENV_OFFSET, 0)
        b.restore(sa)
        a.close(); b.close()
    venv.close()


@pytest.mark.parametrize("kind,impulse_inertia", [("damping", 0), ("spring", 0), ("friction", 0), ("limit", 0),
                                                  ("damped_friction", 0), ("damped_friction", 1), ("spring", 1)])
def test_single_dof_closed_forms_on_the_kernel(kind, impulse_inertia):
    """Implicit joint damping / spring, Coulomb joint friction and an inelastic joint limit on a 1-dof wheel: the kernel against
    the update rules themselves (tests/test_oracle_physics.py::flywheel_closed_forms), fp64 and fp32, no oracle in between.
    'damped_friction' is the closed form that separates the two settings of card.impulse_inertia (A3): a saturated friction
    impulse changes the velocity by mu dt / I under DART 6's rule and by mu dt / (I + dt d) under the other."""
    from dart_env_amd.stepper import HipStepper
    from tests.test_oracle_physics import FLY_CARDS, flywheel_closed_forms
    card = FLY_CARDS[kind]()
    card.impulse_inertia = impulse_inertia
    steps = 2700 if kind == "friction" else 600
    card.frame_skip = 20                                   # 20 world steps per launch
    ref = flywheel_closed_forms(kind, steps, impulse_inertia=impulse_inertia)
    for prec, tq, tv in ((64, 1e-10, 1e-9), (32, 2e-4, 2e-4)):
        s = HipStepper(card, 3, precision=prec)
        s.set_state(np.zeros((3, 1)), np.full((3, 1), 2.0))
        for j in range(steps // 20):
            s.step(np.zeros((3, 1), dtype=np.float32))
            q, dq = s.get_state()
            k = 20 * (j + 1) - 1
            assert np.abs(q[:, 0] - ref[k, 0]).max() < tq * (1 + abs(ref[k, 0])) and np.abs(dq[:, 0] - ref[k, 1]).max() < tv, (kind, prec, k)
        s.close()


def test_landing_is_inelastic_and_recovers_at_the_capped_rate_on_the_kernel():
    """Box dropped from 1 mm (tests/test_oracle_physics.py: same closed form): after the landing step the vertical velocity is
    the penetration-correction velocity min(erp depth / dt, max_erv), and the depth decays by exactly that per step."""
    from dart_env_amd.stepper import HipStepper
    from tests.test_oracle_physics import _sled_card
    c = _sled_card()
    s = HipStepper(c, 2, precision=64)
    s.set_state(np.tile([0.0, 0.001, 0.0], (2, 1)), np.zeros((2, 3)))
    g, dt = 9.81, c.dt
    y, v, landed = 0.001, 0.0, False
    for k in range(300):
        s.step(np.zeros((2, 3), dtype=np.float32))
        q, dq = s.get_state()
        depth = -y
        if depth >= 0:
            v = min(c.erp * depth / dt, c.max_erv); landed = True
        else:
            v = v - g * dt
        y = y + dt * v
        assert np.abs(dq[:, 1] - v).max() < 3e-6 and np.abs(q[:, 1] - y).max() < 1e-8, (k, q[:, 1], y, dq[:, 1], v)
        y = q[0, 1]
        if not landed:
            v = dq[0, 1]
    assert landed and -2e-4 < y < 0
    s.close()
