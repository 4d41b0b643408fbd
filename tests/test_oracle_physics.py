"""Known-answer tests that pin the fp64 oracle WITHOUT DART: closed forms and invariants."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for, build_card, load_model
from tests.oracle_lib import OracleWorld


def _world(env_id="DartHopper-v1", **kw):
    return OracleWorld(card_for(env_id), **kw)


def test_total_mass_and_standing_force():
    w = _world()
    M = w.mass_matrix()
    assert M[0, 0] == pytest.approx(15.26499871, abs=1e-9) and M[1, 1] == pytest.approx(15.26499871, abs=1e-9)
    assert w.bias()[1] == pytest.approx(15.26499871 * 9.81, rel=1e-12)   # 149.75 N standing normal force
    assert card_for("DartWalker2d-v1").ndofs == 9
    assert OracleWorld(card_for("DartWalker2d-v1")).mass_matrix()[0, 0] == pytest.approx(22.69800692, abs=1e-8)


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_mass_matrix_symmetric_pd_and_rnea_columns(env_id):
    w = _world(env_id)
    rng = np.random.RandomState(0)
    n = w.n
    for _ in range(5):
        w.set_state(rng.uniform(-1, 1, n), rng.uniform(-2, 2, n))
        M = w.mass_matrix()
        assert np.allclose(M, M.T, atol=1e-12)
        assert np.linalg.eigvalsh(M).min() > 0
        for i in range(n):
            e = np.zeros(n); e[i] = 1
            assert np.allclose(w.inverse_dynamics(np.zeros(n), e, False), M[:, i], atol=1e-11)


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_dynamics_match_independent_lagrangian(env_id):
    """M and C from an independent numpy derivation: point-mass + inertia Lagrangian with numerical Jacobians."""
    model = load_model(card_for(env_id).name.decode() if False else {"DartHopper-v1": "hopper", "DartWalker2d-v1": "walker2d"}[env_id])
    w = _world(env_id)
    n = w.n
    rng = np.random.RandomState(1)

    def fk(q):  # planar forward kinematics straight from the card: world COM (x, y) and angle of every body
        ang = [0.0] * model.nbodies
        org = [None] * model.nbodies
        out = []
        for i, b in enumerate(model.bodies):
            p = b.parent
            pa = 0.0 if p < 0 else ang[p]
            po = np.zeros(2) if p < 0 else org[p]
            R = lambda a: np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            j = po + R(pa) @ b.T_pj[:2, 3]
            a = pa
            if b.jtype == 1:      # prismatic
                j = j + R(pa) @ (b.axes[0][:2] * q[b.dof_offset])
            elif b.jtype == 2:    # revolute about +-z
                a = pa + b.axes[0][2] * q[b.dof_offset]
            ang[i] = a
            org[i] = j - R(a) @ b.T_cj[:2, 3]
            out.append((org[i] + R(a) @ b.com[:2], a))
        return out

    def kinetic_M(q):
        eps = 1e-6
        M = np.zeros((n, n))
        base = fk(q)
        J = []
        for k in range(n):
            dq_ = np.zeros(n); dq_[k] = eps
            fp, fm = fk(q + dq_), fk(q - dq_)
            J.append([((fp[i][0] - fm[i][0]) / (2 * eps), (fp[i][1] - fm[i][1]) / (2 * eps)) for i in range(model.nbodies)])
        for i, b in enumerate(model.bodies):
            Jv = np.array([J[k][i][0] for k in range(n)]).T   # 2 x n
            Jw = np.array([J[k][i][1] for k in range(n)])     # n
            M += b.mass * Jv.T @ Jv + (b.inertia[2, 2] if b.mass > 0 else 0.0) * np.outer(Jw, Jw)
        return M

    def potential(q):
        return sum(b.mass * 9.81 * fk(q)[i][0][1] for i, b in enumerate(model.bodies))

    for _ in range(3):
        q = rng.uniform(-0.8, 0.8, n); dq = rng.uniform(-2, 2, n)
        w.set_state(q, dq)
        M = w.mass_matrix()
        assert np.allclose(M, kinetic_M(q), atol=2e-6)
        # C = dM/dt dq - d/dq (1/2 dq^T M dq) + dV/dq  by finite differences
        eps = 1e-5
        dM = [(kinetic_M(q + eps * np.eye(n)[k]) - kinetic_M(q - eps * np.eye(n)[k])) / (2 * eps) for k in range(n)]
        Mdot = sum(dM[k] * dq[k] for k in range(n))
        dT = np.array([0.5 * dq @ dM[k] @ dq for k in range(n)])
        dV = np.array([(potential(q + eps * np.eye(n)[k]) - potential(q - eps * np.eye(n)[k])) / (2 * eps) for k in range(n)])
        C = Mdot @ dq - dT + dV
        assert np.allclose(w.bias(), C, atol=5e-4), np.abs(w.bias() - C).max()


def test_free_fall_closed_form_until_first_contact():
    """q=dq=0: foot bottom starts 0.04 m above the floor; semi-implicit Euler gives y_k = -g dt^2 k(k+1)/2."""
    w = _world()
    w.reset()
    g, dt = 9.81, 0.002
    k = 0
    while True:
        w.step(); k += 1
        q, dq = w.get_state()
        if len(w.last_contacts()):   # (limit rows exist from step 1: q = 0 sits on the knee/hip upper limits)
            break
        assert q[1] == pytest.approx(-g * dt * dt * k * (k + 1) / 2, abs=1e-13)
        assert dq[1] == pytest.approx(-g * dt * k, abs=1e-13)
        assert np.abs(np.delete(q, 1)).max() < 1e-13
    # first contact happens on the step after y_k < -0.04:  k(k+1)/2 > 0.04/(g dt^2)  ->  k = 45, detected at k+1
    assert k == 46


def test_energy_conserved_without_damping_and_contacts():
    model = load_model("hopper")
    model.damping[:] = 0
    model.ground_y = -np.inf
    model.limited[:] = False
    rng = np.random.RandomState(3)
    q0, v0 = rng.uniform(-.3, .3, 6), rng.uniform(-1, 1, 6)
    drift = []
    for dt in (0.002, 0.001):   # first-order integrator: the energy error halves with dt
        model.dt = dt
        w = OracleWorld(build_card(model, None))
        w.set_state(q0, v0)
        e0 = w.energy()
        for _ in range(int(round(0.4 / dt))):
            w.step()
        drift.append(abs(w.energy() - e0))
    ke0 = 0.5 * v0 @ OracleWorld(build_card(model, None)).mass_matrix() @ v0
    assert drift[0] < 0.05 * (ke0 + 15.26 * 9.81 * 0.8)      # small against the energy being exchanged
    assert drift[1] < 0.7 * drift[0]


def test_single_pendulum_period():
    """Thigh swinging about the hip with everything else removed: small-angle period 2 pi sqrt(I / (m g l))."""
    model = load_model("hopper")
    model.damping[:] = 0
    model.ground_y = -np.inf
    model.limited[:] = False
    for b in model.bodies:
        if b.name == "h_pelvis":
            b.mass = 1e7; b.inertia = np.eye(3) * 1e7          # ~ fixed base
        elif b.name in ("h_shin", "h_foot"):
            b.mass = 1e-7; b.inertia = np.eye(3) * 1e-9        # negligible distal links
    w = OracleWorld(build_card(model, None))
    th0 = 0.01
    q = np.zeros(6); q[3] = th0
    w.set_state(q, np.zeros(6))
    b = [x for x in model.bodies if x.name == "h_thigh"][0]
    I = b.inertia[2, 2] + b.mass * 0.225 ** 2
    T = 2 * np.pi * np.sqrt(I / (b.mass * 9.81 * 0.225))
    model.gravity = np.array([0.0, -9.81, 0.0])
    # find first return to the starting side (zero crossing of dq from - to +  happens at T/2)
    prev = 0.0; t_half = None
    hold = np.zeros(6); hold[1] = model.total_mass * 9.81   # hold the (1e7 kg) base against gravity
    for k in range(int(2 * T / 0.002)):
        w.set_forces(hold)
        w.step()
        dq3 = w.dq[3] - w.dq[2] * 0   # joint velocity
        if prev < 0 and dq3 >= 0:
            t_half = (k + 1) * 0.002; break
        prev = dq3
    assert t_half == pytest.approx(T / 2, rel=0.02)


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_lcp_complementarity_and_pgs_limit(env_id):
    card = card_for(env_id)
    n = card.ndofs
    we = OracleWorld(card)
    wp = OracleWorld(card, OracleWorld.PGS, 4000, 4000)
    rng = np.random.RandomState(5)
    worst = 0.0
    seen = 0
    for ep in range(6):
        we.reset(); we.set_state(rng.uniform(-.005, .005, n), rng.uniform(-.005, .005, n))
        for t in range(120):
            tau = np.zeros(n); tau[3:] = rng.uniform(-1, 1, n - 3) * 20
            q, dq = we.get_state()
            wp.set_state(q, dq); wp.set_forces(tau); wp.step()
            we.set_forces(tau); we.step()
            lam, wv, lo, hi, res = we.last_lcp()
            if len(lam):
                seen += 1
                assert res < 1e-9                                        # w = A lam - b complementary to the box
                assert np.all(lam >= lo - 1e-12) and np.all(lam <= hi + 1e-12)
                worst = max(worst, np.abs(wp.dq - we.dq).max())          # PGS converges to the pivoting solution
    assert seen > 50
    assert worst < 1e-6, worst


def test_foot_only_contacts_equal_all_capsules_until_done():
    """BASELINE config[1] is 'no contacts beyond foot-ground'.  With random actions of scale >= 0.3 no other capsule
    reaches the floor before termination (measured 0 / 200 episodes, also for Walker2d); only slow collapses under
    tiny torques let the knee touch just before height < 0.7 ends the episode (known deviation)."""
    a_card = card_for("DartHopper-v1", all_bodies_collide=True)
    f_card = card_for("DartHopper-v1", all_bodies_collide=False)
    rng = np.random.RandomState(11)
    for ep in range(30):
        wa, wf = OracleWorld(a_card), OracleWorld(f_card)
        qn, vn = rng.uniform(-.005, .005, 6), rng.uniform(-.005, .005, 6)
        wa.set_state(qn, vn); wf.set_state(qn, vn)
        for t in range(200):
            a = rng.uniform(-1, 1, 3) * (0.3 if ep % 2 else 1.0)
            oa, ra, da = wa.env_step(a)
            of, rf, df = wf.env_step(a)
            assert da == df
            if da:
                break
            assert np.array_equal(oa, of) and ra == rf


# ------------------------------------------------------------------ link-link box contacts (Walker3d self-collision)
def _rot(axis, ang):
    a = np.asarray(axis, dtype=float); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_box_box_known_answers():
    from tests.oracle_lib import box_box
    I = np.eye(3)
    h = np.array([0.5, 0.5, 0.5])
    # separated -> nothing
    assert len(box_box([0, 0, 0], I, h, [1.2, 0, 0], I, h)[1]) == 0
    # face-face along x, overlap 0.1: normal +x (from 1 to 2), four points, depth 0.1, on the touching faces' overlap
    n, c = box_box([0, 0, 0], I, h, [0.9, 0.2, 0], I, h)
    assert np.allclose(n, [1, 0, 0]) and len(c) == 4 and np.allclose(c[:, 3], 0.1)
    assert np.allclose(sorted(c[:, 1]), [-0.3, -0.3, 0.5, 0.5]) and np.allclose(sorted(c[:, 2]), [-0.5, -0.5, 0.5, 0.5])
    # vertex of a tilted box 2 pushed 0.05 into the top face of box 1: one point, normal +y, depth 0.05
    R2 = _rot([1, 0, 1], np.arccos(1 / np.sqrt(3)))         # body diagonal along y: a vertex points down
    low = (R2 @ (np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * 0.5).T).T[:, 1].min()
    n, c = box_box([0, 0, 0], I, h, [0.1, 0.5 - low - 0.05, -0.1], R2, h)
    assert np.allclose(n, [0, 1, 0]) and len(c) == 1 and c[0, 3] == pytest.approx(0.05) and np.allclose(c[0, :3], [0.1, 0.45, -0.1], atol=1e-9)
    # crossed edges: box 2 rotated 45 deg about z and 45 deg about x sits edge-on above an edge of box 1 -> edge-edge, 1 point
    Ra = _rot([0, 0, 1], np.pi / 4); Rb = _rot([1, 0, 0], np.pi / 4)
    d = 0.5 * np.sqrt(2)
    n, c = box_box([0, 0, 0], Ra, h, [0, 2 * d - 0.02, 0], Rb, h)
    assert len(c) == 1 and np.allclose(n, [0, 1, 0], atol=1e-9) and c[0, 3] == pytest.approx(0.02, abs=5e-5) and np.allclose(c[0, :3], [0, d - 0.01, 0], atol=1e-9)   # ODE's fudge2 = 1e-5 on the edge axes
    # swapping the boxes flips the normal and keeps the depths
    n2, c2 = box_box([0.9, 0.2, 0], I, h, [0, 0, 0], I, h)
    assert np.allclose(n2, [-1, 0, 0]) and len(c2) == 4 and np.allclose(c2[:, 3], 0.1)


def test_walker3d_self_collision_keeps_the_legs_apart():
    """With card.self_collision the oracle generates link-link contacts: pressing the thighs together with constant hip
    torques stops at a few mm of overlap; without them the boxes pass through each other."""
    from tests.oracle_lib import box_box

    shp = [(3, [0, -0.225, 0], [0.05, 0.225, 0.05]), (4, [0, -0.25, 0], [0.04, 0.25, 0.04]), (5, [0.065, 0, 0], [0.1, 0.06, 0.06])]
    left = {3: 6, 4: 7, 5: 8}                              # thigh / shin / foot and their left twins (same boxes)

    def overlap_after(self_collision):
        card = card_for("DartWalker3d-v1")
        card.self_collision = int(self_collision)
        w = OracleWorld(card)
        w.reset()
        worst = 0.0
        for t in range(300):
            tau = np.zeros(21); tau[1] = 41.512 * 9.81          # hold the root up
            tau[11] = 15.0; tau[17] = -15.0                     # hip rotation about x (limits +-0.25 rad): squeeze the legs
            w.set_forces(tau); w.step()
            for b, off, h in shp:
                Ta, Tb = w.body_pose(b), w.body_pose(left[b])
                n, c = box_box(Ta[:3, 3] + Ta[:3, :3] @ off, Ta[:3, :3], h, Tb[:3, 3] + Tb[:3, :3] @ off, Tb[:3, :3], h)
                if len(c):
                    worst = max(worst, c[:, 3].max())
        return worst
    on, off = overlap_after(True), overlap_after(False)
    print("max leg-leg penetration with / without self-collision:", on, off)
    assert off > 0.03 and 0 < on < 0.01


# ------------------------------------------------------------------ DART FreeJoint root (dog.skel)
def _dog_card(gravity=True, ground=True):
    from dart_env_amd.model_card import build_card, load_model
    m = load_model("dog")
    if not gravity:
        m.gravity = np.zeros(3)
    if not ground:
        m.ground_y = -np.inf
    return build_card(m, None), m


def test_free_joint_coordinates_are_darts():
    """q[0:3] rotation vector, q[3:6] translation, dq[0:6] body-frame twist: pose and velocities of the trunk."""
    from scipy.spatial.transform import Rotation as Rot
    card, m = _dog_card()
    w = OracleWorld(card)
    rng = np.random.RandomState(0)
    for _ in range(20):
        q = np.zeros(22); dq = np.zeros(22)
        q[:3] = rng.uniform(-1.2, 1.2, 3); q[3:6] = rng.uniform(-1, 1, 3); q[6:] = rng.uniform(-.2, .2, 16)
        dq[:] = rng.uniform(-1, 1, 22)
        w.set_state(q, dq)
        T = w.body_pose(0)
        R = Rot.from_rotvec(q[:3]).as_matrix()
        assert np.allclose(T[:3, :3], R, atol=1e-12) and np.allclose(T[:3, 3], np.array([0, 1.25, 0]) + q[3:6], atol=1e-12)
        sv = w.body_com_spatial_velocity(0)
        assert np.allclose(sv[:3], R @ dq[:3], atol=1e-12) and np.allclose(sv[3:], R @ dq[3:6], atol=1e-12)   # trunk COM = frame origin
        q2, dq2 = w.get_state()
        assert np.array_equal(q2, q) and np.array_equal(dq2, dq)


def test_free_flight_conserves_momentum_to_first_order():
    """No floor: total linear momentum changes by m g t and the angular momentum about the COM stays put, both up to the
    O(dt) drift of integrating a BODY-FRAME twist the way DART does (Q <- Q exp(twist dt)) -- the drift halves with dt.
    Without the Jacobian-derivative terms of the coordinate change (-w x v, Edot rates) the drift would be O(1)."""
    def run(dt, steps):
        card, m = _dog_card(ground=False)
        card.dt = dt
        w = OracleWorld(card)
        rng = np.random.RandomState(1)
        q = np.zeros(22); dq = np.zeros(22)
        q[:3] = [0.3, -0.4, 0.2]; dq[:3] = [1.5, -0.8, 0.6]; dq[3:6] = [0.4, 1.0, -0.3]; dq[6:] = rng.uniform(-2, 2, 16)
        w.set_state(q, dq)
        masses = np.array([b.mass for b in m.bodies])
        inert = [np.asarray(b.inertia).reshape(3, 3) for b in m.bodies]

        def momenta():
            P = np.zeros(3); L = np.zeros(3); cs = []; vs = []
            for b in range(9):
                sv = w.body_com_spatial_velocity(b); c = w.body_com(b); R = w.body_pose(b)[:3, :3]
                P += masses[b] * sv[3:]; cs.append(c); vs.append(sv)
            com = sum(masses[b] * cs[b] for b in range(9)) / masses.sum()
            for b in range(9):
                R = w.body_pose(b)[:3, :3]
                L += R @ inert[b] @ R.T @ vs[b][:3] + masses[b] * np.cross(cs[b] - com, vs[b][3:])
            return P, L
        P0, L0 = momenta()
        for t in range(steps):
            tau = np.zeros(22); tau[6:] = 20 * np.sin(0.05 * t * dt / 0.002 + np.arange(16))
            w.set_forces(tau); w.step()
        P1, L1 = momenta()
        return P0, P1, L0, L1, masses.sum(), steps * dt
    P0, P1, L0, L1, M, T = run(0.002, 200)
    g = np.array([0, -M * 9.81 * T, 0])
    p1 = np.abs(P1 - P0 - g).max(); d1 = np.abs(L1 - L0).max()
    P0h, P1h, L0h, L1h, _, _ = run(0.001, 400)
    p2 = np.abs(P1h - P0h - g).max(); d2 = np.abs(L1h - L0h).max()
    print("linear momentum drift", p1, p2, "angular momentum drift", d1, d2, "|P0|", np.abs(P0).max(), "|L0|", np.abs(L0).max())
    assert p1 < 0.02 * np.abs(P0).max() and p2 < 0.65 * p1
    assert d1 < 0.05 * np.abs(L0).max() and d2 < 0.65 * d1


def test_free_root_dynamics_do_not_depend_on_the_orientation_chart():
    """No gravity, no floor: the body-frame twist and the joint trajectory must not depend on where the trunk points -- also
    when it points exactly along a singular direction of any fixed Euler chart (heading +-90 degrees, upside down): the
    chart of the internal root chain is re-centred on the current orientation before every world step."""
    from scipy.spatial.transform import Rotation as Rot
    card, m = _dog_card(gravity=False, ground=False)
    rng = np.random.RandomState(5)
    dq0 = rng.uniform(-2, 2, 22); qj = rng.uniform(-.3, .3, 16)

    def run(Q0):
        w = OracleWorld(card)
        q = np.zeros(22); q[:3] = Q0.as_rotvec(); q[3:6] = Q0.apply([0.3, -0.2, 0.5]); q[6:] = qj
        w.set_state(q, dq0)
        for t in range(150):
            tau = np.zeros(22); tau[6:] = 30 * np.sin(0.07 * t + np.arange(16))
            w.set_forces(tau); w.step()
        q1, dq1 = w.get_state()
        return Rot.from_rotvec(q1[:3]), q1[3:6], q1[6:], dq1
    R_ref, p_ref, qj_ref, dq_ref = run(Rot.identity())
    for Q0 in (Rot.from_euler("y", 90, degrees=True), Rot.from_euler("y", -90, degrees=True), Rot.from_euler("x", 180, degrees=True),
               Rot.from_euler("z", 90, degrees=True), Rot.from_rotvec([0.7, -2.1, 1.3])):
        R1, p1, qj1, dq1 = run(Q0)
        assert np.allclose(dq1, dq_ref, atol=1e-9) and np.allclose(qj1, qj_ref, atol=1e-10)
        assert np.allclose((Q0.inv() * R1).as_matrix(), R_ref.as_matrix(), atol=1e-10)
        assert np.allclose(Q0.inv().apply(p1), p_ref, atol=1e-10)


# ------------------------------------------------------------------ contact report (pydart2 collision_result.contacts)
@pytest.mark.parametrize("env_id,trans", [("DartHopper-v1", [0, 1]), ("DartHumanWalker-v1", [0, 1, 2]), ("DartWalker3d-v1", [0, 1, 2]),
                                          ("DartDog-v1", [3, 4, 5])])
def test_contact_report_forces_sum_to_the_root_constraint_force(env_id, trans):
    """The reported per-contact forces (n l_n + t1 l_1 + t2 l_2) / dt must add up to the generalized constraint force on the
    root translation dofs (J^T lambda / dt, assembled independently from the Jacobian rows; joint-limit rows do not touch
    those dofs, link-link contacts cancel in the sum), points must lie at the floor, bodies must be the colliding ones."""
    card = card_for(env_id)
    w = OracleWorld(card)
    rng = np.random.RandomState(0)
    w.reset()
    seen = pairs = 0
    axes = {3: [0, 1, 2], 2: [0, 1]}[len(trans)]
    for t in range(600):
        tau = np.zeros(w.n); tau[6 if w.n > 9 else 3:] = rng.uniform(-1, 1, w.n - (6 if w.n > 9 else 3)) * 30
        q_prev = w.q.copy()
        w.set_forces(tau); w.step()
        rep = w.contact_report()
        cf = w.constraint_forces()
        if env_id == "DartDog-v1":   # FreeJoint: generalized forces on the body-frame twist, tau_b = R^T f_world (R of the step's start)
            from scipy.spatial.transform import Rotation as Rot
            cf = cf.copy(); cf[3:6] = Rot.from_rotvec(q_prev[:3]).apply(cf[3:6])
        ground = rep[rep[:, 1] < 0]
        if len(rep) == 0:
            assert np.abs(cf[trans]).max() < 1e-9
            continue
        seen += 1
        pairs += int((rep[:, 1] >= 0).any())
        total = ground[:, 5:8].sum(axis=0) if len(ground) else np.zeros(3)
        assert np.allclose(total[axes], cf[trans], rtol=1e-9, atol=1e-7), (t, total, cf[trans])
        assert (ground[:, 6] >= -1e-9).all()                                   # the floor pushes up
        # contact points at the floor, within the penetration: DART corrects it at 1 mm/s at most (DART_MAX_ERV), so under
        # random 30 Nm slamming without termination the O(dt^2) drift of fast joint motion outruns it and reaches centimetres
        assert np.all(np.abs(ground[:, 3] - card.ground_y) < 0.25)
        nb = card.nbodies
        assert np.all((rep[:, 0] >= 0) & (rep[:, 0] < nb) & (rep[:, 1] < nb))
    assert seen > 100
    if env_id == "DartWalker3d-v1":
        print("world steps with a link-link contact:", pairs)


# ------------------------------------------------------------------ Coulomb friction known answer (own asset: tests/golden/assets/sled.skel)
def _sled_card():
    import os
    from dart_env_amd.skel import parse_skel
    m = parse_skel(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets", "sled.skel"))
    return build_card(m, None)      # four coplanar box-face contacts: regular thanks to DART's contact cfm of 1e-5


def sled_closed_form(steps, v0=1.0, mu=1.0, g=9.81, dt=0.002):
    """Flat box sliding on the floor: v_{k+1} = v_k - mu g dt until the friction impulse can stop it, then it sticks."""
    v, x, out = v0, 0.0, []
    for _ in range(steps):
        v = v - mu * g * dt if v > mu * g * dt else 0.0
        x += dt * v
        out.append((x, v))
    return np.array(out)


def test_sliding_box_decelerates_at_mu_g_and_sticks():
    """3.5 kg box (0.4 x 0.1 x 0.3) sliding at 1 m/s: the friction pyramid along DART's tangent t1 = z x n = -x gives exactly
    mu g; the two-stage bounds (+-mu * frictionless normal impulse, uniform m g dt / 4 per vertex) cap every vertex at the same
    friction while the final normals shift forward to balance the friction torque about the COM; after v / (mu g dt) = 51 steps
    the box sticks."""
    c = _sled_card()
    w = OracleWorld(c)
    w.set_state(np.zeros(3), np.array([1.0, 0.0, 0.0]))
    ref = sled_closed_form(80)
    m, g, mu, hl, hh = 3.5, 9.81, c.friction, 0.2, 0.05
    for k in range(80):
        w.set_forces(np.zeros(3)); w.step()
        assert abs(w.dq[0] - ref[k, 1]) < 2e-5 and abs(w.q[0] - ref[k, 0]) < 2e-5, k       # 1e-6 m penetration / ERP transient
        assert abs(w.q[1]) < 1e-6 and abs(w.q[2]) < 1e-6                                    # stays flat on the floor
        rep = w.contact_report()
        assert len(rep) == 4 and np.all(rep[:, 0] == 2) and np.all(rep[:, 1] == -1)
        assert abs(rep[:, 6].sum() - m * g) < 2e-2                                         # normals carry the weight
        if 1 <= k < 50:
            assert np.allclose(rep[:, 5], -mu * m * g / 4, rtol=2e-4)                      # every vertex at its stage-1 bound
            front = rep[rep[:, 2] > w.q[0]][:, 6].sum(); rear = rep[rep[:, 2] < w.q[0]][:, 6].sum()
            assert abs((front - rear) * hl - mu * m * g * hh) < 2e-2                       # torque balance about the COM
        if k > 52:
            assert np.abs(rep[:, 5]).max() < 1e-3 and w.dq[0] == pytest.approx(0.0, abs=1e-6)   # sticking: no friction needed


# ------------------------------------------------------------------ single-dof closed forms (own asset: tests/golden/assets/flywheel.skel)
def flywheel_card(damping=0.0, stiffness=0.0, rest=0.0, friction=0.0, lower=None, upper=None, dt=0.002):
    import os
    from dart_env_amd.skel import parse_skel
    m = parse_skel(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets", "flywheel.skel"), dt=dt)
    m.damping[:] = damping; m.stiffness[:] = stiffness; m.rest[:] = rest
    m.joint_friction = np.array([friction])
    if lower is not None or upper is not None:
        m.lower[:] = -np.inf if lower is None else lower
        m.upper[:] = np.inf if upper is None else upper
        m.limited[:] = True
    return build_card(m, None)


FLY_I = 0.5      # izz of the wheel about its axle


def flywheel_closed_forms(kind, steps, dt=0.002, impulse_inertia=0):
    """(q, dq) after every step of the 1-dof wheel, from the update rules DART's semantics imply (A3, A10, joint friction):
    implicit damping / spring  (I + dt d + dt^2 k) a = tau - d v - k (q + dt v - rest);   Coulomb friction: an impulse within
    +-mu dt that drives v to zero;   limit: inelastic stop while q >= upper (inclusive) and v > 0, no position correction.
    The impulse acts on I_imp = I (impulse_inertia = 0 = DART_IMPULSE_MASS: DART 6's impulse pass reads the non-implicit articulated inertia) or on
    I + dt d + dt^2 k (impulse_inertia = 1 = DART_IMPULSE_AUGMENTED).  A hard stop cannot tell the two apart (the row's own inertia cancels); a SATURATED
    friction impulse can: dv = mu dt / I_imp -- kind 'damped_friction'."""
    q, v, out = 0.0, 2.0, []
    d, k, rest, mu, up = dict(damping=(0.7, 0, 0, 0, None), spring=(0.3, 5.0, 0.1, 0, None), friction=(0, 0, 0, 0.2, None),
                              limit=(0, 0, 0, 0, 0.05), damped_friction=(25.0, 0, 0, 0.2, None))[kind]
    I_fd = FLY_I + dt * d + dt * dt * k
    I_imp = FLY_I if impulse_inertia == 0 else I_fd
    for _ in range(steps):
        a = (-d * v - k * (q + dt * v - rest)) / I_fd
        vs = v + dt * a
        if mu:
            imp = max(-mu * dt, min(mu * dt, -I_imp * vs / (1 + 1e-9)))   # row: (1 + cfm) imp / I_imp = -vs within +-mu dt
            vs = vs + imp / I_imp
        if up is not None and q >= up and vs > 0:
            vs = vs - vs / (1 + 1e-9)                                # (1 + cfm) on the diagonal leaves vs cfm / (1 + cfm)
        v = vs
        q = q + dt * v
        out.append((q, v))
    return np.array(out)


FLY_CARDS = dict(damping=lambda: flywheel_card(damping=0.7), spring=lambda: flywheel_card(damping=0.3, stiffness=5.0, rest=0.1),
                 friction=lambda: flywheel_card(friction=0.2), limit=lambda: flywheel_card(lower=-1.0, upper=0.05),
                 damped_friction=lambda: flywheel_card(damping=25.0, friction=0.2))


@pytest.mark.parametrize("impulse_inertia", [0, 1])
@pytest.mark.parametrize("kind", ["damping", "spring", "friction", "limit", "damped_friction"])
def test_single_dof_closed_forms(kind, impulse_inertia):
    card = FLY_CARDS[kind]()
    card.impulse_inertia = impulse_inertia
    w = OracleWorld(card)
    w.set_state(np.zeros(1), np.array([2.0]))
    ref = flywheel_closed_forms(kind, 3000, impulse_inertia=impulse_inertia)
    for t in range(3000):
        w.set_forces(np.zeros(1)); w.step()
        assert abs(w.q[0] - ref[t, 0]) < 1e-10 and abs(w.dq[0] - ref[t, 1]) < 1e-9, (kind, t, w.q[0], ref[t, 0], w.dq[0], ref[t, 1])
    if kind == "damping":
        assert ref[-1, 1] == pytest.approx(2.0 * (FLY_I / (FLY_I + 0.002 * 0.7)) ** 3000, rel=1e-12)
    if kind == "friction":
        assert abs(ref[-1, 1]) < 1e-12 and ref[2498, 1] > 0          # stops after v0 I / (mu dt) = 2500 steps, then stays
    if kind == "limit":
        assert abs(ref[-1, 1]) < 1e-8 and 0.05 <= ref[-1, 0] < 0.05 + 2.0 * 0.002   # stuck one step beyond the limit: no correction


def test_damped_wheel_with_saturated_friction_separates_the_two_impulse_inertias():
    """The closed form that tells the A3 settings apart: while the friction impulse is saturated the wheel loses
    mu dt / I per step under DART 6's rule and mu dt / (I + dt d) under the augmented one -- with d = 25 a 10 % difference
    in the friction deceleration, far above every tolerance of this suite (and of the north star's 1e-4)."""
    a = flywheel_closed_forms("damped_friction", 400, impulse_inertia=0)
    b = flywheel_closed_forms("damped_friction", 400, impulse_inertia=1)
    dt, d, mu = 0.002, 25.0, 0.2
    r = FLY_I / (FLY_I + dt * d)
    # one step from v0 = 2: v1 = v0 r - mu dt / I_imp
    assert a[0, 1] == pytest.approx(2.0 * r - mu * dt / FLY_I, rel=1e-13)
    assert b[0, 1] == pytest.approx(2.0 * r - mu * dt / (FLY_I + dt * d), rel=1e-13)
    assert abs(a[0, 1] - b[0, 1]) == pytest.approx(mu * dt * (1 / FLY_I - 1 / (FLY_I + dt * d)), rel=1e-9)
    assert np.abs(a[:20, 1] - b[:20, 1]).max() > 5e-4          # 20 steps in, the two rules are 7e-4 rad/s apart
    stop_a = int(np.argmax(np.abs(a[:, 1]) < 1e-6)); stop_b = int(np.argmax(np.abs(b[:, 1]) < 1e-6))
    assert 0 < stop_a < stop_b                                   # and the wheel comes to rest earlier under DART 6's rule


def test_impulse_pass_runs_on_the_mass_matrix_multi_dof():
    """A3 on a real model: the Hopper in the air at q = 0 sits on the (inclusive) upper limits of its thigh and knee joints, which
    carry damping 1.0.  One world step restated in numpy from the oracle's own M and c -- unconstrained velocity from
    (M + dt D)^-1, limit rows solved by enumeration of their active sets on A = J Minv J^T, velocity change Minv J^T lambda -- with
    Minv = M^-1 (DART 6) or (M + dt D)^-1 (knob 1).  The oracle must follow the setting it is given, and the two must differ."""
    import itertools
    res = {}
    for knob in (0, 1):
        card = card_for("DartHopper-v1")
        card.impulse_inertia = knob
        w = OracleWorld(card)
        n = card.ndofs
        rng = np.random.RandomState(3)
        q = np.zeros(n); q[1] = 0.3                          # lifted: no contact
        dq = rng.uniform(-1, 1, n); dq[3] = 1.5; dq[4] = 2.0   # driving both limited joints into their upper limit (0)
        tau = rng.uniform(-50, 50, n); tau[:3] = 0
        w.set_state(q, dq)
        M = w.mass_matrix(); c = w.bias()
        D = np.array([card.damping[i] for i in range(n)]); dt = card.dt
        H = M + dt * np.diag(D)
        vs = dq + dt * np.linalg.solve(H, tau - c - D * dq)
        Minv = np.linalg.inv(M if knob == 0 else H)
        rows = [i for i in range(n) if card.limited[i] and (q[i] <= card.lower[i] or q[i] >= card.upper[i])]
        assert rows == [3, 4]
        A = Minv[np.ix_(rows, rows)] * (np.ones((2, 2)) + np.eye(2) * card.cfm)
        b = -vs[rows]
        sol = None
        for act in itertools.product([0, 1], repeat=2):     # upper limits: lambda <= 0, w = A lambda - b <= 0 where lambda = 0
            idx = [k for k in range(2) if act[k]]
            lam = np.zeros(2)
            if idx:
                lam[idx] = np.linalg.solve(A[np.ix_(idx, idx)], b[idx])
            wv = A @ lam - b
            if all(lam[k] <= 1e-15 for k in idx) and all(wv[k] <= 1e-12 for k in range(2) if not act[k]):
                sol = lam
        assert sol is not None and (sol < 0).all()           # both limits push back
        v_new = vs + Minv[:, rows] @ sol
        w.set_forces(tau); w.step()
        assert np.abs(w.dq - v_new).max() < 1e-10, (knob, np.abs(w.dq - v_new).max())
        res[knob] = w.dq.copy()
    assert np.abs(res[1] - res[0]).max() > 1e-3              # percent-level on the joint velocities next to the limits


def test_landing_is_inelastic_and_penetration_recovers_at_the_capped_rate():
    """Box dropped from 1 mm: free fall until a vertex is at / below the floor, then the normal rows set the vertical velocity
    to the penetration-correction velocity min(erp depth / dt, max_erv) = min(5 depth, 1e-3 m/s) (DART ContactConstraint: ERP
    0.01, DART_MAX_ERV 1e-3) -- no bounce -- and the depth decays by exactly that much per step."""
    c = _sled_card()
    w = OracleWorld(c)
    q0 = np.array([0.0, 0.001, 0.0])
    w.set_state(q0, np.zeros(3))
    g, dt = 9.81, c.dt
    y, v, landed = 0.001, 0.0, False
    for k in range(400):
        w.set_forces(np.zeros(3)); w.step()
        depth = -y                                   # box bottom at y (q[1]) relative to the floor
        if depth >= 0:                               # contact rows exist (inclusive): v_n = correction velocity
            v = min(c.erp * depth / dt, c.max_erv)
            landed = True
        else:
            v = v - g * dt
        y = y + dt * v
        # the (1 + cfm) diagonal lets the contact give way by cfm * (velocity change): 2e-6 m/s at the landing, 2e-7 after
        assert abs(w.dq[1] - v) < 3e-6 and abs(w.q[1] - y) < 1e-8, (k, w.q[1], y, w.dq[1], v)
        y = w.q[1]; v = w.dq[1] if not landed else v
        assert abs(w.q[0]) < 1e-9 and abs(w.q[2]) < 1e-9
    assert landed and -2e-4 < w.q[1] < 0 and 0 < w.dq[1] <= 1e-3 + 1e-6


def test_free_root_mass_matrix_is_in_darts_coordinates():
    """pydart2's skel.M for a FreeJoint root is the inertia of DART's generalized velocities -- dq[0:6] = BODY-frame twist -- : the
    kinetic energy 1/2 dq^T M dq must equal the sum over the bodies of 1/2 m |v_com|^2 + 1/2 w^T I_world w, computed here from the
    oracle's world-frame body velocities (an independent route: Jacobians per body, no mass matrix).  The internal chain's M
    (world-frame root rates) fails this at any orientation but the identity."""
    card, m = _dog_card()
    w = OracleWorld(card)
    rng = np.random.RandomState(4)
    masses = np.array([b.mass for b in m.bodies])
    inert = [np.asarray(b.inertia).reshape(3, 3) for b in m.bodies]
    for k in range(12):
        q = np.zeros(22); q[:3] = rng.uniform(-2.5, 2.5, 3); q[3:6] = rng.uniform(-1, 1, 3); q[6:] = rng.uniform(-.4, .4, 16)
        dq = rng.uniform(-2, 2, 22)
        w.set_state(q, dq)
        M = w.mass_matrix()
        assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
        T = 0.0
        for b in range(len(m.bodies)):
            sv = w.body_com_spatial_velocity(b); R = w.body_pose(b)[:3, :3]
            T += 0.5 * masses[b] * sv[3:] @ sv[3:] + 0.5 * sv[:3] @ (R @ inert[b] @ R.T) @ sv[:3]
        assert 0.5 * dq @ M @ dq == pytest.approx(T, rel=1e-12)
    # and at a generic orientation the body-frame and the world-frame root blocks really differ
    assert np.abs(M[:3, 6:]).max() > 1e-3


def test_free_root_equation_of_motion_in_darts_coordinates():
    """M qdd + c = tau in DART's coordinates: in free flight (no floor), with the joint damping set to zero, one world step changes the
    velocities by exactly dt M^-1 (tau - c) -- M = skel.M, c = skel.c as the getters return them (body-frame twist accelerations for
    the root, the w x v and chart terms inside c)."""
    from dart_env_amd.model_card import build_card, load_model
    m = load_model("dog")
    m.ground_y = -np.inf
    m.damping[:] = 0; m.stiffness[:] = 0; m.limited[:] = False
    card = build_card(m, None)
    w = OracleWorld(card)
    rng = np.random.RandomState(6)
    for k in range(6):
        q = np.zeros(22); q[:3] = rng.uniform(-2, 2, 3); q[3:6] = rng.uniform(-1, 1, 3); q[6:] = rng.uniform(-.3, .3, 16)
        dq = rng.uniform(-3, 3, 22)
        tau = np.zeros(22); tau[6:] = rng.uniform(-30, 30, 16)
        w.set_state(q, dq)
        M, c = w.mass_matrix(), w.bias()
        w.set_forces(tau); w.step()
        _, dq1 = w.get_state()
        assert np.abs((dq1 - dq) / card.dt - np.linalg.solve(M, tau - c)).max() < 1e-8 * (1 + np.abs(c).max())
