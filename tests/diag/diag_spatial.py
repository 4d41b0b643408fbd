"""Diagnostic (GPU box): spatial kernel vs oracle, step by step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd.model_card import build_card, load_model
from dart_env_amd.stepper import HipStepper
from tests.oracle_lib import OracleWorld
np.set_printoptions(precision=4, linewidth=220, suppress=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "nocontact"
m = load_model("humanwalker")
for s in m.shapes: s.collidable = m.bodies[s.body].name in ("l-foot", "r-foot")
if mode == "nocontact": m.ground_y = -np.inf
if mode == "capfeet":
    m.limited[:] = False
    from dart_env_amd.skel import SH_CAPSULE
    for s in m.shapes:
        if s.collidable: s.kind = SH_CAPSULE; s.size = np.array([0.05, 0.12, 0.0])
if mode == "mu0": m.friction = 0.0; m.limited[:] = False
if mode == "onefoot":
    m.limited[:] = False
    for s in m.shapes: s.collidable = m.bodies[s.body].name == "l-foot"
if mode == "onefoot_mu0":
    m.limited[:] = False; m.friction = 0.0
    for s in m.shapes: s.collidable = m.bodies[s.body].name == "l-foot"
if mode in ("nolimit", "nocontact_nolimit"):
    m.limited[:] = False
    if mode == "nocontact_nolimit": m.ground_y = -np.inf
card = build_card(m, None)
card.contact_cfm = float(os.environ.get('CCFM', '1e-4'))
n, nd = int(os.environ.get('NENV','4')), card.ndofs
rng = np.random.RandomState(0)
gpu = HipStepper(card, n, precision=64)
from dart_env_amd import stepper as st
gpu.configure(st.CFG_STATS, 1)
worlds = [OracleWorld(card) for _ in range(n)]
q0 = rng.uniform(-.05, .05, (n, nd)); v0 = rng.uniform(-.3, .3, (n, nd)); q0[:, 1] -= 0.045
gpu.set_state(q0, v0)
for i, w in enumerate(worlds): w.set_state(q0[i], v0[i])
for t in range(int(os.environ.get('STEPS','12'))):
    tau = (rng.uniform(-1, 1, (n, nd)) * 20).astype(np.float32); tau[:, :6] = 0
    gpu.step(tau)
    rows = []
    for i, w in enumerate(worlds):
        w.set_forces(tau[i].astype(np.float64)); w.step(); rows.append((len(w.last_lcp()[0]), len(w.last_contacts())))
    qg, dqg = gpu.get_state()
    qo = np.stack([w.q for w in worlds]); dqo = np.stack([w.dq for w in worlds])
    e = np.abs(dqg - dqo)
    if e.max() > 1e-9 or t % 10 == 0: print(t, "max|dq err|", np.abs(qg - qo).max(), "max|ddq err|", e.max(), "rows,contacts", rows, "worst dof", np.unravel_index(e.argmax(), e.shape))
    if e.max() > 1e-3:
        i = np.unravel_index(e.argmax(), e.shape)[0]
        print("  gpu dq", dqg[i]); print("  ora dq", dqo[i])
        D = gpu.debug_dump()[i]; mm = int(D[0])
        lam, wv, lo, hi, res = worlds[i].last_lcp()
        print("  gpu m,ncp", D[0], D[1], "oracle m", len(lam))
        print("  gpu x  ", D[2:2+mm]); print("  ora lam", lam)
        print("  gpu hi ", D[82:82+mm]); print("  ora hi ", hi)
        print("  gpu b  ", D[42:42+mm])
        A, b = worlds[i].last_Ab()
        def kkt(x):
            wv = A @ x - b; r = 0.0
            for k in range(len(x)):
                if lo[k] == hi[k]: continue
                if x[k] <= lo[k] + 1e-12: r = max(r, max(0, -wv[k]))
                elif x[k] >= hi[k] - 1e-12: r = max(r, max(0, wv[k]))
                else: r = max(r, abs(wv[k]))
            return r, wv
        print("  b diff", np.abs(b - D[42:42+mm]).max(), "diagA diff", np.abs(np.diag(A) - D[122:122+mm]).max(), "eig min", np.linalg.eigvalsh(A).min())
        print("  KKT residual oracle", kkt(lam)[0], "gpu", kkt(D[2:2+mm])[0]); print("  w(gpu x)", kkt(D[2:2+mm])[1])
        break
