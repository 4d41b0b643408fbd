/* dart_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of what one `dart_world.step()` does for the Dart envs
 * (reference call site gym/envs/dart/dart_env.py:170-175) and of the task
 * epilogues wrapped around it (hopper.py:24-74, walker2d.py:22-74).
 *
 * PARITY UNPINNED.  The arithmetic of World::step lives in pydart2 -> DART
 * (C++, libdart6), an un-vendored, un-pinned third-party dependency that is
 * absent from /root/reference and from this image (reference tox.ini:22,52-55
 * and test.dockerfile:27-29 are its only, commented-out, mentions).  The
 * reference holds no golden vector for this path (gym/envs/tests/rollout.json
 * is `{}`; test_envs_semantics.py:68-70 is disabled).  This file therefore
 * restates DART 6's *published algorithm* (SURVEY.md Appendix B):
 *
 *   1. forward dynamics with implicit joint damping/springs:
 *        (M + dt D + dt^2 K) qdd = tau - C(q,qd) - D qd - K (q + dt qd - rest)
 *      (DART GenericJoint::updateTotalForceDynamics / updateInvProjArtInertiaImplicit)
 *   2. qd* = qd + dt qdd                        (Skeleton::integrateVelocities)
 *   3. constraints detected at q_t: capsule/ground contacts (ODE capsule-box:
 *      one contact at the lowest segment endpoint), joint limits (inclusive)
 *   4. boxed LCP  A = J M^-1 J^T (1+cfm on diag; card.impulse_inertia = DART_IMPULSE_AUGMENTED: H^-1, see oracle_step), b = -J qd* + erp*depth/dt,
 *      friction rows bounded by +-mu * (frictionless normal impulse) exactly as
 *      the ODE Dantzig driver DART calls sets lo/hi when it reaches the first
 *      findex row (two-stage solve)
 *   5. qd = qd* + M^-1 J^T lambda ; q += dt qd  (Skeleton::computeImpulseForwardDynamics, integratePositions)
 *      constants: ContactConstraint.cpp ERP 0.01 / MAX_ERV 1e-3 / CFM 1e-5, JointLimitConstraint.cpp CFM 1e-9 with an
 *      error allowance of 0 (no position correction), JointCoulombFrictionConstraint.cpp CFM 1e-9 -- card knobs
 *
 * It is pinned only by known-answer tests (tests/test_oracle_physics.py):
 * closed-form free fall, total mass / standing normal force, M symmetric PD and
 * M[:,i] = RNEA(q,0,e_i) - RNEA(q,0,0), an independent Lagrangian derivation in
 * numpy, energy drift, LCP complementarity, PGS -> exact-solver limit.
 *
 * Formulation (deliberately different from the GPU kernel so the two check each
 * other): full 3-D Featherstone spatial algebra in body coordinates, every
 * multi-dof joint expanded into a chain of 1-dof links with massless carriers,
 * dense Cholesky, block-principal-pivoting exact BLCP solver and a fixed-count
 * PGS that mirrors the device iteration order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dart_model_card.h"

#define MAXL 96  /* internal 1-dof links */
#define MAXN DART_MAX_DOFS
#define MAXC 64 /* contact points (a box can give 4) */
#define MAXM (3 * MAXC + MAXN)

typedef struct {
  double E[9]; /* rotation parent->child coords (rows = child axes in parent coords) */
  double r[3]; /* child origin in parent coords */
} Xform;

typedef struct {
  int parent;      /* link index, -1 world */
  int jtype;       /* weld / prismatic / revolute */
  int dof;         /* index into q, -1 for weld */
  double axis[3];  /* joint frame */
  double Tpre[16]; /* joint frame in parent link frame */
  double Tpost[16];/* joint frame in child link frame */
  double S[6];     /* motion subspace in child link frame [w; v], constant */
  double I[36];    /* spatial inertia in link frame */
  double mass, com[3];
} Link;

typedef struct OracleWorld {
  DartModelCard card;
  int nl, n;
  Link L[MAXL];
  int body_link[DART_MAX_BODIES];
  /* state (public coordinates: what pydart2's skel.q / skel.dq hold) */
  double q[MAXN], dq[MAXN], tau[MAXN];
  /* DART FreeJoint root (dog.skel): q[0:3] = rotation vector, q[3:6] = translation, dq[0:6] = twist of the child joint frame
   * expressed in that frame [w; v]; positions integrate as Q <- Q * [exp(w dt), v dt].  The dynamics run on an internal chain
   * of 1-dof links (translation x y z, then rotations about x y z in a chart re-centred on the current orientation) whose coordinates
   * qi / dqi are re-derived from (q, dq) before every use; velocities are mapped back with the exact instantaneous Jacobian, so only the parametrisation differs from DART. */
  int free_root, free_rot_link;
  double qi[MAXN], dqi[MAXN], R0[9], free_Tpost[16];
  double time;
  /* solver options */
  int solver;       /* 0 = exact (block principal pivoting), 1 = PGS fixed count */
  int pgs_k1, pgs_k2;
  int planar_drop_z;/* drop friction rows whose A_ii == 0 */
  /* alternatives to two DART-semantic assumptions that are not card fields (tools/knob_sensitivity.py measures
   * how far each moves a trajectory -- where a future capture from real DART should look first):
   *   a5_surface_point  1: a capsule's contact point sits on the capsule surface (pe - r n) instead of ODE's sphere-sphere
   *                        midpoint of the penetration (pe - n (r + d) / 2)
   *   a7_iterate_bounds 1: the friction bounds +-mu lambda_n are re-evaluated from the latest normal impulses until they stop
   *                        moving (a consistent pyramid) instead of being fixed once from the frictionless solve (ODE's driver) */
  int a5_surface_point, a7_iterate_bounds;
  /* scratch / last-step diagnostics */
  Xform X[MAXL];
  double W[MAXL][16]; /* world pose of each link */
  double M[MAXN * MAXN], C[MAXN];
  int m_last;
  double lambda_last[MAXM], w_last[MAXM], lo_last[MAXM], hi_last[MAXM];
  int ncontacts_last;
  double contact_last[MAXC][8]; /* body, px,py,pz, depth, fn, ft1, ft2 */
  double contact_dirs[MAXC][9];  /* normal, t1, t2 of each contact of the last world step */
  double contact_report[MAXC][8]; /* body a, body b (-1: ground), point, force on body a -- pydart2 Contact.point / .force */
  double lcp_residual_last;
  double A_last[MAXM * MAXM], b_last[MAXM]; /* debug copies of the last LCP */
  double init_height; /* human_walker.py:163 head COM height right after reset_model's set_state */
  double cf_last[MAXN]; /* generalized constraint forces of the last world step: J^T lambda / dt (pydart2 constraint_forces()) */
  double task_state[4]; /* per-env task state beyond (q, dq): the reach target of the reacher envs */
  int ext_all;        /* 1: ext_fb holds one world-frame force per body (snake fluid model), applied at the body origins */
  double ext_fb[DART_MAX_BODIES][3];
  int ext_body;       /* -1: none.  bodynodes[ext_body].add_ext_force(ext_f) before every world step (dart_env.py:170-172) */
  double ext_f[3];    /* world-frame force applied at the body frame origin (pydart2's default offset) */
} OracleWorld;

/* ------------------------------------------------------------------ small linear algebra */
static void mat4_mul(const double* A, const double* B, double* C) {
  double T[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j];
      T[4 * i + j] = s;
    }
  memcpy(C, T, sizeof T);
}
static void mat4_inv_rigid(const double* A, double* B) {
  double T[16] = {0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[4 * i + j] = A[4 * j + i];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) s += A[4 * j + i] * A[4 * j + 3];
    T[4 * i + 3] = -s;
  }
  T[15] = 1;
  memcpy(B, T, sizeof T);
}
static void mat4_identity(double* A) {
  memset(A, 0, 16 * sizeof(double));
  A[0] = A[5] = A[10] = A[15] = 1;
}
static void cross3(const double* a, const double* b, double* c) {
  double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
  c[0] = t0; c[1] = t1; c[2] = t2;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Rodrigues rotation about unit axis a by angle t -> 3x3 row-major */
static void rot_axis(const double* a, double t, double* R) {
  double c = cos(t), s = sin(t), v = 1 - c;
  R[0] = a[0] * a[0] * v + c;        R[1] = a[0] * a[1] * v - a[2] * s; R[2] = a[0] * a[2] * v + a[1] * s;
  R[3] = a[1] * a[0] * v + a[2] * s; R[4] = a[1] * a[1] * v + c;        R[5] = a[1] * a[2] * v - a[0] * s;
  R[6] = a[2] * a[0] * v - a[1] * s; R[7] = a[2] * a[1] * v + a[0] * s; R[8] = a[2] * a[2] * v + c;
}

/* spatial motion transform: vc = X vp */
static void xf_motion(const Xform* X, const double* vp, double* vc) {
  double t[3], u[3];
  cross3(vp, X->r, t); /* w x r */
  for (int i = 0; i < 3; i++) u[i] = vp[3 + i] + t[i];
  for (int i = 0; i < 3; i++) {
    vc[i] = X->E[3 * i] * vp[0] + X->E[3 * i + 1] * vp[1] + X->E[3 * i + 2] * vp[2];
    vc[3 + i] = X->E[3 * i] * u[0] + X->E[3 * i + 1] * u[1] + X->E[3 * i + 2] * u[2];
  }
}
/* spatial force transform child->parent: fp = X^T fc */
static void xf_force_T(const Xform* X, const double* fc, double* fp) {
  double n[3], f[3], t[3];
  for (int i = 0; i < 3; i++) {
    n[i] = X->E[i] * fc[0] + X->E[3 + i] * fc[1] + X->E[6 + i] * fc[2];
    f[i] = X->E[i] * fc[3] + X->E[3 + i] * fc[4] + X->E[6 + i] * fc[5];
  }
  cross3(X->r, f, t);
  for (int i = 0; i < 3; i++) { fp[i] = n[i] + t[i]; fp[3 + i] = f[i]; }
}
/* 6x6 of X (motion) */
static void xf_matrix(const Xform* X, double* M6) {
  /* [E 0; -E rx  E] */
  double rx[9] = {0, -X->r[2], X->r[1], X->r[2], 0, -X->r[0], -X->r[1], X->r[0], 0};
  memset(M6, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      M6[6 * i + j] = X->E[3 * i + j];
      M6[6 * (3 + i) + 3 + j] = X->E[3 * i + j];
      double s = 0;
      for (int k = 0; k < 3; k++) s += X->E[3 * i + k] * rx[3 * k + j];
      M6[6 * (3 + i) + j] = -s;
    }
}
static void mat6_XtIX(const double* X6, const double* I6, double* out) {
  double T[36];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += I6[6 * i + k] * X6[6 * k + j];
      T[6 * i + j] = s;
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += X6[6 * k + i] * T[6 * k + j];
      out[6 * i + j] = s;
    }
}
static void mat6_vec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += A[6 * i + k] * x[k];
    y[i] = s;
  }
}
/* v x m (motion cross motion) */
static void crm(const double* v, const double* m, double* out) {
  double a[3], b[3], c[3];
  cross3(v, m, a);
  cross3(v, m + 3, b);
  cross3(v + 3, m, c);
  for (int i = 0; i < 3; i++) { out[i] = a[i]; out[3 + i] = b[i] + c[i]; }
}
/* v x* f (motion cross force) */
static void crf(const double* v, const double* f, double* out) {
  double a[3], b[3], c[3];
  cross3(v, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f + 3, c);
  for (int i = 0; i < 3; i++) { out[i] = a[i] + b[i]; out[3 + i] = c[i]; }
}

static void spatial_inertia(double m, const double* c, const double* Ic, double* I6) {
  double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  memset(I6, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double cc = 0;
      for (int k = 0; k < 3; k++) cc += cx[3 * i + k] * cx[3 * k + j];
      I6[6 * i + j] = Ic[3 * i + j] - m * cc;
      I6[6 * i + 3 + j] = m * cx[3 * i + j];
      I6[6 * (3 + i) + j] = -m * cx[3 * i + j];
    }
  for (int i = 0; i < 3; i++) I6[6 * (3 + i) + 3 + i] = m;
}

/* ------------------------------------------------------------------ model build */
static void link_set_S(Link* l) {
  /* joint-frame twist -> child link frame via Tpost (joint frame in child frame) */
  double wj[3] = {0, 0, 0}, vj[3] = {0, 0, 0};
  memset(l->S, 0, sizeof l->S);
  if (l->jtype == DART_JT_REVOLUTE) memcpy(wj, l->axis, sizeof wj);
  else if (l->jtype == DART_JT_PRISMATIC) memcpy(vj, l->axis, sizeof vj);
  else return;
  double R[9], p[3] = {l->Tpost[3], l->Tpost[7], l->Tpost[11]};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = l->Tpost[4 * i + j];
  double w[3], v[3], t[3];
  for (int i = 0; i < 3; i++) {
    w[i] = R[3 * i] * wj[0] + R[3 * i + 1] * wj[1] + R[3 * i + 2] * wj[2];
    v[i] = R[3 * i] * vj[0] + R[3 * i + 1] * vj[1] + R[3 * i + 2] * vj[2];
  }
  cross3(p, w, t); /* v(body origin) = v(joint origin) + w x (O_B - O_J) = v_J + p x w */
  for (int i = 0; i < 3; i++) { l->S[i] = w[i]; l->S[3 + i] = v[i] + t[i]; }
}

static int add_link(OracleWorld* w, int parent, int jtype, int dof, const double* axis, const double* Tpre,
                    const double* Tpost) {
  Link* l = &w->L[w->nl];
  memset(l, 0, sizeof *l);
  l->parent = parent; l->jtype = jtype; l->dof = dof;
  if (axis) memcpy(l->axis, axis, 3 * sizeof(double));
  if (Tpre) memcpy(l->Tpre, Tpre, 16 * sizeof(double)); else mat4_identity(l->Tpre);
  if (Tpost) memcpy(l->Tpost, Tpost, 16 * sizeof(double)); else mat4_identity(l->Tpost);
  link_set_S(l);
  return w->nl++;
}

OracleWorld* oracle_create(const DartModelCard* card) {
  if (!card || card->version != DART_CARD_VERSION || card->struct_bytes != (int32_t)sizeof(DartModelCard)) return NULL;
  OracleWorld* w = (OracleWorld*)calloc(1, sizeof(OracleWorld));
  w->card = *card;
  w->n = card->ndofs;
  w->solver = 0; w->pgs_k1 = 30; w->pgs_k2 = 30; w->planar_drop_z = 1; w->ext_body = -1;
  static const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
  for (int b = 0; b < card->nbodies; b++) {
    int pl = card->parent[b] < 0 ? -1 : w->body_link[card->parent[b]];
    int d0 = card->dof_offset[b];
    const double* Tpj = card->T_pj[b];
    const double* Tcj = card->T_cj[b];
    const double* ax = card->axes[b];
    int last = -1;
    switch (card->jtype[b]) {
      case DART_JT_WELD: last = add_link(w, pl, DART_JT_WELD, -1, NULL, Tpj, Tcj); break;
      case DART_JT_PRISMATIC: last = add_link(w, pl, DART_JT_PRISMATIC, d0, ax, Tpj, Tcj); break;
      case DART_JT_REVOLUTE: last = add_link(w, pl, DART_JT_REVOLUTE, d0, ax, Tpj, Tcj); break;
      case DART_JT_TRANSLATIONAL: {
        int a = add_link(w, pl, DART_JT_PRISMATIC, d0, ex, Tpj, NULL);
        int c = add_link(w, a, DART_JT_PRISMATIC, d0 + 1, ey, NULL, NULL);
        last = add_link(w, c, DART_JT_PRISMATIC, d0 + 2, ez, NULL, Tcj);
      } break;
      case DART_JT_EULER_XYZ: { /* R = Rx(q0) Ry(q1) Rz(q2) */
        int a = add_link(w, pl, DART_JT_REVOLUTE, d0, ex, Tpj, NULL);
        int c = add_link(w, a, DART_JT_REVOLUTE, d0 + 1, ey, NULL, NULL);
        last = add_link(w, c, DART_JT_REVOLUTE, d0 + 2, ez, NULL, Tcj);
      } break;
      case DART_JT_EULER_ZYX: { /* R = Rz(q0) Ry(q1) Rx(q2) */
        int a = add_link(w, pl, DART_JT_REVOLUTE, d0, ez, Tpj, NULL);
        int c = add_link(w, a, DART_JT_REVOLUTE, d0 + 1, ey, NULL, NULL);
        last = add_link(w, c, DART_JT_REVOLUTE, d0 + 2, ex, NULL, Tcj);
      } break;
      case DART_JT_UNIVERSAL: {
        int a = add_link(w, pl, DART_JT_REVOLUTE, d0, ax, Tpj, NULL);
        last = add_link(w, a, DART_JT_REVOLUTE, d0 + 1, ax + 3, NULL, Tcj);
      } break;
      case DART_JT_FREE: { /* translation x y z (dofs d0+3..5), then rotations x y z (dofs d0..d0+2); see OracleWorld.free_root */
        if (b != 0 || card->parent[b] >= 0 || d0 != 0) { free(w); return NULL; }
        int a = add_link(w, pl, DART_JT_PRISMATIC, d0 + 3, ex, Tpj, NULL);
        int c2 = add_link(w, a, DART_JT_PRISMATIC, d0 + 4, ey, NULL, NULL);
        int c3 = add_link(w, c2, DART_JT_PRISMATIC, d0 + 5, ez, NULL, NULL);
        int r1 = add_link(w, c3, DART_JT_REVOLUTE, d0, ex, NULL, NULL);
        int r2 = add_link(w, r1, DART_JT_REVOLUTE, d0 + 1, ey, NULL, NULL);
        last = add_link(w, r2, DART_JT_REVOLUTE, d0 + 2, ez, NULL, Tcj);
        w->free_root = 1; w->free_rot_link = last; memcpy(w->free_Tpost, w->L[last].Tpost, sizeof w->free_Tpost);
      } break;
      default: free(w); return NULL;
    }
    w->body_link[b] = last;
    Link* l = &w->L[last];
    l->mass = card->mass[b];
    memcpy(l->com, card->com[b], sizeof l->com);
    spatial_inertia(card->mass[b], card->com[b], card->mass[b] > 0 ? card->inertia[b] : (const double[9]){0}, l->I);
  }
  for (int i = 0; i < w->n; i++) { w->q[i] = card->init_pos[i]; w->dq[i] = card->init_vel[i]; }
  return w;
}
void oracle_destroy(OracleWorld* w) { free(w); }
void oracle_set_solver(OracleWorld* w, int solver, int k1, int k2) { w->solver = solver; w->pgs_k1 = k1; w->pgs_k2 = k2; }
void oracle_set_assumption(OracleWorld* w, int id, int value) {
  if (id == 5) w->a5_surface_point = value;
  if (id == 7) w->a7_iterate_bounds = value;
}
void oracle_set_state(OracleWorld* w, const double* q, const double* dq) {
  memcpy(w->q, q, w->n * sizeof(double)); memcpy(w->dq, dq, w->n * sizeof(double));
}
void oracle_get_state(const OracleWorld* w, double* q, double* dq) {
  memcpy(q, w->q, w->n * sizeof(double)); memcpy(dq, w->dq, w->n * sizeof(double));
}
void oracle_set_forces(OracleWorld* w, const double* tau) { memcpy(w->tau, tau, w->n * sizeof(double)); }
void oracle_set_task_state(OracleWorld* w, const double* v4) { memcpy(w->task_state, v4, sizeof w->task_state); }
void oracle_get_task_state(const OracleWorld* w, double* v4) { memcpy(v4, w->task_state, sizeof w->task_state); }
void oracle_set_ext_force(OracleWorld* w, int body, const double* f3) {
  w->ext_body = (f3 && body >= 0 && body < w->card.nbodies) ? body : -1;
  if (w->ext_body >= 0) memcpy(w->ext_f, f3, sizeof w->ext_f);
}
/* pydart2 World.reset: time 0, positions/velocities back to initial, forces cleared (dart_world.py:20-22) */
void oracle_constraint_forces(const OracleWorld* w, double* out) { memcpy(out, w->cf_last, w->n * sizeof(double)); }
void oracle_reset(OracleWorld* w) {
  memset(w->cf_last, 0, sizeof w->cf_last);
  for (int i = 0; i < w->n; i++) { w->q[i] = w->card.init_pos[i]; w->dq[i] = w->card.init_vel[i]; w->tau[i] = 0; }
  w->time = 0;
}

/* ------------------------------------------------------------------ kinematics */

/* ------------------------------------------------------------------ free-joint coordinate maps */
static void so3_exp(const double* r, double* R) {
  double th = sqrt(dot3(r, r));
  double a = th < 1e-8 ? 1.0 - th * th / 6.0 : sin(th) / th, b = th < 1e-8 ? 0.5 - th * th / 24.0 : (1.0 - cos(th)) / (th * th);
  double K[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double k2 = 0;
      for (int k = 0; k < 3; k++) k2 += K[3 * i + k] * K[3 * k + j];
      R[3 * i + j] = (i == j ? 1.0 : 0.0) + a * K[3 * i + j] + b * k2;
    }
}
/* log map through the unit quaternion (Shepperd's largest-pivot extraction), angle = 2 atan2(|v|, w) in [0, pi] */
static void so3_log(const double* R, double* r) {
  double tr = R[0] + R[4] + R[8], q[4]; /* w x y z */
  if (tr > 0) {
    double s = 2.0 * sqrt(tr + 1.0);
    q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
    q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
    q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
  } else {
    double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
    q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
  }
  if (q[0] < 0) for (int a = 0; a < 4; a++) q[a] = -q[a];
  double nv = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double k = nv < 1e-12 ? 2.0 / q[0] : 2.0 * atan2(nv, q[0]) / nv;
  for (int a = 0; a < 3; a++) r[a] = k * q[1 + a];
}
/* public (q, dq) -> internal (qi, dqi); identity unless the root is a FreeJoint.  The rotation chart is re-centred on the
 * current orientation R0 = exp(q[0:3]): joint rotation = Rx(a) Ry(b) Rz(c) R0 with a = b = c = 0, so the rate matrix E is the
 * identity (rates = angular velocity in the joint's parent frame) and there is no chart singularity however far the body turns. */
static void sync_internal(OracleWorld* w) {
  memcpy(w->qi, w->q, sizeof w->qi); memcpy(w->dqi, w->dq, sizeof w->dqi);
  if (!w->free_root) return;
  double* R = w->R0;
  so3_exp(w->q, R);
  for (int a = 0; a < 3; a++) {                                /* w_parent = R w_body, pdot = R v_body */
    w->qi[a] = 0.0;                                            /* qi[3:6] = translation (same slots) */
    w->dqi[a] = R[3 * a] * w->dq[0] + R[3 * a + 1] * w->dq[1] + R[3 * a + 2] * w->dq[2];
    w->dqi[3 + a] = R[3 * a] * w->dq[3] + R[3 * a + 1] * w->dq[4] + R[3 * a + 2] * w->dq[5];
  }
  /* joint transform Rz(c) R0: R0 is folded into the last root link's joint-to-child transform, Tpost_eff = Tpost R0^T */
  Link* l = &w->L[w->free_rot_link];
  double Rt[16];
  mat4_identity(Rt);
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[4 * a + b] = R[3 * b + a];
  mat4_mul(w->free_Tpost, Rt, l->Tpost);
  link_set_S(l);
}
/* after a world step: new internal velocities vs (at the old pose) -> new public twist, DART's position update */
static void free_root_advance(OracleWorld* w, const double* vs, double dt) {
  double wb[3], vb[3], dR[9], Rn[9], step[3];
  const double* R = w->R0;
  for (int a = 0; a < 3; a++) {
    wb[a] = R[a] * vs[0] + R[3 + a] * vs[1] + R[6 + a] * vs[2];          /* R^T w */
    vb[a] = R[a] * vs[3] + R[3 + a] * vs[4] + R[6 + a] * vs[5];          /* R^T pdot */
  }
  for (int a = 0; a < 3; a++) step[a] = wb[a] * dt;
  so3_exp(step, dR);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rn[3 * i + j] = 0; for (int k = 0; k < 3; k++) Rn[3 * i + j] += R[3 * i + k] * dR[3 * k + j]; }
  for (int a = 0; a < 3; a++) w->q[3 + a] += dt * vs[3 + a];             /* p += R v_b dt = pdot dt */
  so3_log(Rn, w->q);
  for (int a = 0; a < 3; a++) { w->dq[a] = wb[a]; w->dq[3 + a] = vb[a]; }
}

static void kinematics(OracleWorld* w) {
  sync_internal(w);
  for (int i = 0; i < w->nl; i++) {
    Link* l = &w->L[i];
    double TJ[16];
    mat4_identity(TJ);
    if (l->jtype == DART_JT_REVOLUTE) {
      double R[9];
      rot_axis(l->axis, w->qi[l->dof], R);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) TJ[4 * a + b] = R[3 * a + b];
    } else if (l->jtype == DART_JT_PRISMATIC) {
      for (int a = 0; a < 3; a++) TJ[4 * a + 3] = l->axis[a] * w->qi[l->dof];
    }
    double Tinv[16], T[16];
    mat4_inv_rigid(l->Tpost, Tinv);
    mat4_mul(l->Tpre, TJ, T);
    mat4_mul(T, Tinv, T); /* child link frame in parent link frame */
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) w->X[i].E[3 * a + b] = T[4 * b + a];
      w->X[i].r[a] = T[4 * a + 3];
    }
    if (l->parent < 0) memcpy(w->W[i], T, sizeof T);
    else mat4_mul(w->W[l->parent], T, w->W[i]);
  }
}

/* RNEA: tau = ID(q, dq, ddq) with gravity; ddq may be NULL (=0) */
static void rnea(OracleWorld* w, const double* dq, const double* ddq, int with_gravity, double* tau) {
  double v[MAXL][6], a[MAXL][6], f[MAXL][6];
  double a0[6] = {0, 0, 0, 0, 0, 0};
  if (with_gravity) for (int i = 0; i < 3; i++) a0[3 + i] = -w->card.gravity[i];
  for (int i = 0; i < w->nl; i++) {
    Link* l = &w->L[i];
    double vp[6] = {0}, ap[6];
    if (l->parent < 0) { memcpy(ap, a0, sizeof ap); }
    else { memcpy(vp, v[l->parent], sizeof vp); memcpy(ap, a[l->parent], sizeof ap); }
    double vJ[6] = {0};
    xf_motion(&w->X[i], vp, v[i]);
    xf_motion(&w->X[i], ap, a[i]);
    if (l->dof >= 0) {
      double qd = dq ? dq[l->dof] : 0.0, qdd = ddq ? ddq[l->dof] : 0.0;
      for (int k = 0; k < 6; k++) vJ[k] = l->S[k] * qd;
      double c[6];
      for (int k = 0; k < 6; k++) v[i][k] += vJ[k];
      crm(v[i], vJ, c);
      for (int k = 0; k < 6; k++) a[i][k] += l->S[k] * qdd + c[k];
    }
    double Iv[6], Ia[6], t[6];
    mat6_vec(l->I, v[i], Iv);
    mat6_vec(l->I, a[i], Ia);
    crf(v[i], Iv, t);
    for (int k = 0; k < 6; k++) f[i][k] = Ia[k] + t[k];
  }
  for (int i = w->nl - 1; i >= 0; i--) {
    Link* l = &w->L[i];
    if (l->dof >= 0) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += l->S[k] * f[i][k];
      tau[l->dof] = s;
    }
    if (l->parent >= 0) {
      double fp[6];
      xf_force_T(&w->X[i], f[i], fp);
      for (int k = 0; k < 6; k++) f[l->parent][k] += fp[k];
    }
  }
}

/* CRBA -> dense M (n x n) */
static void crba(OracleWorld* w, double* M) {
  static __thread double Ic[MAXL][36];
  int n = w->n;
  memset(M, 0, n * n * sizeof(double));
  for (int i = 0; i < w->nl; i++) memcpy(Ic[i], w->L[i].I, 36 * sizeof(double));
  for (int i = w->nl - 1; i >= 0; i--) {
    Link* l = &w->L[i];
    if (l->parent >= 0) {
      double X6[36], T[36];
      xf_matrix(&w->X[i], X6);
      mat6_XtIX(X6, Ic[i], T);
      for (int k = 0; k < 36; k++) Ic[l->parent][k] += T[k];
    }
  }
  for (int i = 0; i < w->nl; i++) {
    Link* l = &w->L[i];
    if (l->dof < 0) continue;
    double F[6];
    mat6_vec(Ic[i], l->S, F);
    double s = 0;
    for (int k = 0; k < 6; k++) s += l->S[k] * F[k];
    M[l->dof * n + l->dof] = s;
    int j = i;
    while (w->L[j].parent >= 0) {
      double Fp[6];
      xf_force_T(&w->X[j], F, Fp);
      memcpy(F, Fp, sizeof F);
      j = w->L[j].parent;
      if (w->L[j].dof >= 0) {
        double t = 0;
        for (int k = 0; k < 6; k++) t += w->L[j].S[k] * F[k];
        M[l->dof * n + w->L[j].dof] = t;
        M[w->L[j].dof * n + l->dof] = t;
      }
    }
  }
}

/* Jacobian row: d . velocity of world point P rigidly attached to link li */
static void point_jacobian(OracleWorld* w, int li, const double* P, const double* d, double* J) {
  for (int k = 0; k < w->n; k++) J[k] = 0;
  for (int j = li; j >= 0; j = w->L[j].parent) {
    Link* l = &w->L[j];
    if (l->dof < 0) continue;
    const double* Wm = w->W[j];
    double ww[3], vw[3], rel[3], t[3];
    for (int a = 0; a < 3; a++) {
      ww[a] = Wm[4 * a] * l->S[0] + Wm[4 * a + 1] * l->S[1] + Wm[4 * a + 2] * l->S[2];
      vw[a] = Wm[4 * a] * l->S[3] + Wm[4 * a + 1] * l->S[4] + Wm[4 * a + 2] * l->S[5];
      rel[a] = P[a] - Wm[4 * a + 3];
    }
    cross3(ww, rel, t);
    J[l->dof] = d[0] * (vw[0] + t[0]) + d[1] * (vw[1] + t[1]) + d[2] * (vw[2] + t[2]);
  }
}


/* ------------------------------------------------------------------ box-box contacts (link-link self-collision)
 * Restatement of ODE's dBoxBox (ode/src/box.cpp, the routine DART's ODE detector calls for two boxes; third-party,
 * unpinned): 15-axis separating-axis test with the 1.05 fudge factor that prefers face axes, then either the single
 * edge-edge point (closest points of the two edges, midpoint) or the face case -- the incident face of the other box
 * is clipped against the reference face's rectangle (intersectRectQuad) and the clipped points that lie below the
 * reference face are the contacts.  R1, R2: row-major 3x3 whose COLUMNS are the box axes in world coordinates;
 * h1, h2: HALF extents.  Output normal points from box 1 towards box 2.  Returns the number of points (<= 8).
 *
 * Attribution: rect_quad and box_box follow ODE's `intersectRectQuad` and `dBoxBox` (ode/src/box.cpp) step for step -- for an
 * oracle, following the published routine of the named third-party dependency is the point.  Open Dynamics Engine, Copyright (C)
 * 2001-2003 Russell L. Smith, all rights reserved; dual-licensed (GNU LGPL 2.1+ / BSD-style), used here under the BSD-style
 * license: redistribution and use in source and binary forms, with or without modification, are permitted provided that source
 * redistributions retain this copyright notice, this list of conditions and the disclaimer; binary redistributions reproduce them
 * in the documentation; and neither the copyright owner's nor the contributors' names are used to endorse or promote derived
 * products without specific prior written permission.  THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS"
 * AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A
 * PARTICULAR PURPOSE ARE DISCLAIMED; IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY DAMAGES ARISING IN ANY
 * WAY OUT OF THE USE OF THIS SOFTWARE. */
typedef struct { double pos[3], depth; } BoxContact;

static int rect_quad(const double h[2], const double p[8], double ret[16]) {
  int nq = 4, nr = 0;
  double buffer[16];
  const double* q = p;
  double* r = ret;
  for (int dir = 0; dir <= 1; dir++) {
    for (int sign = -1; sign <= 1; sign += 2) {
      const double* pq = q;
      double* pr = r;
      nr = 0;
      for (int i = nq; i > 0; i--) {
        if (sign * pq[dir] < h[dir]) {
          pr[0] = pq[0]; pr[1] = pq[1]; pr += 2; nr++;
          if (nr & 8) { q = r; goto done; }
        }
        const double* nextq = (i > 1) ? pq + 2 : q;
        if ((sign * pq[dir] < h[dir]) ^ (sign * nextq[dir] < h[dir])) {
          pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (sign * h[dir] - pq[dir]);
          pr[dir] = sign * h[dir];
          pr += 2; nr++;
          if (nr & 8) { q = r; goto done; }
        }
        pq += 2;
      }
      q = r;
      r = (q == ret) ? buffer : ret;
      nq = nr;
    }
  }
done:
  if (q != ret) memcpy(ret, q, nr * 2 * sizeof(double));
  return nr;
}

static void col3(const double* R, int j, double* out) { out[0] = R[j]; out[1] = R[3 + j]; out[2] = R[6 + j]; }

static int box_box(const double* p1, const double* R1, const double* A, const double* p2, const double* R2, const double* B,
                   double* normal, BoxContact* out) {
  const double fudge_factor = 1.05, fudge2 = 1.0e-5, eps = 2.220446049250313e-16;
  double p[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, pp[3], R[3][3], Q[3][3];
  double u[3][3], v[3][3];
  for (int j = 0; j < 3; j++) { col3(R1, j, u[j]); col3(R2, j, v[j]); }
  for (int i = 0; i < 3; i++) { pp[i] = dot3(u[i], p); for (int j = 0; j < 3; j++) { R[i][j] = dot3(u[i], v[j]); Q[i][j] = fabs(R[i][j]); } }
  double s = -INFINITY, normalC[3] = {0, 0, 0};
  const double* normalR = NULL;
  int invert_normal = 0, code = 0;
#define TST1(expr1, expr2, nrm, cc)                                                              \
  { double e1 = (expr1), s2 = fabs(e1) - (expr2); if (s2 > 0) return 0;                          \
    if (s2 > s) { s = s2; normalR = (nrm); invert_normal = (e1 < 0); code = (cc); } }
  TST1(pp[0], A[0] + B[0] * Q[0][0] + B[1] * Q[0][1] + B[2] * Q[0][2], u[0], 1);
  TST1(pp[1], A[1] + B[0] * Q[1][0] + B[1] * Q[1][1] + B[2] * Q[1][2], u[1], 2);
  TST1(pp[2], A[2] + B[0] * Q[2][0] + B[1] * Q[2][1] + B[2] * Q[2][2], u[2], 3);
  TST1(dot3(v[0], p), A[0] * Q[0][0] + A[1] * Q[1][0] + A[2] * Q[2][0] + B[0], v[0], 4);
  TST1(dot3(v[1], p), A[0] * Q[0][1] + A[1] * Q[1][1] + A[2] * Q[2][1] + B[1], v[1], 5);
  TST1(dot3(v[2], p), A[0] * Q[0][2] + A[1] * Q[1][2] + A[2] * Q[2][2] + B[2], v[2], 6);
#undef TST1
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] += fudge2;
#define TST2(expr1, expr2, n1, n2, n3, cc)                                                       \
  { double e1 = (expr1), s2 = fabs(e1) - (expr2); if (s2 > eps) return 0;                        \
    double l = sqrt((n1) * (n1) + (n2) * (n2) + (n3) * (n3));                                    \
    if (l > eps) { s2 /= l; if (s2 * fudge_factor > s) { s = s2; normalR = NULL;                 \
      normalC[0] = (n1) / l; normalC[1] = (n2) / l; normalC[2] = (n3) / l; invert_normal = (e1 < 0); code = (cc); } } }
  TST2(pp[2] * R[1][0] - pp[1] * R[2][0], A[1] * Q[2][0] + A[2] * Q[1][0] + B[1] * Q[0][2] + B[2] * Q[0][1], 0, -R[2][0], R[1][0], 7);
  TST2(pp[2] * R[1][1] - pp[1] * R[2][1], A[1] * Q[2][1] + A[2] * Q[1][1] + B[0] * Q[0][2] + B[2] * Q[0][0], 0, -R[2][1], R[1][1], 8);
  TST2(pp[2] * R[1][2] - pp[1] * R[2][2], A[1] * Q[2][2] + A[2] * Q[1][2] + B[0] * Q[0][1] + B[1] * Q[0][0], 0, -R[2][2], R[1][2], 9);
  TST2(pp[0] * R[2][0] - pp[2] * R[0][0], A[0] * Q[2][0] + A[2] * Q[0][0] + B[1] * Q[1][2] + B[2] * Q[1][1], R[2][0], 0, -R[0][0], 10);
  TST2(pp[0] * R[2][1] - pp[2] * R[0][1], A[0] * Q[2][1] + A[2] * Q[0][1] + B[0] * Q[1][2] + B[2] * Q[1][0], R[2][1], 0, -R[0][1], 11);
  TST2(pp[0] * R[2][2] - pp[2] * R[0][2], A[0] * Q[2][2] + A[2] * Q[0][2] + B[0] * Q[1][1] + B[1] * Q[1][0], R[2][2], 0, -R[0][2], 12);
  TST2(pp[1] * R[0][0] - pp[0] * R[1][0], A[0] * Q[1][0] + A[1] * Q[0][0] + B[1] * Q[2][2] + B[2] * Q[2][1], -R[1][0], R[0][0], 0, 13);
  TST2(pp[1] * R[0][1] - pp[0] * R[1][1], A[0] * Q[1][1] + A[1] * Q[0][1] + B[0] * Q[2][2] + B[2] * Q[2][0], -R[1][1], R[0][1], 0, 14);
  TST2(pp[1] * R[0][2] - pp[0] * R[1][2], A[0] * Q[1][2] + A[1] * Q[0][2] + B[0] * Q[2][1] + B[1] * Q[2][0], -R[1][2], R[0][2], 0, 15);
#undef TST2
  if (!code) return 0;
  if (normalR) { normal[0] = normalR[0]; normal[1] = normalR[1]; normal[2] = normalR[2]; }
  else for (int a = 0; a < 3; a++) normal[a] = u[0][a] * normalC[0] + u[1][a] * normalC[1] + u[2][a] * normalC[2];
  if (invert_normal) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
  const double depth = -s;
  if (code > 6) { /* edge-edge: the closest points of the two edges */
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int j = 0; j < 3; j++) {
      double sg = dot3(normal, u[j]) > 0 ? 1.0 : -1.0;
      for (int a = 0; a < 3; a++) pa[a] += sg * A[j] * u[j][a];
      sg = dot3(normal, v[j]) > 0 ? -1.0 : 1.0;
      for (int a = 0; a < 3; a++) pb[a] += sg * B[j] * v[j][a];
    }
    const double* ua = u[(code - 7) / 3];
    const double* ub = v[(code - 7) % 3];
    double d3[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double uaub = dot3(ua, ub), q1 = dot3(ua, d3), q2 = -dot3(ub, d3), d = 1 - uaub * uaub, alpha = 0, beta = 0;
    if (d > 1e-4) { d = 1 / d; alpha = (q1 + uaub * q2) * d; beta = (uaub * q1 + q2) * d; }
    for (int a = 0; a < 3; a++) out[0].pos[a] = 0.5 * ((pa[a] + ua[a] * alpha) + (pb[a] + ub[a] * beta));
    out[0].depth = depth;
    return 1;
  }
  /* face case: reference face on box a, incident face on box b */
  const double (*Ra)[3] = code <= 3 ? u : v;
  const double (*Rb)[3] = code <= 3 ? v : u;
  const double *pa = code <= 3 ? p1 : p2, *pb = code <= 3 ? p2 : p1, *Sa = code <= 3 ? A : B, *Sb = code <= 3 ? B : A;
  double normal2[3], nr[3], anr[3];
  for (int a = 0; a < 3; a++) normal2[a] = code <= 3 ? normal[a] : -normal[a];
  for (int j = 0; j < 3; j++) { nr[j] = dot3(Rb[j], normal2); anr[j] = fabs(nr[j]); }
  int lanr, a1, a2;
  if (anr[1] > anr[0]) { if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  else { if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  double center[3];
  for (int a = 0; a < 3; a++) center[a] = pb[a] - pa[a] + (nr[lanr] < 0 ? 1.0 : -1.0) * Sb[lanr] * Rb[lanr][a];
  int codeN = code <= 3 ? code - 1 : code - 4, code1, code2;
  if (codeN == 0) { code1 = 1; code2 = 2; } else if (codeN == 1) { code1 = 0; code2 = 2; } else { code1 = 0; code2 = 1; }
  double quad[8], c1 = dot3(center, Ra[code1]), c2 = dot3(center, Ra[code2]);
  double m11 = dot3(Ra[code1], Rb[a1]), m12 = dot3(Ra[code1], Rb[a2]), m21 = dot3(Ra[code2], Rb[a1]), m22 = dot3(Ra[code2], Rb[a2]);
  {
    double k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
    quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4; quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
    quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4; quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
  }
  double rect[2] = {Sa[code1], Sa[code2]}, ret[16];
  int nq = rect_quad(rect, quad, ret);
  if (nq < 1) return 0;
  double det1 = 1.0 / (m11 * m22 - m12 * m21);
  m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
  int cnum = 0;
  for (int j = 0; j < nq; j++) {
    double k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2), k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
    double pt[3];
    for (int a = 0; a < 3; a++) pt[a] = center[a] + k1 * Rb[a1][a] + k2 * Rb[a2][a];
    double dep = Sa[codeN] - dot3(normal2, pt);
    if (dep >= 0) {
      for (int a = 0; a < 3; a++) out[cnum].pos[a] = pt[a] + pa[a] - (code < 4 ? 0.0 : normal[a] * dep);
      out[cnum].depth = dep;
      cnum++;
    }
  }
  return cnum;
}
int oracle_box_box(const double* p1, const double* R1, const double* h1, const double* p2, const double* R2, const double* h2,
                   double* normal, double* pos_depth /* [8][4] */) {
  BoxContact c[8];
  int k = box_box(p1, R1, h1, p2, R2, h2, normal, c);
  for (int i = 0; i < k; i++) { pos_depth[4 * i] = c[i].pos[0]; pos_depth[4 * i + 1] = c[i].pos[1]; pos_depth[4 * i + 2] = c[i].pos[2]; pos_depth[4 * i + 3] = c[i].depth; }
  return k;
}

/* ------------------------------------------------------------------ dense SPD solve */
static int cholesky(double* A, int n) { /* in place lower, row-major */
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s <= 0) return -1;
    A[j * n + j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / A[j * n + j];
    }
  }
  return 0;
}
static void chol_solve(const double* Lm, int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= Lm[i * n + k] * x[k];
    x[i] = s / Lm[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= Lm[k * n + i] * x[k];
    x[i] = s / Lm[i * n + i];
  }
}

/* ------------------------------------------------------------------ boxed LCP solvers
 * find x, w = A x - b with  lo<=x<=hi,  (x==lo -> w>=0), (x==hi -> w<=0), (lo<x<hi -> w==0)
 * rows listed in idx[0..m) of the full system of stride `ld`. */
long oracle_bpp_hist[64];
static int blcp_exact(const double* A, const double* b, const double* lo, const double* hi, const int* idx, int m,
                      int ld, double* x) {
  /* block principal pivoting with single-pivot (least index) fallback; sets: 0 free, 1 at lo, 2 at hi */
  int set[MAXM];
  for (int i = 0; i < m; i++) {
    int g = idx[i];
    set[i] = (lo[g] == 0.0 && hi[g] > 0) ? 1 : (hi[g] == 0.0 && lo[g] < 0 ? 2 : (lo[g] == hi[g] ? 1 : 0));
    if (set[i] == 0 && x[g] <= lo[g]) set[i] = 1;
    if (set[i] == 0 && x[g] >= hi[g]) set[i] = 2;
  }
  int best = m + 1, patience = 10, single = 0;
  for (int iter = 0; iter < 4000; iter++) {
    int fi[MAXM], nf = 0;
    for (int i = 0; i < m; i++) if (set[i] == 0) fi[nf++] = i;
    static __thread double AF[MAXM * MAXM];
    double rhs[MAXM];
    for (int a = 0; a < nf; a++) {
      int ga = idx[fi[a]];
      double s = b[ga];
      for (int j = 0; j < m; j++) if (set[j] != 0) s -= A[ga * ld + idx[j]] * (set[j] == 1 ? lo[idx[j]] : hi[idx[j]]);
      rhs[a] = s;
      for (int c = 0; c < nf; c++) AF[a * nf + c] = A[ga * ld + idx[fi[c]]];
    }
    if (nf > 0) {
      if (cholesky(AF, nf) != 0) return -1;
      chol_solve(AF, nf, rhs);
    }
    for (int i = 0; i < m; i++) x[idx[i]] = set[i] == 1 ? lo[idx[i]] : (set[i] == 2 ? hi[idx[i]] : 0.0);
    for (int a = 0; a < nf; a++) x[idx[fi[a]]] = rhs[a];
    /* infeasibilities */
    int bad[MAXM], nb = 0;
    for (int i = 0; i < m; i++) {
      int g = idx[i];
      if (set[i] == 0) {
        if (x[g] < lo[g] - 1e-14 * (1 + fabs(lo[g])) || x[g] > hi[g] + 1e-14 * (1 + fabs(hi[g]))) bad[nb++] = i;
      } else {
        double wv = -b[g];
        for (int j = 0; j < m; j++) wv += A[g * ld + idx[j]] * x[idx[j]];
        if (lo[g] == hi[g]) continue; /* pinned */
        if ((set[i] == 1 && wv < -1e-13) || (set[i] == 2 && wv > 1e-13)) bad[nb++] = i;
      }
    }
    if (nb == 0) { oracle_bpp_hist[iter < 63 ? iter : 63]++; return 0; }
    if (nb < best) { best = nb; patience = 10; single = 0; }
    else if (--patience <= 0) single = 1;
    int from = single ? nb - 1 : 0; /* single: switch only the largest index (Murty) */
    for (int k = from; k < nb; k++) {
      int i = bad[k], g = idx[i];
      if (set[i] == 0) set[i] = x[g] < lo[g] ? 1 : 2;
      else set[i] = 0;
    }
  }
  return -2;
}

static void pgs_sweeps(const double* A, const double* b, const double* lo, const double* hi, const int* idx, int m,
                       int ld, int iters, double* x) {
  for (int it = 0; it < iters; it++)
    for (int i = 0; i < m; i++) {
      int g = idx[i];
      double r = b[g];
      for (int j = 0; j < m; j++) r -= A[g * ld + idx[j]] * x[idx[j]];
      double xn = x[g] + r / A[g * ld + g];
      x[g] = xn < lo[g] ? lo[g] : (xn > hi[g] ? hi[g] : xn);
    }
}

/* ------------------------------------------------------------------ World::step */
int oracle_step(OracleWorld* w) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double dt = c->dt;
  /* a world that has left the representable regime (a coordinate or velocity beyond 1e6, or not finite) is frozen: it cannot come
   * back under the tasks' validity bound |s| < 100 within an env-step, the env reports done either way (hopper.py:60-62), and the
   * tree kernel skips such worlds because their LCPs would stall a whole launch (csrc/spatial_world_step.hpp) */
  for (int i = 0; i < n; i++)
    if (!(fabs(w->q[i]) < 1e6 && fabs(w->dq[i]) < 1e6)) { w->time += dt; return 0; }
  kinematics(w);
  crba(w, w->M);
  rnea(w, w->dqi, NULL, 1, w->C);
  /* H = M + dt D + dt^2 K ; rhs = tau - C - D dq - K (q + dt dq - rest) */
  static __thread double H[MAXN * MAXN];
  double rhs[MAXN];
  memcpy(H, w->M, n * n * sizeof(double));
  for (int i = 0; i < n; i++) {
    H[i * n + i] += dt * c->damping[i] + dt * dt * c->stiffness[i];
    rhs[i] = w->tau[i] - w->C[i] - c->damping[i] * w->dqi[i] -
             c->stiffness[i] * (w->qi[i] + dt * w->dqi[i] - c->rest[i]);
  }
  if (w->ext_all) {
    static const double E3b[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int bdy = 0; bdy < c->nbodies; bdy++) {
      int li = w->body_link[bdy];
      double P[3] = {w->W[li][3], w->W[li][7], w->W[li][11]}, Jd[MAXN];
      for (int a = 0; a < 3; a++) {
        if (w->ext_fb[bdy][a] == 0.0) continue;
        point_jacobian(w, li, P, E3b[a], Jd);
        for (int i = 0; i < n; i++) rhs[i] += Jd[i] * w->ext_fb[bdy][a];
      }
    }
  }
  if (w->ext_body >= 0) { /* generalized force of the external body force: J(origin)^T f */
    int li = w->body_link[w->ext_body];
    double P[3] = {w->W[li][3], w->W[li][7], w->W[li][11]}, Jd[MAXN];
    static const double E3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int a = 0; a < 3; a++) {
      point_jacobian(w, li, P, E3[a], Jd);
      for (int i = 0; i < n; i++) rhs[i] += Jd[i] * w->ext_f[a];
    }
  }
  if (cholesky(H, n) != 0) return -1;
  chol_solve(H, n, rhs);
  /* A3 (card.impulse_inertia): DART's impulse pass -- the unit-impulse tests that build A and the final velocity change --
   * runs on the NON-implicit articulated inertia (BodyNode::updateBiasImpulse / updateVelocityChangeFD ->
   * GenericJoint::updateVelocityChangeDynamic: getInvProjArtInertia()), i.e. on M, while the forward dynamics above used
   * the implicit one (updateAccelerationDynamic: getInvProjArtInertiaImplicit()), i.e. H.  HI = factor of the impulse inertia. */
  static __thread double HI[MAXN * MAXN];
  if (c->impulse_inertia == DART_IMPULSE_MASS) {
    memcpy(HI, w->M, n * n * sizeof(double));
    if (cholesky(HI, n) != 0) return -1;
  } else {
    memcpy(HI, H, n * n * sizeof(double));
  }
  double vs[MAXN];
  for (int i = 0; i < n; i++) vs[i] = w->dqi[i] + dt * rhs[i];
  if (w->free_root) {
    /* DART integrates the FreeJoint's BODY-FRAME twist: twist' = twist + dt * twist_acc.  With dq_int = T(q) twist the
     * acceleration maps as qdd_int = T twist_acc + Tdot twist, so the internal unconstrained velocity that corresponds to
     * DART's is v* - dt Tdot twist:  rotation  E rates = R w_b  ->  -Tdot twist = E^-1 Edot rates;
     *                                translation  pdot = R v_b  ->  -Tdot twist = -(w x pdot). */
    /* at the chart centre E = I and Edot rates = (rb rc, -ra rc, ra rb) */
    double x[3], wxp[3];
    double ra = w->dqi[0], rb = w->dqi[1], rc = w->dqi[2];
    x[0] = rb * rc; x[1] = -ra * rc; x[2] = ra * rb;
    cross3(w->dqi, w->dqi + 3, wxp);
    for (int a = 0; a < 3; a++) { vs[a] += dt * x[a]; vs[3 + a] -= dt * wxp[a]; }
  }

  /* ---- constraints at q_t ---- */
  static __thread double J[MAXM][MAXN], Y[MAXM][MAXN], A[MAXM * MAXM];
  double b[MAXM], lo[MAXM], hi[MAXM], x[MAXM];
  int findex[MAXM];
  int m = 0;
  w->ncontacts_last = 0;
  /* ---- contact points against the ground plane y = ground_y (normal +y) ---- */
  int ncp = 0;
  int cp_shape[MAXC], cp_shape_b[MAXC];   /* cp_shape_b: -1 = the ground, else the second shape of a link-link contact */
  double cp_P[MAXC][3], cp_depth[MAXC], cp_n[MAXC][3];
  for (int s = 0; s < c->nshapes && ncp < MAXC - 4; s++) {
    if (!c->shape_collidable[s] || !isfinite(c->ground_y)) continue;
    int li = w->body_link[c->shape_body[s]];
    double Ts[16];
    mat4_mul(w->W[li], c->shape_pose[s], Ts);
    if (c->shape_type[s] == DART_SH_CAPSULE) {
      double r = c->shape_size[s][0], hl = 0.5 * c->shape_size[s][1];
      double p1[3], p2[3];
      for (int a = 0; a < 3; a++) { p1[a] = Ts[4 * a + 3] + hl * Ts[4 * a + 2]; p2[a] = Ts[4 * a + 3] - hl * Ts[4 * a + 2]; }
      const double* pe = (p2[1] < p1[1]) ? p2 : p1; /* lowest endpoint; exact tie -> +axis end (ODE t=0) */
      double d = pe[1] - c->ground_y;
      if (d > r) continue;
      /* ODE dCollideSpheres(pl, r, pb, 0): pos = pl - n (r + d)/2 */
      cp_P[ncp][0] = pe[0]; cp_P[ncp][1] = w->a5_surface_point ? pe[1] - r : pe[1] - 0.5 * (r + d); cp_P[ncp][2] = pe[2];
      cp_depth[ncp] = r - d; cp_shape[ncp] = s; cp_shape_b[ncp] = -1; cp_n[ncp][0] = 0; cp_n[ncp][1] = 1; cp_n[ncp][2] = 0; ncp++;
    } else if (c->shape_type[s] == DART_SH_BOX) {
      /* box vs. the (huge) ground box, ODE/DART dBoxBox face case with the ground's +y face as reference: the
       * incident face is the box face most anti-parallel to the normal; its vertices that lie below the ground
       * surface are the contact points (position = the vertex, depth = distance below the surface). */
      int k = 0;
      double best = -1;
      for (int a = 0; a < 3; a++) if (fabs(Ts[4 * 1 + a]) > best) { best = fabs(Ts[4 * 1 + a]); k = a; }
      double sgn = Ts[4 * 1 + k] > 0 ? -1.0 : 1.0; /* face whose outward normal points down */
      int a1 = (k + 1) % 3, a2 = (k + 2) % 3;
      double hk = 0.5 * c->shape_size[s][k], h1 = 0.5 * c->shape_size[s][a1], h2 = 0.5 * c->shape_size[s][a2];
      static const double sg[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
      for (int v = 0; v < 4; v++) {
        double P[3];
        for (int a = 0; a < 3; a++)
          P[a] = Ts[4 * a + 3] + sgn * hk * Ts[4 * a + k] + sg[v][0] * h1 * Ts[4 * a + a1] + sg[v][1] * h2 * Ts[4 * a + a2];
        double depth = c->ground_y - P[1];
        if (depth < 0) continue;
        memcpy(cp_P[ncp], P, sizeof P);
        cp_depth[ncp] = depth; cp_shape[ncp] = s; cp_shape_b[ncp] = -1; cp_n[ncp][0] = 0; cp_n[ncp][1] = 1; cp_n[ncp][2] = 0; ncp++;
      }
    }
  }
  /* ---- link-link contacts (set_self_collision_check(True), walker3d.py:26): every pair of box shapes whose bodies are
   * not parent and child (DART skips adjacent bodies by default).  The lower-indexed shape is ODE's o1; the contact
   * normal handed to the constraint points into o1 (dCollideBoxBox negates dBoxBox's). */
  if (c->self_collision) {
    for (int sa = 0; sa < c->nshapes; sa++)
      for (int sb = sa + 1; sb < c->nshapes; sb++) {
        int ba = c->shape_body[sa], bb = c->shape_body[sb];
        if (c->shape_type[sa] != DART_SH_BOX || c->shape_type[sb] != DART_SH_BOX) continue;
        if (!c->shape_collidable[sa] || !c->shape_collidable[sb]) continue;
        if (ba == bb || c->parent[ba] == bb || c->parent[bb] == ba) continue;
        double Ta[16], Tb[16], Ra[9], Rb[9], pa[3], pb[3], ha[3], hb[3], nrm[3];
        mat4_mul(w->W[w->body_link[ba]], c->shape_pose[sa], Ta);
        mat4_mul(w->W[w->body_link[bb]], c->shape_pose[sb], Tb);
        for (int i = 0; i < 3; i++) {
          pa[i] = Ta[4 * i + 3]; pb[i] = Tb[4 * i + 3]; ha[i] = 0.5 * c->shape_size[sa][i]; hb[i] = 0.5 * c->shape_size[sb][i];
          for (int j = 0; j < 3; j++) { Ra[3 * i + j] = Ta[4 * i + j]; Rb[3 * i + j] = Tb[4 * i + j]; }
        }
        BoxContact bc[8];
        int k = box_box(pa, Ra, ha, pb, Rb, hb, nrm, bc);
        for (int i = 0; i < k && ncp < MAXC; i++) {
          memcpy(cp_P[ncp], bc[i].pos, sizeof bc[i].pos);
          cp_depth[ncp] = bc[i].depth; cp_shape[ncp] = sa; cp_shape_b[ncp] = sb;
          cp_n[ncp][0] = -nrm[0]; cp_n[ncp][1] = -nrm[1]; cp_n[ncp][2] = -nrm[2];
          ncp++;
        }
      }
  }
  for (int ci = 0; ci < ncp; ci++) {
    int s = cp_shape[ci];
    int li = w->body_link[c->shape_body[s]];
    int lb = cp_shape_b[ci] >= 0 ? w->body_link[c->shape_body[cp_shape_b[ci]]] : -1;
    const double* P = cp_P[ci];
    double depth = cp_depth[ci];
    /* DART ContactConstraint tangent basis: t1 = normalize(z x n) (x x n when z and n are parallel), t2 = n x t1 */
    const double* nrm = cp_n[ci];
    static const double ez[3] = {0, 0, 1}, ex[3] = {1, 0, 0};
    double t1[3], t2[3];
    cross3(ez, nrm, t1);
    if (dot3(t1, t1) < 1e-12) cross3(ex, nrm, t1);
    { double l = sqrt(dot3(t1, t1)); t1[0] /= l; t1[1] /= l; t1[2] /= l; }
    cross3(nrm, t1, t2);
    const double* dirs[3] = {nrm, t1, t2};
    int base = m;
    for (int k2 = 0; k2 < 3; k2++) {
      point_jacobian(w, li, P, dirs[k2], J[m]);
      if (lb >= 0) {   /* relative motion of the two links: J = J_a - J_b */
        double Jb[MAXN];
        point_jacobian(w, lb, P, dirs[k2], Jb);
        for (int k = 0; k < n; k++) J[m][k] -= Jb[k];
      }
      double rel = 0, nn = 0;
      for (int k = 0; k < n; k++) { rel += J[m][k] * vs[k]; nn += J[m][k] * J[m][k]; }
      if (k2 == 0) {
        double bounce = depth * c->erp / dt;
        if (bounce > c->max_erv) bounce = c->max_erv;
        b[m] = -rel + bounce; lo[m] = 0; hi[m] = INFINITY; findex[m] = -1;
      } else {
        if (w->planar_drop_z && nn == 0.0) continue; /* out-of-plane direction of a planar model */
        b[m] = -rel; lo[m] = -c->friction; hi[m] = c->friction; findex[m] = base;
      }
      m++;
    }
    double* cl = w->contact_last[w->ncontacts_last++];
    cl[0] = c->shape_body[s]; cl[1] = P[0]; cl[2] = P[1]; cl[3] = P[2]; cl[4] = depth; cl[5] = base;
    {
      double* cd = w->contact_dirs[w->ncontacts_last - 1];
      double* cr = w->contact_report[w->ncontacts_last - 1];
      for (int a = 0; a < 3; a++) { cd[a] = nrm[a]; cd[3 + a] = t1[a]; cd[6 + a] = t2[a]; cr[2 + a] = P[a]; cr[5 + a] = 0; }
      cr[0] = c->shape_body[s]; cr[1] = cp_shape_b[ci] >= 0 ? c->shape_body[cp_shape_b[ci]] : -1;
    }
  }
  const int contact_rows = m; /* rows [0, contact_rows) belong to contacts, the rest to joint limits */
  for (int i = 0; i < n; i++) {
    if (!c->limited[i]) continue;
    int side = 0;
    double viol = 0;
    if (w->qi[i] <= c->lower[i]) { side = -1; viol = w->qi[i] - c->lower[i]; }
    else if (w->qi[i] >= c->upper[i]) { side = +1; viol = w->qi[i] - c->upper[i]; }
    if (!side) continue;
    memset(J[m], 0, sizeof J[m]);
    J[m][i] = 1.0;
    double bounce = -viol * c->limit_erp / dt;
    if (bounce > c->max_erv) bounce = c->max_erv;
    if (bounce < -c->max_erv) bounce = -c->max_erv;
    b[m] = -vs[i] + bounce;
    if (side < 0) { lo[m] = 0; hi[m] = INFINITY; } else { lo[m] = -INFINITY; hi[m] = 0; }
    findex[m] = -1;
    m++;
  }
  /* DART JointCoulombFrictionConstraint: drive the joint velocity to zero with an impulse within +-friction * dt */
  for (int i = 0; i < n; i++) {
    if (!(c->joint_friction[i] != 0.0)) continue;
    memset(J[m], 0, sizeof J[m]);
    J[m][i] = 1.0;
    b[m] = -vs[i];
    lo[m] = -c->joint_friction[i] * dt; hi[m] = c->joint_friction[i] * dt;
    findex[m] = -1;
    m++;
  }
  w->m_last = m;
  w->lcp_residual_last = 0;
  for (int k = 0; k < n; k++) w->cf_last[k] = 0;
  if (m > 0) {
    for (int i = 0; i < m; i++) {
      memcpy(Y[i], J[i], n * sizeof(double));
      chol_solve(HI, n, Y[i]);
    }
    for (int i = 0; i < m; i++)
      for (int j = 0; j < m; j++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += J[i][k] * Y[j][k];
        A[i * m + j] = s;
      }
    for (int i = 0; i < m; i++) A[i * m + i] *= (1.0 + (i < contact_rows ? c->contact_cfm : c->cfm));
    memcpy(w->A_last, A, (size_t)m * m * sizeof(double)); memcpy(w->b_last, b, m * sizeof(double));
    /* stage 1: rows without findex */
    int idx1[MAXM], m1 = 0, idx2[MAXM];
    for (int i = 0; i < m; i++) { x[i] = 0; idx2[i] = i; if (findex[i] < 0) idx1[m1++] = i; }
    int has_fric = (m1 != m);
    if (w->solver == 0) {
      if (blcp_exact(A, b, lo, hi, idx1, m1, m, x) != 0) return -2;
    } else {
      pgs_sweeps(A, b, lo, hi, idx1, m1, m, w->pgs_k1, x); /* device: stage 2 only runs for envs with a contact */
    }
    if (has_fric) {
      /* ODE lcp.cpp: at the first findex row, hi = |hi * x[findex]|, lo = -hi (0 if the normal impulse is 0) */
      for (int i = 0; i < m; i++)
        if (findex[i] >= 0) { double wf = x[findex[i]]; hi[i] = fabs(hi[i] * wf); lo[i] = -hi[i]; }
      if (w->solver == 0) {
        if (blcp_exact(A, b, lo, hi, idx2, m, m, x) != 0) return -3;
      } else {
        pgs_sweeps(A, b, lo, hi, idx2, m, m, w->pgs_k2, x);
      }
      for (int pass = 0; w->a7_iterate_bounds && w->solver == 0 && pass < 20; pass++) {   /* sensitivity study only */
        double moved = 0;
        for (int i = 0; i < m; i++)
          if (findex[i] >= 0) { double nh = fabs(c->friction * x[findex[i]]); moved = fmax(moved, fabs(nh - hi[i])); hi[i] = nh; lo[i] = -nh; }
        if (moved < 1e-12) break;
        if (blcp_exact(A, b, lo, hi, idx2, m, m, x) != 0) return -3;
      }
    }
    for (int i = 0; i < m; i++) {
      for (int k = 0; k < n; k++) { vs[k] += Y[i][k] * x[i]; w->cf_last[k] += J[i][k] * x[i] / dt; }
    }
    if (w->free_root) {   /* internal root coordinates are world-frame: DART's are body-frame, tau_b = R^T tau_w */
      double t[6];
      for (int g = 0; g < 6; g += 3)
        for (int a = 0; a < 3; a++) t[g + a] = w->R0[a] * w->cf_last[g] + w->R0[3 + a] * w->cf_last[g + 1] + w->R0[6 + a] * w->cf_last[g + 2];
      memcpy(w->cf_last, t, sizeof t);
    }
    /* diagnostics: complementarity residual */
    double res = 0;
    for (int i = 0; i < m; i++) {
      double wv = -b[i];
      for (int j = 0; j < m; j++) wv += A[i * m + j] * x[j];
      w->lambda_last[i] = x[i]; w->w_last[i] = wv; w->lo_last[i] = lo[i]; w->hi_last[i] = hi[i];
      double e;
      if (lo[i] == hi[i]) e = 0; /* pinned row (friction under a zero normal impulse): any w is admissible */
      else if (x[i] <= lo[i]) e = wv < 0 ? -wv : 0;
      else if (x[i] >= hi[i]) e = wv > 0 ? wv : 0;
      else e = fabs(wv);
      if (e > res) res = e;
    }
    w->lcp_residual_last = res;
    for (int k = 0; k < w->ncontacts_last; k++) {
      int base = (int)w->contact_last[k][5];
      w->contact_last[k][5] = x[base];
      w->contact_last[k][6] = (base + 1 < m && findex[base + 1] == base) ? x[base + 1] : 0.0;
      w->contact_last[k][7] = (base + 2 < m && findex[base + 2] == base) ? x[base + 2] : 0.0;
      for (int a = 0; a < 3; a++)   /* DART ContactConstraint: force on the first body = (n l_n + t1 l_1 + t2 l_2) / dt */
        w->contact_report[k][5 + a] = (w->contact_dirs[k][a] * w->contact_last[k][5] + w->contact_dirs[k][3 + a] * w->contact_last[k][6] +
                                       w->contact_dirs[k][6 + a] * w->contact_last[k][7]) / dt;
    }
  }
  if (w->free_root) {
    free_root_advance(w, vs, dt);
    for (int i = 6; i < n; i++) { w->dq[i] = vs[i]; w->q[i] += dt * vs[i]; }
    for (int i = 0; i < n; i++) w->tau[i] = 0;
  } else {
    for (int i = 0; i < n; i++) { w->dq[i] = vs[i]; w->q[i] += dt * vs[i]; w->tau[i] = 0; }
  }
  if (w->ext_all) { memset(w->ext_fb, 0, sizeof w->ext_fb); w->ext_all = 0; }   /* like DART: per-step external forces */
  w->time += dt;
  return 0;
}

/* ------------------------------------------------------------------ introspection (known-answer tests) */
/* pydart2 skel.M / skel.c (walker3d_spd.py:40-55) in DART's generalized coordinates.  For a FreeJoint root the internal chain's rates
 * are dqi = T dq with T = blockdiag(R0, R0, I) (world-frame angular velocity and origin velocity from the body-frame twist) and its
 * accelerations are qddi = T qdd + a0, a0 = Tdot dq = (-(rb rc), ra rc, -(ra rb); w x pdot; 0 ...) at the chart centre (r = dqi[0:3] =
 * w, pdot = dqi[3:6]; the same terms as free_root_advance / the kernel's sp_free_root_velocity_correction); generalized forces map by
 * virtual work, tau = T^T taui.  Hence  M = T^T Mi T  and  c = T^T (ci + Mi a0). */
static void free_root_to_dart(const OracleWorld* w, double* M /* n x n or NULL */, double* c /* n or NULL */, const double* Mi) {
  const int n = w->n;
  const double* R = w->R0;
  if (c) {
    const double ra = w->dqi[0], rb = w->dqi[1], rc = w->dqi[2], px = w->dqi[3], py = w->dqi[4], pz = w->dqi[5];
    const double a0[6] = {-(rb * rc), ra * rc, -(ra * rb), rb * pz - rc * py, rc * px - ra * pz, ra * py - rb * px};
    double t[MAXN];
    for (int i = 0; i < n; i++) { t[i] = c[i]; for (int k = 0; k < 6; k++) t[i] += Mi[i * n + k] * a0[k]; }
    for (int g = 0; g < 6; g += 3)
      for (int a = 0; a < 3; a++) c[g + a] = R[a] * t[g] + R[3 + a] * t[g + 1] + R[6 + a] * t[g + 2];   /* R^T */
    for (int i = 6; i < n; i++) c[i] = t[i];
  }
  if (M) {
    static __thread double X[MAXN * MAXN];
    for (int i = 0; i < n; i++) {            /* X = Mi T */
      for (int g = 0; g < 6; g += 3)
        for (int b = 0; b < 3; b++) X[i * n + g + b] = Mi[i * n + g] * R[b] + Mi[i * n + g + 1] * R[3 + b] + Mi[i * n + g + 2] * R[6 + b];
      for (int j = 6; j < n; j++) X[i * n + j] = Mi[i * n + j];
    }
    for (int j = 0; j < n; j++) {            /* M = T^T X */
      for (int g = 0; g < 6; g += 3)
        for (int a = 0; a < 3; a++) M[(g + a) * n + j] = R[a] * X[g * n + j] + R[3 + a] * X[(g + 1) * n + j] + R[6 + a] * X[(g + 2) * n + j];
      for (int i = 6; i < n; i++) M[i * n + j] = X[i * n + j];
    }
  }
}
void oracle_mass_matrix(OracleWorld* w, double* M) {
  kinematics(w); crba(w, w->M);
  if (w->free_root) free_root_to_dart(w, M, NULL, w->M); else memcpy(M, w->M, w->n * w->n * sizeof(double));
}
void oracle_bias(OracleWorld* w, double* C) {
  kinematics(w); rnea(w, w->dqi, NULL, 1, C);
  if (w->free_root) { crba(w, w->M); free_root_to_dart(w, NULL, C, w->M); }
}
void oracle_inverse_dynamics(OracleWorld* w, const double* dq, const double* ddq, int with_gravity, double* tau) {
  kinematics(w); rnea(w, dq, ddq, with_gravity, tau);
}
void oracle_body_pose(OracleWorld* w, int body, double* T16) { kinematics(w); memcpy(T16, w->W[w->body_link[body]], 16 * sizeof(double)); }
/* bodynode.add_ext_force(f) for the NEXT world step only (world frame, at the body frame origin) */
void oracle_add_body_force(OracleWorld* w, int body, const double* f3) {
  for (int a = 0; a < 3; a++) w->ext_fb[body][a] += f3[a];
  w->ext_all = 1;
}
/* pydart2 bodynode.com_spatial_velocity(): [angular; linear velocity of the COM], world frame */
void oracle_body_com_spatial_velocity(OracleWorld* w, int body, double* out6) {
  kinematics(w);
  int li = w->body_link[body];
  const double* T = w->W[li];
  const double* cb = w->card.com[body];
  double cm[3];
  for (int a = 0; a < 3; a++) cm[a] = T[4 * a] * cb[0] + T[4 * a + 1] * cb[1] + T[4 * a + 2] * cb[2] + T[4 * a + 3];
  for (int a = 0; a < 6; a++) out6[a] = 0;
  for (int j = li; j >= 0; j = w->L[j].parent) {
    Link* l = &w->L[j];
    if (l->dof < 0) continue;
    const double* Wm = w->W[j];
    double ww[3], vw[3], rel[3], t[3];
    for (int a = 0; a < 3; a++) {
      ww[a] = Wm[4 * a] * l->S[0] + Wm[4 * a + 1] * l->S[1] + Wm[4 * a + 2] * l->S[2];
      vw[a] = Wm[4 * a] * l->S[3] + Wm[4 * a + 1] * l->S[4] + Wm[4 * a + 2] * l->S[5];
      rel[a] = cm[a] - Wm[4 * a + 3];
    }
    cross3(ww, rel, t);
    for (int a = 0; a < 3; a++) { out6[a] += ww[a] * w->dqi[l->dof]; out6[3 + a] += (vw[a] + t[a]) * w->dqi[l->dof]; }
  }
}
void oracle_body_com(OracleWorld* w, int body, double* out3) {
  kinematics(w);
  const double* T = w->W[w->body_link[body]];
  const double* cm = w->card.com[body];
  for (int a = 0; a < 3; a++) out3[a] = T[4 * a] * cm[0] + T[4 * a + 1] * cm[1] + T[4 * a + 2] * cm[2] + T[4 * a + 3];
}
int oracle_last_lcp(const OracleWorld* w, double* lambda, double* wv, double* lo, double* hi, double* residual) {
  for (int i = 0; i < w->m_last; i++) { lambda[i] = w->lambda_last[i]; wv[i] = w->w_last[i]; lo[i] = w->lo_last[i]; hi[i] = w->hi_last[i]; }
  *residual = w->lcp_residual_last;
  return w->m_last;
}
int oracle_last_Ab(const OracleWorld* w, double* A, double* b) {
  memcpy(A, w->A_last, (size_t)w->m_last * w->m_last * sizeof(double)); memcpy(b, w->b_last, w->m_last * sizeof(double));
  return w->m_last;
}
int oracle_last_contacts(const OracleWorld* w, double* out8) {
  memcpy(out8, w->contact_last, w->ncontacts_last * 8 * sizeof(double));
  return w->ncontacts_last;
}
/* world.collision_result.contacts of the last world step: per contact {body a, body b (-1 = ground), point[3], force on a[3]} */
int oracle_contact_report(const OracleWorld* w, double* out8) {
  memcpy(out8, w->contact_report, w->ncontacts_last * 8 * sizeof(double));
  return w->ncontacts_last;
}
/* total mechanical energy (kinetic + gravitational potential) */
double oracle_energy(OracleWorld* w) {
  kinematics(w); crba(w, w->M);
  double ke = 0, pe = 0;
  for (int i = 0; i < w->n; i++) for (int j = 0; j < w->n; j++) ke += 0.5 * w->dqi[i] * w->M[i * w->n + j] * w->dqi[j];
  for (int b = 0; b < w->card.nbodies; b++) {
    double cm[3];
    const double* T = w->W[w->body_link[b]];
    const double* cb = w->card.com[b];
    for (int a = 0; a < 3; a++) cm[a] = T[4 * a] * cb[0] + T[4 * a + 1] * cb[1] + T[4 * a + 2] * cb[2] + T[4 * a + 3];
    pe -= w->card.mass[b] * dot3(w->card.gravity, cm);
  }
  return ke + pe;
}

/* ------------------------------------------------------------------ task epilogue (hopper.py:24-74, walker2d.py:22-74)
 * One env-step for the planar locomotion tasks; actions already float64.  Returns done. */
/* human_walker.py:157-165: values captured by reset_model after set_state */
void oracle_env_after_reset(OracleWorld* w) {
  if (w->card.task == DART_TASK_HUMANWALKER) {
    double cm[3];
    oracle_body_com(w, w->card.aux_body[1], cm);
    w->init_height = cm[1];
  }
}

static void humanwalker_obs(OracleWorld* w, const int* contact_info, double* obs) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  for (int i = 1; i < n; i++) obs[i - 1] = w->q[i];
  for (int i = 0; i < n; i++) {
    double v = w->dq[i];
    obs[n - 1 + i] = v < -c->obs_vel_clip ? -c->obs_vel_clip : (v > c->obs_vel_clip ? c->obs_vel_clip : v);
  }
  obs[2 * n - 1] = contact_info[0];
  obs[2 * n] = contact_info[1];
}

/* DartHumanWalkerEnv.step (human_walker.py:75-138) */
static int humanwalker_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, abs_sum = 0;
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    abs_sum += fabs(a[k]);
  }
  double cm[3];
  oracle_body_com(w, c->aux_body[0], cm);
  double posbefore = cm[0];
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  oracle_body_com(w, c->aux_body[0], cm);
  double posafter = cm[0];
  oracle_body_com(w, c->aux_body[1], cm);
  double height = cm[1], side = cm[2], angle = w->q[3];
  const double* Th = w->W[w->body_link[c->aux_body[1]]];
  /* to_world(e_y) - to_world(0), normalised (a rotation column is already unit) -> arccos of its y / x component */
  double up[3] = {Th[1], Th[5], Th[9]}, fw[3] = {Th[0], Th[4], Th[8]};
  double nu = sqrt(dot3(up, up)), nf = sqrt(dot3(fw, fw));
  double ang_uwd = acos(up[1] / nu), ang_fwd = acos(fw[0] / nf);
  int info[2] = {0, 0};
  for (int k = 0; k < w->ncontacts_last; k++) {  /* contacts of the last world step (pydart2 collision_result) */
    if ((int)w->contact_last[k][0] == c->aux_body[2]) info[0] = 1;
    if ((int)w->contact_last[k][0] == c->aux_body[3]) info[1] = 1;
  }
  double envdt = c->dt * c->frame_skip;
  double vel = (posafter - posbefore) / envdt;
  double tv = c->aux_real[0];
  double vel_rew = 2 * (tv - fabs(tv - vel));
  double action_pen = c->aux_real[2] * abs_sum;
  double deviation_pen = c->aux_real[3] * fabs(side);
  double r = vel_rew + c->aux_real[1] - action_pen - deviation_pen;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!(height - w->init_height > c->aux_real[4] && height - w->init_height < c->aux_real[5] &&
        fabs(ang_uwd) < c->angle_max && fabs(ang_fwd) < c->angle_max && fabs(angle) < c->aux_real[6] &&
        fabs(w->q[5]) < c->aux_real[7] && fabs(side) < c->aux_real2[0]))
    ok = 0;
  if (!ok) r = 0;
  *reward = r;
  humanwalker_obs(w, info, obs);
  return !ok;
}

/* DartWalker3dEnv.step (walker3d.py:44-97).  All body quantities are those of bodynodes[0] (aux_body[0]). */
static void walker3d_obs(OracleWorld* w, double* obs) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  for (int i = 1; i < n; i++) obs[i - 1] = w->q[i];
  for (int i = 0; i < n; i++) {
    double v = w->dq[i];
    obs[n - 1 + i] = v < -c->obs_vel_clip ? -c->obs_vel_clip : (v > c->obs_vel_clip ? c->obs_vel_clip : v);
  }
}
static int walker3d_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, sq = 0;
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq += a[k] * a[k];
  }
  double cm[3];
  oracle_body_com(w, c->aux_body[0], cm);
  double posbefore = cm[0];
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  oracle_body_com(w, c->aux_body[0], cm);
  double posafter = cm[0], height = cm[1], side = cm[2];
  const double* Tb = w->W[w->body_link[c->aux_body[0]]];
  double up[3] = {Tb[1], Tb[5], Tb[9]}, fw[3] = {Tb[0], Tb[4], Tb[8]};
  double ang_uwd = acos(up[1] / sqrt(dot3(up, up))), ang_fwd = acos(fw[0] / sqrt(dot3(fw, fw)));
  double pen = 0;
  for (int k = 1; k <= 2; k++) {
    int j = c->aux_body[k];
    if ((c->lower[j] - w->q[j]) > -c->penalty_margin) pen += 1.5;
    if ((c->upper[j] - w->q[j]) < c->penalty_margin) pen += 1.5;
  }
  double envdt = c->dt * c->frame_skip;
  double r = (posafter - posbefore) / envdt + c->alive_bonus;
  r -= c->ctrl_cost * sq;
  r -= c->limit_penalty * pen;
  r -= c->aux_real[0] * fabs(side);
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!(height > c->height_lo && height < c->height_hi && fabs(ang_uwd) < c->angle_max && fabs(ang_fwd) < c->angle_max)) ok = 0;
  if (!ok) r = 0;
  *reward = r;
  walker3d_obs(w, obs);
  return !ok;
}

/* DartCartPoleEnv.step (cart_pole.py:12-24) and DartHalfCheetahEnv.step (half_cheetah.py:28-78) */
static void qdq_obs(OracleWorld* w, int skip_first, double* obs) {
  int n = w->n, o = 0;
  for (int i = skip_first; i < n; i++) obs[o++] = w->q[i];
  for (int i = 0; i < n; i++) obs[o++] = w->dq[i];
}
static int cartpole_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  double tau[MAXN] = {0};
  tau[c->act_dof0] = a[0] * c->act_scale[0];   /* no clamp (cart_pole.py:16) */
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  qdq_obs(w, 0, obs);
  int ok = 1;
  for (int i = 0; i < 2 * w->n; i++) if (!isfinite(obs[i])) ok = 0;
  if (!(fabs(obs[1]) <= c->angle_max)) ok = 0;
  *reward = c->alive_bonus;
  return !ok;
}
static int halfcheetah_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, sq = 0;
  double posbefore = w->q[0];
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq += a[k] * a[k];
  }
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  double envdt = c->dt * c->frame_skip;
  double r = (w->q[0] - posbefore) / envdt * 1.0;   /* velrew_weight = 1 (half_cheetah.py:12) */
  r += c->alive_bonus * 1;
  r -= c->ctrl_cost * sq;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!ok) r = 0;
  *reward = r;
  int done = !(ok && fabs(w->q[2]) < c->angle_max);
  qdq_obs(w, 1, obs);
  return done;
}

/* DartCartPoleSwingUpEnv.step (cartpole_swingup.py:14-33) */
static int swingup_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  double tau[MAXN] = {0};
  tau[c->act_dof0] = a[0] * c->act_scale[0];
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  qdq_obs(w, 0, obs);
  double ang = w->q[1];
  double ang_cost = 1.0 * fabs(ang), quad_ctrl_cost = c->aux_real[1] * (a[0] * a[0]), com_cost = c->aux_real[2] * fabs(w->q[0]);
  *reward = c->aux_real[0] - ang_cost - quad_ctrl_cost - com_cost;
  return (fabs(ang) > c->aux_real[3]) || (fabs(w->dq[1]) > c->aux_real[4]) || (fabs(w->q[0]) > c->aux_real[5]);
}
/* DartDoubleInvertedPendulumEnv.step / _get_obs (inverted_double_pendulum.py:19-53) */
static void double_pendulum_obs(OracleWorld* w, double* obs) {
  obs[0] = w->q[0]; obs[1] = sin(w->q[1]); obs[2] = sin(w->q[2]); obs[3] = cos(w->q[1]); obs[4] = cos(w->q[2]);
  obs[5] = w->dq[0]; obs[6] = w->dq[1]; obs[7] = w->dq[2];
}
static int double_pendulum_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  double tau[MAXN] = {0};
  tau[c->act_dof0] = a[0] * c->act_scale[0];
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  double_pendulum_obs(w, obs);
  kinematics(w);
  double base = w->W[w->body_link[c->aux_body[0]]][4 * 1 + 3], raw = w->W[w->body_link[c->aux_body[1]]][4 * 1 + 3];
  double height = 2.0 * (raw - base - c->aux_real[4]) / c->aux_real[5];
  double v1 = w->dq[1], v2 = w->dq[2];
  double dist_penalty = c->aux_real[1] * (obs[0] * obs[0]) + (height - 2.) * (height - 2.);
  double vel_penalty = c->aux_real[2] * (v1 * v1) + c->aux_real[3] * (v2 * v2);
  *reward = c->aux_real[0] - dist_penalty - vel_penalty;
  return height <= 1;
}

/* DartSnake7LinkEnv (snake_7link.py:35-96).  do_simulation adds, before EVERY world step, a fluid force to EVERY body:
 * with n = the body's z axis in the world and v = its COM velocity, vel_pos/neg = v +- (w x n) 0.05 have the same
 * component along n as v itself, so the force is -50 (v . n) n whenever that component is non-zero. */
static int snake_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, sq = 0;
  double posbefore = w->q[0];
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq += a[k] * a[k];
  }
  for (int f = 0; f < c->frame_skip; f++) {
    kinematics(w);
    for (int bdy = 0; bdy < c->nbodies; bdy++) {
      int li = w->body_link[bdy];
      const double* Wm = w->W[li];
      double nd[3] = {Wm[2], Wm[6], Wm[10]}, cm[3], Jd[MAXN], vn = 0;
      for (int x = 0; x < 3; x++) cm[x] = Wm[4 * x] * c->com[bdy][0] + Wm[4 * x + 1] * c->com[bdy][1] + Wm[4 * x + 2] * c->com[bdy][2] + Wm[4 * x + 3];
      point_jacobian(w, li, cm, nd, Jd);
      for (int i = 0; i < n; i++) vn += Jd[i] * w->dq[i];
      for (int x = 0; x < 3; x++) w->ext_fb[bdy][x] = (vn != 0.0) ? -c->aux_real[3] * vn * nd[x] : 0.0;
    }
    w->ext_all = 1;
    oracle_set_forces(w, tau);
    oracle_step(w);
  }
  double envdt = c->dt * c->frame_skip;
  double r = (w->q[0] - posbefore) / envdt;
  r += c->aux_real[0];
  r -= c->aux_real[1] * sq;
  r -= fabs(w->q[2]) * c->aux_real[2];
  *reward = r;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  qdq_obs(w, 1, obs);
  return !(ok && fabs(w->q[2]) < c->angle_max);
}

/* DartReacherEnv (reacher.py:13-45, DartReacher3d-v1) and DartReacher2dEnv (reacher2d.py:18-45, DartReacher-v1).
 * tip = to_world(aux_body[0], aux_real[0..2]); target = task_state[0..2]. */
static void reacher_tip(OracleWorld* w, double* tip) {
  const DartModelCard* c = &w->card;
  kinematics(w);
  const double* T = w->W[w->body_link[c->aux_body[0]]];
  for (int a = 0; a < 3; a++) tip[a] = T[4 * a] * c->aux_real[0] + T[4 * a + 1] * c->aux_real[1] + T[4 * a + 2] * c->aux_real[2] + T[4 * a + 3];
}
static void reacher_obs(OracleWorld* w, double* obs) {
  const DartModelCard* c = &w->card;
  int n = w->n, o = 0;
  double tip[3];
  reacher_tip(w, tip);
  for (int i = 0; i < n; i++) obs[o++] = cos(w->q[i]);
  for (int i = 0; i < n; i++) obs[o++] = sin(w->q[i]);
  if (c->task == DART_TASK_REACHER2D) { obs[o++] = w->task_state[0]; obs[o++] = w->task_state[2]; }
  else for (int a = 0; a < 3; a++) obs[o++] = w->task_state[a];
  for (int i = 0; i < n; i++) obs[o++] = w->dq[i];
  for (int a = 0; a < 3; a++) obs[o++] = tip[a] - w->task_state[a];
}
static int reacher_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, sq_a = 0, sq_tau = 0, tip[3], vec[3];
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq_a += a[k] * a[k]; sq_tau += tau[c->act_dof0 + k] * tau[c->act_dof0 + k];
  }
  reacher_tip(w, tip);
  for (int x = 0; x < 3; x++) vec[x] = tip[x] - w->task_state[x];
  double dist_before = sqrt(dot3(vec, vec));
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  reacher_obs(w, obs);
  if (c->task == DART_TASK_REACHER3D) {
    double reward_dist = -dist_before, reward_ctrl = -sq_tau * c->aux_real[3];
    *reward = reward_dist + reward_ctrl + 0;
    int fin = 1;
    for (int i = 0; i < n; i++) if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) fin = 0;
    return !(fin && (-reward_dist > c->aux_real[4]));
  }
  reacher_tip(w, tip);
  for (int x = 0; x < 3; x++) vec[x] = tip[x] - w->task_state[x];
  *reward = -sqrt(dot3(vec, vec)) + -sq_a;
  return 0;
}

/* DartWalker3dSPDEnv.step (walker3d_spd.py:40-113).  _spd runs before EVERY world step with the env dt (0.008) in the law. */
static void spd_torque(OracleWorld* w, const double* target, double* tau) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double envdt = c->dt * c->frame_skip;
  static __thread double A2[MAXN * MAXN];
  double rhs2[MAXN], p[MAXN], d[MAXN], cb[MAXN];
  kinematics(w);
  crba(w, w->M);
  rnea(w, w->dq, NULL, 1, cb);
  memcpy(A2, w->M, n * n * sizeof(double));
  for (int i = 0; i < n; i++) {
    A2[i * n + i] += c->spd_kd[i] * envdt;
    p[i] = -c->spd_kp[i] * (w->q[i] + w->dq[i] * envdt - target[i]);
    d[i] = -c->spd_kd[i] * w->dq[i];
    rhs2[i] = -cb[i] + p[i] + d[i] + w->cf_last[i];
  }
  cholesky(A2, n);
  chol_solve(A2, n, rhs2);   /* qddot */
  for (int i = 0; i < n; i++) tau[i] = p[i] + d[i] - c->spd_kd[i] * rhs2[i] * envdt;
  for (int i = 0; i < c->act_dof0; i++) tau[i] = 0;
  for (int k = 0; k < c->act_dim; k++) {
    double lim = c->act_scale[k];
    if (fabs(tau[c->act_dof0 + k]) > lim) tau[c->act_dof0 + k] = (tau[c->act_dof0 + k] > 0 ? 1.0 : -1.0) * lim;
  }
}
static int walker3d_spd_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double target[MAXN] = {0}, tau[MAXN], sq = 0;
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    int dd = c->act_dof0 + k;
    target[dd] = (cl + 1.0) / 2.0 * (c->upper[dd] - c->lower[dd]) + c->lower[dd];
    sq += a[k] * a[k];
  }
  double cm[3];
  oracle_body_com(w, c->aux_body[0], cm);
  double posbefore = cm[0];
  for (int f = 0; f < c->frame_skip; f++) { spd_torque(w, target, tau); oracle_set_forces(w, tau); oracle_step(w); }
  oracle_body_com(w, c->aux_body[0], cm);
  double posafter = cm[0], height = cm[1], side = cm[2];
  const double* Tb = w->W[w->body_link[c->aux_body[0]]];
  double up[3] = {Tb[1], Tb[5], Tb[9]}, fw[3] = {Tb[0], Tb[4], Tb[8]};
  double ang_uwd = acos(up[1] / sqrt(dot3(up, up))), ang_fwd = acos(fw[0] / sqrt(dot3(fw, fw)));
  double envdt = c->dt * c->frame_skip;
  double vel_rew = c->aux_real[1] * (posafter - posbefore) / envdt;
  double action_pen = c->ctrl_cost * sq, deviation_pen = c->aux_real[0] * fabs(side);
  *reward = vel_rew + c->alive_bonus - action_pen - deviation_pen;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!(height > c->height_lo && height < c->height_hi && fabs(ang_uwd) < c->angle_max && fabs(ang_fwd) < c->angle_max)) ok = 0;
  walker3d_obs(w, obs);
  return !ok;
}

/* DartDogEnv.step (dog.py:18-48).  q / dq are DART's FreeJoint coordinates (see OracleWorld.free_root). */
static int dog_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  int n = w->n;
  double tau[MAXN] = {0}, sq = 0, cm[3];
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq += a[k] * a[k];
  }
  oracle_body_com(w, c->aux_body[0], cm);
  double posbefore = cm[0];
  for (int f = 0; f < c->frame_skip; f++) { oracle_set_forces(w, tau); oracle_step(w); }
  oracle_body_com(w, c->aux_body[0], cm);
  double posafter = cm[0], height = cm[1], side = fabs(cm[2]);
  double envdt = c->dt * c->frame_skip;
  double r = c->aux_real[0] * (posafter - posbefore) / envdt;
  r += c->alive_bonus;
  r -= c->ctrl_cost * sq;
  *reward = r;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!(height > c->height_lo && height < c->height_hi && side < c->aux_real[1])) ok = 0;
  walker3d_obs(w, obs);   /* q[1:], clip(dq) */
  return !ok;
}

int oracle_env_step(OracleWorld* w, const double* a, double* obs, double* reward) {
  const DartModelCard* c = &w->card;
  if (c->task == DART_TASK_DOG) return dog_step(w, a, obs, reward);
  if (c->task == DART_TASK_WALKER3D_SPD) return walker3d_spd_step(w, a, obs, reward);
  if (c->task == DART_TASK_REACHER2D || c->task == DART_TASK_REACHER3D) return reacher_step(w, a, obs, reward);
  if (c->task == DART_TASK_SNAKE) return snake_step(w, a, obs, reward);
  if (c->task == DART_TASK_CARTPOLE_SWINGUP) return swingup_step(w, a, obs, reward);
  if (c->task == DART_TASK_DOUBLE_PENDULUM) return double_pendulum_step(w, a, obs, reward);
  if (c->task == DART_TASK_CARTPOLE) return cartpole_step(w, a, obs, reward);
  if (c->task == DART_TASK_HALFCHEETAH) return halfcheetah_step(w, a, obs, reward);
  if (c->task == DART_TASK_HUMANWALKER) return humanwalker_step(w, a, obs, reward);
  if (c->task == DART_TASK_WALKER3D) return walker3d_step(w, a, obs, reward);
  int n = w->n;
  double tau[MAXN] = {0};
  double sq = 0;
  for (int k = 0; k < c->act_dim; k++) {
    double cl = a[k];
    if (cl > c->act_high[k]) cl = c->act_high[k];
    if (cl < c->act_low[k]) cl = c->act_low[k];
    tau[c->act_dof0 + k] = cl * c->act_scale[k];
    sq += a[k] * a[k]; /* reward uses the UNclamped action (hopper.py:55) */
  }
  double posbefore = w->q[0];
  int rc = 0;
  for (int f = 0; f < c->frame_skip; f++) {
    oracle_set_forces(w, tau);  /* forces are cleared by every world step (dart_env.py:174) */
    rc |= oracle_step(w);
  }
  if (c->task == DART_TASK_NONE) {   /* physics only (include/dart_model_card.h): obs = [q, dq], reward 0, never done */
    qdq_obs(w, 0, obs);
    *reward = 0.0;
    return 0;
  }
  double posafter = w->q[0], ang = w->q[2];
  double cm[3];
  oracle_body_com(w, c->height_body, cm);
  double height = cm[1];
  double pen = 0;
  if (c->penalty_dof >= 0) {
    int j = c->penalty_dof;
    if ((c->lower[j] - w->q[j]) > -c->penalty_margin) pen += 1.5;
    if ((c->upper[j] - w->q[j]) < c->penalty_margin) pen += 1.5;
  }
  double envdt = c->dt * c->frame_skip;
  double r = (posafter - posbefore) / envdt;
  r += c->alive_bonus;
  r -= c->ctrl_cost * sq;
  r -= c->limit_penalty * pen;
  *reward = r;
  int ok = 1;
  for (int i = 0; i < n; i++) {
    if (!isfinite(w->q[i]) || !isfinite(w->dq[i])) ok = 0;
    if (i >= 2 && !(fabs(w->q[i]) < c->state_abs_max)) ok = 0;
    if (!(fabs(w->dq[i]) < c->state_abs_max)) ok = 0;
  }
  if (!(height > c->height_lo && height < c->height_hi && fabs(ang) < c->angle_max)) ok = 0;
  for (int i = 1; i < n; i++) obs[i - 1] = w->q[i];
  for (int i = 0; i < n; i++) {
    double v = w->dq[i];
    obs[n - 1 + i] = v < -c->obs_vel_clip ? -c->obs_vel_clip : (v > c->obs_vel_clip ? c->obs_vel_clip : v);
  }
  obs[0] = height;
  (void)rc;
  return !ok;
}
void oracle_env_obs(OracleWorld* w, double* obs) {
  const DartModelCard* c = &w->card;
  if (c->task == DART_TASK_HUMANWALKER) { int z[2] = {0, 0}; humanwalker_obs(w, z, obs); return; } /* reset_model zeroes contact_info */
  if (c->task == DART_TASK_WALKER3D || c->task == DART_TASK_WALKER3D_SPD || c->task == DART_TASK_DOG) { walker3d_obs(w, obs); return; }
  if (c->task == DART_TASK_NONE || c->task == DART_TASK_CARTPOLE || c->task == DART_TASK_CARTPOLE_SWINGUP) { qdq_obs(w, 0, obs); return; }
  if (c->task == DART_TASK_DOUBLE_PENDULUM) { double_pendulum_obs(w, obs); return; }
  if (c->task == DART_TASK_REACHER2D || c->task == DART_TASK_REACHER3D) { reacher_obs(w, obs); return; }
  if (c->task == DART_TASK_HALFCHEETAH || c->task == DART_TASK_SNAKE) { qdq_obs(w, 1, obs); return; }
  int n = w->n;
  double cm[3];
  oracle_body_com(w, c->height_body, cm);
  for (int i = 1; i < n; i++) obs[i - 1] = w->q[i];
  for (int i = 0; i < n; i++) {
    double v = w->dq[i];
    obs[n - 1 + i] = v < -c->obs_vel_clip ? -c->obs_vel_clip : (v > c->obs_vel_clip ? c->obs_vel_clip : v);
  }
  obs[0] = cm[1];
}

/* ------------------------------------------------------------------ Philox4x32-10 reset noise
 * Same counter-based stream as the device auto-reset (dart_env_amd/csrc/planar_kernel.hpp reset_noise):
 * counter = (env lo, env hi, episode, block), key = (seed lo, seed hi), u = (x >> 8) * 2^-24. */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void oracle_philox_noise(uint64_t seed, uint64_t gid, uint32_t ep, double r, double rv, int n, double* q, double* dq) {
  double u[2 * MAXN + 4];
  int nw = 2 * n;
  for (int blk = 0; blk < (nw + 3) / 4; ++blk) {
    uint32_t o[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), ep, (uint32_t)blk, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    for (int j = 0; j < 4; ++j) u[4 * blk + j] = (double)(o[j] >> 8) * (1.0 / 16777216.0);
  }
  for (int i = 0; i < n; i++) { q[i] = -r + 2.0 * r * u[i]; dq[i] = -rv + 2.0 * rv * u[n + i]; }
}

/* Rollout of `n_envs` independent envs for `steps` env-steps with auto-reset (Philox noise), used as the timed CPU
 * baseline and as the reference for bench.py's RMS-state-error figure.
 *   actions: [steps][n_envs][act_dim] float32 (the same buffer the GPU ring holds)
 *   q_out/dq_out: [n_envs][ndofs] final states; episode_out/elapsed_out: per-env counters (history fingerprint)
 * Returns the number of env-steps executed. */
int64_t oracle_rollout(const DartModelCard* card, int solver, int64_t n_envs, int steps, const float* actions,
                       uint64_t seed, uint64_t env_offset, double* q_out, double* dq_out, uint32_t* episode_out,
                       int32_t* elapsed_out, double* reward_sum_out) {
  OracleWorld* w = oracle_create(card);
  if (!w) return -1;
  w->solver = solver;
  int n = w->n, na = card->act_dim;
  int64_t count = 0;
  double obs[2 * MAXN], a[DART_MAX_ACTIONS];
  for (int64_t e = 0; e < n_envs; e++) {
    uint32_t ep = 1;
    int elapsed = 0;
    double qn[MAXN], vn[MAXN], rs = 0;
    oracle_reset(w);
    oracle_philox_noise(seed, env_offset + (uint64_t)e, ep, card->reset_noise, card->reset_noise_vel, n, qn, vn);
    for (int i = 0; i < n; i++) { w->q[i] += qn[i]; w->dq[i] += vn[i]; }
    oracle_env_after_reset(w);
    for (int t = 0; t < steps; t++) {
      const float* at = actions + ((size_t)t * n_envs + e) * na;
      for (int k = 0; k < na; k++) a[k] = (double)at[k];
      double r;
      int done = oracle_env_step(w, a, obs, &r);
      rs += r;
      elapsed++;
      count++;
      if (card->max_episode_steps > 0 && elapsed >= card->max_episode_steps) done = 1;
      if (done) {
        ep++;
        oracle_reset(w);
        oracle_philox_noise(seed, env_offset + (uint64_t)e, ep, card->reset_noise, card->reset_noise_vel, n, qn, vn);
        for (int i = 0; i < n; i++) { w->q[i] += qn[i]; w->dq[i] += vn[i]; }
        oracle_env_after_reset(w);
        elapsed = 0;
      }
    }
    if (q_out) memcpy(q_out + e * n, w->q, n * sizeof(double));
    if (dq_out) memcpy(dq_out + e * n, w->dq, n * sizeof(double));
    if (episode_out) episode_out[e] = ep;
    if (elapsed_out) elapsed_out[e] = elapsed;
    if (reward_sum_out) reward_sum_out[e] = rs;
  }
  oracle_destroy(w);
  return count;
}

/* Auto-resetting rollout of env range [0, n_envs) with a trace: the done flag of every env-step (before the reset it causes)
 * and the PRE-RESET state after the steps listed in snap_steps (1-based step counts, ascending).  Thread-safe (one world per
 * call, thread-local scratch): the parity harness runs one call per host core.  actions are [steps][act_stride_envs][act]
 * with this call's envs starting at column act_env0. */
int64_t oracle_rollout_trace(const DartModelCard* card, int solver, int64_t n_envs, int steps, const float* actions,
                             int64_t act_stride_envs, int64_t act_env0, uint64_t seed, uint64_t env_offset, int n_snap,
                             const int32_t* snap_steps, double* q_snap, double* dq_snap, uint8_t* done_trace,
                             int64_t trace_stride_envs, uint32_t* episode_out, int32_t* elapsed_out) {
  OracleWorld* w = oracle_create(card);
  if (!w) return -1;
  w->solver = solver;
  int n = w->n, na = card->act_dim;
  int64_t count = 0;
  double obs[2 * MAXN + 8], a[DART_MAX_ACTIONS];
  for (int64_t e = 0; e < n_envs; e++) {
    uint32_t ep = 1;
    int elapsed = 0, si = 0;
    double qn[MAXN], vn[MAXN];
    oracle_reset(w);
    oracle_philox_noise(seed, env_offset + (uint64_t)e, ep, card->reset_noise, card->reset_noise_vel, n, qn, vn);
    for (int i = 0; i < n; i++) { w->q[i] += qn[i]; w->dq[i] += vn[i]; }
    oracle_env_after_reset(w);
    for (int t = 0; t < steps; t++) {
      const float* at = actions + ((size_t)t * act_stride_envs + act_env0 + e) * na;
      for (int k = 0; k < na; k++) a[k] = (double)at[k];
      double r;
      int done = oracle_env_step(w, a, obs, &r);
      elapsed++;
      count++;
      if (card->max_episode_steps > 0 && elapsed >= card->max_episode_steps) done = 1;
      if (done_trace) done_trace[(size_t)t * trace_stride_envs + act_env0 + e] = (uint8_t)(done ? 1 : 0);
      if (si < n_snap && snap_steps[si] == t + 1) {
        memcpy(q_snap + ((size_t)si * trace_stride_envs + act_env0 + e) * n, w->q, n * sizeof(double));
        memcpy(dq_snap + ((size_t)si * trace_stride_envs + act_env0 + e) * n, w->dq, n * sizeof(double));
        si++;
      }
      if (done) {
        ep++;
        oracle_reset(w);
        oracle_philox_noise(seed, env_offset + (uint64_t)e, ep, card->reset_noise, card->reset_noise_vel, n, qn, vn);
        for (int i = 0; i < n; i++) { w->q[i] += qn[i]; w->dq[i] += vn[i]; }
        oracle_env_after_reset(w);
        elapsed = 0;
      }
    }
    if (episode_out) episode_out[act_env0 + e] = ep;
    if (elapsed_out) elapsed_out[act_env0 + e] = elapsed;
  }
  oracle_destroy(w);
  return count;
}
