// spatial_kernel.hpp -- gfx950 device code for general (3-D, branching) skeletons: DartHumanWalker-v1 class models.
//
// Replaces for one env per WAVEFRONT what the reference does per env through pydart2/DART
// (reference gym/envs/dart/human_walker.py:60-165, dart_env.py:158-175).
//
// Design: the 21/29-dof models do not fit one lane's registers (H is 29x29, the contact/limit LCP has up to 40 rows),
// so one 64-lane wavefront owns one environment and the per-skeleton block lives in LDS (~35 KB fp32):
// link frames / velocities / composite inertias, H and its Cholesky factor, the constraint Jacobian (overwritten by
// W = L^-1 J^T), the Delassus matrix A = W W^T and the pivoting solver's LDL^T workspace.  Tree recursions run on lane 0
// (they are a dependent chain), everything dense is spread over the lanes: one lane per dof for the mass-matrix
// rows, one lane per constraint row for Jacobians / triangular solves / A, row-owner right-looking factorisations,
// and the LCP active-set logic is wave-uniform (row infeasibility flags are gathered with __ballot).
// Strides of the LDS matrices are odd so that row-per-lane access is bank-conflict free.
//
// Dynamics formulation: world-aligned recursive Newton-Euler + composite bodies taken about each joint origin
// (coordinates relative to the floating base translation so fp32 does not see the travelled distance) -- again a
// different derivation from the oracle's body-frame spatial algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planar_kernel.hpp"  // philox, sincos_, rcp_, tol_

namespace dartk {

constexpr int SP_MAXL = 48;   // expanded 1-dof links
constexpr int SP_MAXN = 32;   // dofs
constexpr int SP_MAXS = 16;   // collidable shapes
// LCP capacity is a property of the model (SpatialModel::maxm / maxcp): 36 rows / 12 contact points by default
// (HumanWalker peaks at ~31 active rows), 64 rows / 20 points for models with link-link contacts (rows = lanes <= 64)
constexpr int SP_MAXPAIRS = 40;   // non-adjacent shape pairs tested for link-link contacts
__device__ __host__ constexpr int sp_tri(int m) { return m * (m + 1) / 2; }   // packed lower triangle of A / LDL workspace
// The pivoting solver's LDL^T workspace (and its PGS start vector) are live only after the Jacobian rows have been
// built, the per-link records only before: when the link block is big enough the two share LDS (HumanWalker: 2.8 KB
// less per workgroup = 10 instead of 8 workgroups per CU).
__device__ __host__ constexpr bool sp_lw_aliases_links(int nl, int maxm) { return nl * 37 >= sp_tri(maxm) + maxm; }
__device__ __host__ constexpr int TI(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
__device__ __host__ constexpr int sp_npad(int n) { return (n + 7) & ~7; }   // H is stored padded with identity rows to a multiple of 8
__device__ __host__ constexpr int TL(int i, int j) { return i * (i + 1) / 2 + j; }   // caller guarantees i >= j
constexpr int SP_LINKF = 37;  // Reals stored per link in LDS
constexpr int SP_LCONST = 48;   // Rpre 9, ppre 3, Rpost 9, ppost 3, axis 3, com 3, inertia 9, axr 3, cpost 3 (+3 pad)
enum { LC_RPRE = 0, LC_PPRE = 9, LC_RPOST = 12, LC_PPOST = 21, LC_AXIS = 24, LC_COM = 27, LC_INERTIA = 30, LC_AXR = 39, LC_CPOST = 42 };
constexpr int SP_ROUNDS = 6;  // pointer-jumping rounds: trees up to 64 links deep

template <class Real>
struct SpatialModel {
  int nl, n, nshapes;
  int parent[SP_MAXL], jtype[SP_MAXL], dof[SP_MAXL], root_trans[SP_MAXL];
  int pre_ident[SP_MAXL], post_ident[SP_MAXL];   // 1: the fixed transform is the identity (carriers of expanded joints)
  int n_root_trans, root_trans_link[8];   // the root-chain prismatic links (floating-base translation)
  int nrounds;                       // ceil(log2(tree depth)): pointer-jumping rounds of the forward pass
  int anc[SP_MAXL][SP_ROUNDS];       // anc[i][k] = 2^k-th ancestor of link i, -1 beyond the root
  // backward pass: links of one expanded joint share their joint origin, so their composite bodies are identical;
  // only the group's last link (the leader, the one that carries the mass) gathers, level by level over GROUPS
  int link_is_body[SP_MAXL];                         // 1: the link that carries a card body (last link of its joint)
  int group_leader[SP_MAXL], group_level[SP_MAXL];   // group_level: depth of the group for leaders, -1 for the others
  int n_group_levels;
  // the forward pass re-reads its link's geometry from here every substep (48 contiguous Reals per link, 12 x 16-byte
  // loads issued together: one L1/L2-resident latency per substep instead of ~45 VGPRs held for the whole kernel)
  Real lconst[SP_MAXL][SP_LCONST];
  int child_start[SP_MAXL + 1], child_list[SP_MAXL];            // children of every link
  Real axis[SP_MAXL][3];
  Real root_axis_world[SP_MAXL][3];   // world axis of the root-chain prismatic links (constant)
  Real Rpre[SP_MAXL][9], ppre[SP_MAXL][3];    // joint frame in the parent link frame
  Real Rpost[SP_MAXL][9], ppost[SP_MAXL][3];  // child link frame in the (moved) joint frame
  Real mass[SP_MAXL], com[SP_MAXL][3], inertia[SP_MAXL][9];
  int dof_link[SP_MAXN], limited[SP_MAXN];
  Real lower[SP_MAXN], upper[SP_MAXN], damp[SP_MAXN], stiff[SP_MAXN], rest[SP_MAXN], q0[SP_MAXN], dq0[SP_MAXN];
  Real spd_kp[SP_MAXN], spd_kd[SP_MAXN];   // DartWalker3dSPD-v1 stable-PD gains (task 12); act_scale = torque limits
  Real envdt;                        // dt * frame_skip (the SPD law uses the env step, walker3d_spd.py:41-46)
  Real* cf_store;                    // [n_envs][n] generalized constraint forces of each env's last world step (task 12)
  Real jfric_dt[SP_MAXN];            // Coulomb joint friction * dt: impulse bound of the dof's friction row (0 = none)
  int has_joint_friction;
  int free_root;                     // 1: body 0 hangs on a DART FreeJoint (public q[0:3] rotation vector, dq[0:6] body twist)
  int free_link;                     // the last of the six root links (carries the body); its joint rotation is Rz(c) R0
  int maxm, maxcp;                   // LCP rows / contact points this model's LDS block is carved for
  int npairs, pair_a[SP_MAXPAIRS], pair_b[SP_MAXPAIRS];   // link-link contact candidates: shape slots, a < b
  int sh_link[SP_MAXS], sh_type[SP_MAXS];
  Real sh_R[SP_MAXS][9], sh_p[SP_MAXS][3], sh_size[SP_MAXS][3];
  Real dt, g[3], ground_y, mu, erp_dt, max_erv, limit_erp_dt, cfm1, ccfm1;   // ccfm1 = 1 + contact_cfm
  // task
  int task, frame_skip, act_dim, obs_dim, act_dof0, max_steps;
  Real act_scale[32], act_lo[32], act_hi[32];
  int aux_link[4];
  Real aux_real[8], aux_real2[4];
  Real s_max, v_clip, noise, noise_v, inv_envdt;
  int solver_iters, pgs_fallback_sweeps;
  int ext_at_joint_origin;     // 1: the force acts at the link's joint origin (redirected from a massless carrier body)
  int ext_link;                // external body force (dart_set_ext_force): link it acts on, at the link frame origin
  const Real* ext_force;       // [n_envs][3] world-frame force per env, nullptr = none
  int link_body[SP_MAXL];      // card body carried by a link (-1: carrier link of an expanded joint)
  Real* creport;               // optional [n_envs][maxcp][8]: contacts of the last world step {body a, body b, point, force on a}
  int* creport_count;          // [n_envs]
  Real* cf_report;             // [n_envs][n]: constraint_forces() of the last world step (recorded with the contacts)
  double* dbg;                 // optional [n_envs][160] dump of the last LCP (debug builds of the tests only)
  unsigned long long* stats;   // optional [64]: [0..31] pivoting iterations per solve, [32] PGS fallbacks, [33] solves
};

// ---- tiny 3-vector helpers on registers
template <class Real> struct V3 { Real x, y, z; };
template <class Real> __device__ __forceinline__ V3<Real> v3(Real x, Real y, Real z) { return {x, y, z}; }
template <class Real> __device__ __forceinline__ V3<Real> operator+(V3<Real> a, V3<Real> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class Real> __device__ __forceinline__ V3<Real> operator-(V3<Real> a, V3<Real> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class Real> __device__ __forceinline__ V3<Real> operator*(V3<Real> a, Real s) { return {a.x * s, a.y * s, a.z * s}; }
template <class Real> __device__ __forceinline__ Real dot(V3<Real> a, V3<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class Real> __device__ __forceinline__ V3<Real> cross(V3<Real> a, V3<Real> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class Real> __device__ __forceinline__ V3<Real> ld3(const Real* p) { return {p[0], p[1], p[2]}; }
template <class Real> __device__ __forceinline__ void st3(Real* p, V3<Real> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
// y = R x, R row-major 3x3
template <class Real> __device__ __forceinline__ V3<Real> mulR(const Real* R, V3<Real> x) {
  return {R[0] * x.x + R[1] * x.y + R[2] * x.z, R[3] * x.x + R[4] * x.y + R[5] * x.z, R[6] * x.x + R[7] * x.y + R[8] * x.z};
}
template <class Real> __device__ __forceinline__ void mulRR(const Real* A, const Real* B, Real* C) {  // C = A B
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// LDS layout of one link (offsets in Reals)
enum { LK_R = 0, LK_P = 9, LK_JO = 12, LK_A = 15, LK_C = 18, LK_F = 21, LK_N = 24, LK_MC = 27, LK_H = 28, LK_IC = 31 };

template <class Real>
struct SpLds {
  Real* link;    // [nl][SP_LINKF]
  Real* q; Real* dq; Real* tau; Real* rhs;   // [n]
  Real* H;       // [n(n+1)/2] packed lower triangle -> Cholesky factor
  Real* W;       // [maxm+1][n]: constraint Jacobian rows, then W = L^-1 J^T
  Real* A;       // [tri(maxm)] packed symmetric
  Real* Lw;      // [tri(maxm)] packed lower
  Real* b; Real* lo; Real* hi; Real* x; Real* r; Real* x0;   // [maxm]
  int* rdof;     // [maxm] limit rows: dof index, contact rows: -1
  int* rfidx;    // [maxm] friction rows: index of their normal row, else -1
  Real* cpP;     // [maxcp][4]: contact point (relative coords) + depth
  Real* cpN;     // [maxcp][3]: contact normal, pointing into the first link (ground: +y)
  int* cplink;   // [maxcp] first link
  int* cplinkB;  // [maxcp] second link of a link-link contact, -1 for the ground
  Real* sinv;    // [n]: 1 / L_jj of the mass-matrix Cholesky factor
  Real* root;    // [24] free root joint: R (9), p (3), body twist w v (6), Euler X-Y-Z of R (3)
  Real* cf;      // [n]: J^T lambda / dt of the previous world step (pydart2 constraint_forces(), SPD task only)
  Real* misc;    // [16]: roff(3), scalars
  int* imisc;    // [8]: ncp, m, contact flags
  unsigned long long* ticks;   // [10] phase cycle counters of this env-step (diagnostics, only touched when stats are on)
  int* topo;     // [nl]: (parent + 1) | (dof + 1) << 8 | jtype << 16 -- ancestor walks read this instead of global memory
};
__device__ __forceinline__ int topo_parent(int w) { return (w & 0xff) - 1; }
__device__ __forceinline__ int topo_dof(int w) { return ((w >> 8) & 0xff) - 1; }
__device__ __forceinline__ int topo_jtype(int w) { return (w >> 16) & 0xff; }

template <class Real>
__device__ __forceinline__ SpLds<Real> sp_carve(Real* base, int nl, int n, int maxm, int maxcp) {
  SpLds<Real> S;
  Real* p = base;
  S.link = p; p += nl * SP_LINKF;
  S.q = p; p += n; S.dq = p; p += n; S.tau = p; p += n; S.rhs = p; p += n;
  S.H = p; p += sp_npad(n) * (sp_npad(n) + 1) / 2;
  S.W = p; p += (maxm + 1) * n;
  S.A = p; p += sp_tri(maxm);
  if (sp_lw_aliases_links(nl, maxm)) { S.Lw = S.link; S.x0 = S.link + sp_tri(maxm); }
  else { S.Lw = p; p += sp_tri(maxm); S.x0 = p; p += maxm; }
  S.b = p; p += maxm; S.lo = p; p += maxm; S.hi = p; p += maxm; S.x = p; p += maxm; S.r = p; p += maxm;
  S.cpP = p; p += maxcp * 4;
  S.cpN = p; p += maxcp * 3;
  S.misc = p; p += 16;
  S.sinv = p; p += n;
  S.cf = p; p += n;
  S.root = p; p += 24;
  S.rdof = (int*)p; p += maxm * sizeof(int) / sizeof(Real) + 1;
  S.rfidx = (int*)p; p += maxm * sizeof(int) / sizeof(Real) + 1;
  S.cplink = (int*)p; p += maxcp * sizeof(int) / sizeof(Real) + 1;
  S.cplinkB = (int*)p; p += maxcp * sizeof(int) / sizeof(Real) + 1;
  S.imisc = (int*)p;
  S.topo = S.imisc + 8;
  S.ticks = (unsigned long long*)(((size_t)(S.topo + nl) + 7) & ~(size_t)7);
  return S;
}
__host__ __device__ inline size_t sp_lds_bytes(int nl, int n, size_t real_bytes, int maxm, int maxcp) {
  const size_t lw = sp_lw_aliases_links(nl, maxm) ? 0 : (size_t)sp_tri(maxm) + maxm;
  size_t reals = (size_t)nl * SP_LINKF + 6 * n + (size_t)sp_npad(n) * (sp_npad(n) + 1) / 2 + (size_t)(maxm + 1) * n + sp_tri(maxm) + lw + 5 * maxm +
                 maxcp * 7 + 16 + 24;
  return reals * real_bytes + (2 * maxm + 2 * maxcp + 8 + nl) * sizeof(int) + 4 * real_bytes + 16 + 10 * sizeof(unsigned long long);
}

// ------------------------------------------------------------------ lane-0 recursions
// forward kinematics (positions relative to the floating-base translation `roff`)
template <class Real>
__device__ __forceinline__ void sp_kinematics(const SpatialModel<Real>& Md, SpLds<Real>& S, int only_link = -1) {
  V3<Real> roff = v3<Real>(0, 0, 0);
  for (int i = (only_link >= 0 ? only_link : 0); i < (only_link >= 0 ? only_link + 1 : Md.nl); i++) {
    Real* L = S.link + i * SP_LINKF;
    const int p = Md.parent[i];
    Real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3<Real> pp = v3<Real>(0, 0, 0);
    if (p >= 0) {
      const Real* Lp = S.link + p * SP_LINKF;
      for (int k = 0; k < 9; k++) Rp[k] = Lp[LK_R + k];
      pp = ld3(Lp + LK_P);
    }
    Real Rj[9];
    V3<Real> pj = pp;
    if (Md.pre_ident[i]) { for (int k = 0; k < 9; k++) Rj[k] = Rp[k]; }
    else { mulRR(Rp, Md.Rpre[i], Rj); pj = pp + mulR(Rp, ld3(Md.ppre[i])); }
    V3<Real> ax = ld3(Md.axis[i]);
    V3<Real> a = mulR(Rj, ax);
    Real Rm[9];
    V3<Real> pm = pj;
    const int d = Md.dof[i];
    if (Md.jtype[i] == 2) {  // revolute: Rm = Rj * Rot(axis, q)
      Real sn, cs;
      sincos_<Real>(S.q[d], sn, cs);
      const Real v = Real(1) - cs;
      Real Rq[9] = {ax.x * ax.x * v + cs,        ax.x * ax.y * v - ax.z * sn, ax.x * ax.z * v + ax.y * sn,
                    ax.y * ax.x * v + ax.z * sn, ax.y * ax.y * v + cs,        ax.y * ax.z * v - ax.x * sn,
                    ax.z * ax.x * v - ax.y * sn, ax.z * ax.y * v + ax.x * sn, ax.z * ax.z * v + cs};
      mulRR(Rj, Rq, Rm);
    } else {
      for (int k = 0; k < 9; k++) Rm[k] = Rj[k];
      if (Md.jtype[i] == 1) {
        if (!Md.root_trans[i]) pm = pj + a * S.q[d];
        else if (only_link < 0) roff = roff + a * S.q[d];
      }
    }
    Real Ri[9];
    V3<Real> pi = pm;
    if (Md.free_root && i == Md.free_link) {   // joint rotation Rz(c) R0 (see sp_free_root_to_internal)
      Real T[9];
      mulRR(Rm, S.root, T);
      mulRR(T, Md.Rpost[i], Ri); pi = pm + mulR(T, ld3(Md.ppost[i]));
    } else if (Md.post_ident[i]) { for (int k = 0; k < 9; k++) Ri[k] = Rm[k]; }
    else { mulRR(Rm, Md.Rpost[i], Ri); pi = pm + mulR(Rm, ld3(Md.ppost[i])); }
    for (int k = 0; k < 9; k++) L[LK_R + k] = Ri[k];
    st3(L + LK_P, pi);
    st3(L + LK_JO, pj);
    st3(L + LK_A, a);
    st3(L + LK_C, pi + mulR(Ri, ld3(Md.com[i])));
  }
  if (only_link < 0) st3(S.misc, roff);
}

// per-link model constants, held in the registers of the lane that owns the link for the whole kernel
template <class Real>
struct LinkConst {
  int parent, jtype, dof, root_trans;
  int anc[SP_ROUNDS];
  int group_leader, group_level, is_body;
  int nchild; unsigned long long children;   // leaders: the leaders of up to 8 child groups, one byte each
  Real mass;
  Real damp, stiff, rest;                    // of this link's dof
  // the same lane also owns dof `lane` (mass-matrix row, limits)
  int d_link; Real d_diag;                   // link of dof `lane`; dt*damping + dt^2*stiffness
  int d_limited; Real d_lower, d_upper, d_fric;
};
template <class Real>
__device__ __forceinline__ void sp_load_link_const(const SpatialModel<Real>& Md, int i, LinkConst<Real>& c) {
  c.parent = Md.parent[i]; c.jtype = Md.jtype[i]; c.dof = Md.dof[i]; c.root_trans = Md.root_trans[i];
  for (int k = 0; k < SP_ROUNDS; k++) c.anc[k] = Md.anc[i][k];
  c.group_leader = Md.group_leader[i]; c.group_level = Md.group_level[i]; c.is_body = Md.link_is_body[i];
  c.mass = Md.mass[i];
  c.nchild = Md.child_start[i + 1] - Md.child_start[i];
  c.children = 0ull;
  for (int k = 0; k < c.nchild && k < 8; k++)
    c.children |= (unsigned long long)(Md.group_leader[Md.child_list[Md.child_start[i] + k]] & 0xff) << (8 * k);
  const int d = c.dof >= 0 ? c.dof : 0;
  c.damp = Md.damp[d]; c.stiff = Md.stiff[d]; c.rest = Md.rest[d];
  const int dl = i < Md.n ? i : 0;
  c.d_link = Md.dof_link[dl];
  c.d_diag = Md.dt * Md.damp[dl] + Md.dt * Md.dt * Md.stiff[dl];
  c.d_limited = (i < Md.n) ? Md.limited[dl] : 0; c.d_lower = Md.lower[dl]; c.d_upper = Md.upper[dl];
  c.d_fric = (i < Md.n) ? Md.jfric_dt[dl] : Real(0);
}

template <class Real> __device__ __forceinline__ V3<Real> shfl3(V3<Real> v, int src) {
  return {__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src)};
}

// Forward pass of the whole tree in O(log depth) wave steps (all 64 lanes call; lane i owns link i).
//   1. every lane builds its link's transform relative to the parent link,
//   2. pointer jumping composes them into world transforms: round k folds in the 2^k-th ancestor's partial product,
//      fetched from that lane's registers with ds_bpermute (__shfl) -- no LDS traffic, no level-by-level serialisation,
//   3. angular velocity, velocity-product angular and linear accelerations are path sums of per-link terms
//      (w_i = a_i qd_i;  t_i = om_parent x w_i;  b_i = the centripetal / Coriolis increment): three more prefix sums,
//   4. the link's wrench and composite-body seeds about its own joint origin go to LDS.
template <class Real, bool EXTRAS = false>
__device__ __forceinline__ void sp_forward(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane,
                                           int64_t env = 0) {
  const bool live = lane < Md.nl;
  const bool rev = lc.jtype == 2, slide = lc.jtype == 1 && !lc.root_trans;
  const Real qv = (live && lc.dof >= 0) ? S.q[lc.dof] : Real(0), qd = (live && lc.dof >= 0) ? S.dq[lc.dof] : Real(0);
  Real G[SP_LCONST];   // this link's geometry block
  {
    const Real* g = Md.lconst[live ? lane : 0];
#pragma unroll
    for (int k = 0; k < SP_LCONST; k++) G[k] = g[k];
  }
  const V3<Real> ax = ld3(G + LC_AXIS);
  if (EXTRAS && Md.free_root && lane == Md.free_link) {   // joint rotation Rz(c) R0: fold R0 into the joint-to-child transform
    Real T[9];
    mulRR(S.root, G + LC_RPOST, T);
    const V3<Real> t = mulR(S.root, ld3(G + LC_PPOST));
    for (int k = 0; k < 9; k++) G[LC_RPOST + k] = T[k];
    st3(G + LC_PPOST, t);
    st3(G + LC_AXR, v3<Real>(T[0] * ax.x + T[3] * ax.y + T[6] * ax.z, T[1] * ax.x + T[4] * ax.y + T[7] * ax.z, T[2] * ax.x + T[5] * ax.y + T[8] * ax.z));
  }
  Real R[9];
  V3<Real> p;
  {
    Real sn = Real(0), cs = Real(1);
    if (rev) sincos_<Real>(qv, sn, cs);
    const Real v = Real(1) - cs;
    const Real Rq[9] = {ax.x * ax.x * v + cs,        ax.x * ax.y * v - ax.z * sn, ax.x * ax.z * v + ax.y * sn,
                        ax.y * ax.x * v + ax.z * sn, ax.y * ax.y * v + cs,        ax.y * ax.z * v - ax.x * sn,
                        ax.z * ax.x * v - ax.y * sn, ax.z * ax.y * v + ax.x * sn, ax.z * ax.z * v + cs};
    Real T[9];
    mulRR(Rq, G + LC_RPOST, T);
    V3<Real> t = mulR(Rq, ld3(G + LC_PPOST));
    if (slide) t = t + ax * qv;
    mulRR(G + LC_RPRE, T, R);
    p = ld3(G + LC_PPRE) + mulR(G + LC_RPRE, t);
    if (!live) { for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? Real(1) : Real(0); p = v3<Real>(0, 0, 0); }
  }
  const int nr = Md.nrounds;
#pragma unroll
  for (int k = 0; k < SP_ROUNDS; k++) {
    if (k < nr) {
      const int hop = lc.anc[k], src = hop >= 0 ? hop : lane;
      Real Rh[9];
      for (int c = 0; c < 9; c++) Rh[c] = __shfl(R[c], src);
      const V3<Real> ph = shfl3(p, src);
      if (hop >= 0) {
        Real Rn[9];
        mulRR(Rh, R, Rn);
        p = ph + mulR(Rh, p);
        for (int c = 0; c < 9; c++) R[c] = Rn[c];
      }
    }
  }
  const V3<Real> a = mulR(R, ld3(G + LC_AXR));
  V3<Real> pj = p - mulR(R, ld3(G + LC_CPOST));
  if (slide) pj = pj - a * qv;
  const V3<Real> c = p + mulR(R, ld3(G + LC_COM));
  // angular velocity
  const V3<Real> w = rev ? a * qd : v3<Real>(0, 0, 0);
  V3<Real> om = w;
#pragma unroll
  for (int k = 0; k < SP_ROUNDS; k++) {
    if (k < nr) {
      const int hop = lc.anc[k];
      const V3<Real> t = shfl3(om, hop >= 0 ? hop : lane);
      if (hop >= 0) om = om + t;
    }
  }
  const V3<Real> omp = om - w;
  // velocity-product angular acceleration
  const V3<Real> ta = rev ? cross(omp, w) : v3<Real>(0, 0, 0);
  V3<Real> al = ta;
#pragma unroll
  for (int k = 0; k < SP_ROUNDS; k++) {
    if (k < nr) {
      const int hop = lc.anc[k];
      const V3<Real> t = shfl3(al, hop >= 0 ? hop : lane);
      if (hop >= 0) al = al + t;
    }
  }
  const V3<Real> alp = al - ta;
  // velocity-product linear acceleration of the link origin
  V3<Real> pp = shfl3(p, lc.parent >= 0 ? lc.parent : lane);
  if (lc.parent < 0) pp = v3<Real>(0, 0, 0);
  const V3<Real> r = pj - pp, sv = p - pj;
  V3<Real> ao = cross(alp, r) + cross(omp, cross(omp, r));
  if (rev) ao = ao + cross(al, sv) + cross(om, cross(om, sv));
  else ao = ao + cross(alp, sv) + cross(omp, cross(omp, sv)) + cross(omp, a * qd) * Real(2);
#pragma unroll
  for (int k = 0; k < SP_ROUNDS; k++) {
    if (k < nr) {
      const int hop = lc.anc[k];
      const V3<Real> t = shfl3(ao, hop >= 0 ? hop : lane);
      if (hop >= 0) ao = ao + t;
    }
  }
  if (!live) return;
  Real* L = S.link + lane * SP_LINKF;
  for (int k = 0; k < 9; k++) L[LK_R + k] = R[k];
  st3(L + LK_P, p); st3(L + LK_JO, pj); st3(L + LK_A, a); st3(L + LK_C, c);
  // wrench and composite seeds about the joint origin
  const Real m = lc.mass;
  const V3<Real> dj = c - pj;
  V3<Real> f = v3<Real>(0, 0, 0), nrm = f;
  Real Iw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (m > Real(0)) {
    Real RI[9];
    mulRR(R, G + LC_INERTIA, RI);
    for (int x = 0; x < 3; x++)
      for (int y = 0; y < 3; y++) Iw[3 * x + y] = RI[3 * x] * R[3 * y] + RI[3 * x + 1] * R[3 * y + 1] + RI[3 * x + 2] * R[3 * y + 2];
    const V3<Real> dc = c - p;
    const V3<Real> ac = ao + cross(al, dc) + cross(om, cross(om, dc));
    f = (ac - ld3(Md.g)) * m;
    nrm = mulR(Iw, al) + cross(om, mulR(Iw, om));
  }
  V3<Real> nj = nrm + cross(dj, f);
  if (EXTRAS && Md.task == 9) {
    // Snake fluid model (snake_7link.py:37-47): every body is pushed by -k (v_com . n) n at its frame origin, n = its z axis.
    // The link-origin velocity is one more path sum of per-link terms.
    V3<Real> vo = cross(omp, r) + (rev ? cross(om, sv) : cross(omp, sv) + a * qd);
#pragma unroll
    for (int k = 0; k < SP_ROUNDS; k++) {
      if (k < nr) {
        const int hop = lc.anc[k];
        const V3<Real> t = shfl3(vo, hop >= 0 ? hop : lane);
        if (hop >= 0) vo = vo + t;
      }
    }
    if (lc.is_body) {
      const V3<Real> vc = vo + cross(om, c - p), nd = v3<Real>(R[2], R[5], R[8]);
      const V3<Real> fe = nd * (-Md.aux_real[3] * dot(vc, nd));
      f = f - fe;
      nj = nj - cross(p - pj, fe);
    }
  }
  if (EXTRAS && Md.ext_force != nullptr && lane == Md.ext_link) {
    // bodynode.add_ext_force(F) before every world step (dart_env.py:170-172): a world-frame force at the body frame
    // origin enters the link's wrench with the opposite sign of its inertial force
    const V3<Real> fe = ld3(Md.ext_force + env * 3);
    f = f - fe;
    if (!Md.ext_at_joint_origin) nj = nj - cross(p - pj, fe);
  }
  st3(L + LK_F, f);
  st3(L + LK_N, nj);
  L[LK_MC] = m;
  st3(L + LK_H, dj * m);
  const Real d2 = dot(dj, dj);
  L[LK_IC + 0] = Iw[0] + m * (d2 - dj.x * dj.x);
  L[LK_IC + 1] = Iw[1] - m * dj.x * dj.y;
  L[LK_IC + 2] = Iw[2] - m * dj.x * dj.z;
  L[LK_IC + 3] = Iw[4] + m * (d2 - dj.y * dj.y);
  L[LK_IC + 4] = Iw[5] - m * dj.y * dj.z;
  L[LK_IC + 5] = Iw[8] + m * (d2 - dj.z * dj.z);
}

// floating-base translation: root-chain prismatic joints have fixed world axes (their ancestors never rotate)
template <class Real>
__device__ __forceinline__ void sp_root_offset(const SpatialModel<Real>& Md, SpLds<Real>& S) {
  V3<Real> roff = v3<Real>(0, 0, 0);
  for (int k = 0; k < Md.n_root_trans; k++) {
    const int i = Md.root_trans_link[k];
    // axis in world = (product of the constant pre/post rotations up to here) * axis; stored by the host
    roff = roff + ld3(Md.root_axis_world[i]) * S.q[Md.dof[i]];
  }
  st3(S.misc, roff);
}

// parent-centric backward step for group leader i (all child groups are complete): gather their wrenches and composite
// bodies (lc.children holds the child groups' leaders)
template <class Real>
__device__ __forceinline__ void sp_gather_children(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int i) {
  Real* Lp = S.link + i * SP_LINKF;
  V3<Real> F = ld3(Lp + LK_F), N = ld3(Lp + LK_N), H = ld3(Lp + LK_H);
  Real mcp = Lp[LK_MC];
  Real I0 = Lp[LK_IC + 0], I1 = Lp[LK_IC + 1], I2 = Lp[LK_IC + 2], I3 = Lp[LK_IC + 3], I4 = Lp[LK_IC + 4], I5 = Lp[LK_IC + 5];
  const V3<Real> jop = ld3(Lp + LK_JO);
  for (int ci = 0; ci < lc.nchild; ci++) {
    const Real* L = S.link + (int)((lc.children >> (8 * ci)) & 0xffull) * SP_LINKF;
    const V3<Real> o = ld3(L + LK_JO) - jop, Fc = ld3(L + LK_F);
    F = F + Fc;
    N = N + ld3(L + LK_N) + cross(o, Fc);
    const Real mc = L[LK_MC];
    const V3<Real> h = ld3(L + LK_H);
    const Real diag = Real(2) * dot(o, h) + mc * dot(o, o);
    I0 += L[LK_IC + 0] + diag - Real(2) * h.x * o.x - mc * o.x * o.x;
    I1 += L[LK_IC + 1] - (h.x * o.y + o.x * h.y) - mc * o.x * o.y;
    I2 += L[LK_IC + 2] - (h.x * o.z + o.x * h.z) - mc * o.x * o.z;
    I3 += L[LK_IC + 3] + diag - Real(2) * h.y * o.y - mc * o.y * o.y;
    I4 += L[LK_IC + 4] - (h.y * o.z + o.y * h.z) - mc * o.y * o.z;
    I5 += L[LK_IC + 5] + diag - Real(2) * h.z * o.z - mc * o.z * o.z;
    H = H + h + o * mc;
    mcp += mc;
  }
  st3(Lp + LK_F, F); st3(Lp + LK_N, N); st3(Lp + LK_H, H);
  Lp[LK_MC] = mcp;
  Lp[LK_IC + 0] = I0; Lp[LK_IC + 1] = I1; Lp[LK_IC + 2] = I2; Lp[LK_IC + 3] = I3; Lp[LK_IC + 4] = I4; Lp[LK_IC + 5] = I5;
}
// every link of a group takes the leader's composite (same joint origin, massless carriers), then emits its rhs entry
template <class Real, bool EXTRAS = false>
__device__ __forceinline__ void sp_link_rhs(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int i) {
  Real* L = S.link + i * SP_LINKF;
  if (lc.group_leader != i) {
    const Real* G = S.link + lc.group_leader * SP_LINKF;
    for (int k = LK_F; k < SP_LINKF; k++) L[k] = G[k];
  }
  const int d = lc.dof;
  if (d >= 0) {
    const V3<Real> a = ld3(L + LK_A);
    const Real Cb = (lc.jtype == 2) ? dot(a, ld3(L + LK_N)) : dot(a, ld3(L + LK_F));
    if (EXTRAS && Md.task == 12) {   // SPD: S.tau holds the target pose; the torque is added once M and c are known
      S.b[d] = Cb;
      S.rhs[d] = -Cb - lc.damp * S.dq[d] - lc.stiff * (S.q[d] + Md.dt * S.dq[d] - lc.rest);
    } else {
      S.rhs[d] = S.tau[d] - Cb - lc.damp * S.dq[d] - lc.stiff * (S.q[d] + Md.dt * S.dq[d] - lc.rest);
    }
  }
}

// row `d` of the mass matrix (lower part): one lane per dof walks its ancestor chain
template <class Real>
__device__ __forceinline__ void sp_mass_row(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int d) {
  const int i = lc.d_link;
  const Real* L = S.link + i * SP_LINKF;
  const V3<Real> a = ld3(L + LK_A), h = ld3(L + LK_H), jo = ld3(L + LK_JO);
  V3<Real> Lm, K;
  if (topo_jtype(S.topo[i]) == 2) {
    Lm = cross(a, h);
    const Real* I = L + LK_IC;
    K = v3<Real>(I[0] * a.x + I[1] * a.y + I[2] * a.z, I[1] * a.x + I[3] * a.y + I[4] * a.z, I[2] * a.x + I[4] * a.y + I[5] * a.z);
  } else {
    Lm = a * L[LK_MC];
    K = cross(h, a);
  }
  for (int k = 0; k < d; k++) S.H[TI(d, k)] = Real(0);
  for (int j = i; j >= 0;) {
    const int w = S.topo[j];
    const int dj = topo_dof(w), jcur = j;
    j = topo_parent(w);
    if (dj < 0) continue;
    const Real* Lj = S.link + jcur * SP_LINKF;
    const V3<Real> aj = ld3(Lj + LK_A);
    Real v;
    if (topo_jtype(w) == 2) v = dot(aj, K + cross(jo - ld3(Lj + LK_JO), Lm));
    else v = dot(aj, Lm);
    if (dj == d) v += lc.d_diag;
    S.H[TI(d, dj)] = v;   // dj <= d because parents come first
  }
}

// ------------------------------------------------------------------ wave-parallel dense kernels (row-owner scheme)
template <class Real> __device__ __forceinline__ Real readlane_(Real x, int l);
template <> __device__ __forceinline__ float readlane_<float>(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
template <> __device__ __forceinline__ double readlane_<double>(double x, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Cholesky of the packed lower triangle M (n <= 32, stored padded to sp_npad(n) with identity rows), run as a
// systolic array over the wave: lane r holds row r in REGISTERS, both loops are fully unrolled, and the finished
// column entry L_kj travels from lane k to everybody through v_readlane (an SGPR operand of the FMA) -- no LDS traffic
// and no barrier inside the factorisation.  Per column j: d_j = readlane(row[j], j); L_rj = row[j] / sqrt(d_j);
// row[k] -= L_rj L_kj for k > j.  Updates beyond a lane's diagonal are garbage that nothing reads (kept finite by the
// identity padding).  The factor is written back to LDS once at the end, with sinv[j] = 1 / L_jj.
template <class Real, int NP>
__device__ __forceinline__ void sp_cholesky_t(Real* M, Real* sinv, int n, int lane) {
  const int r = lane < NP ? lane : 0;   // spare lanes shadow lane 0 (convergent code, results discarded)
  const int rb = TL(r, 0);
  Real row[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) row[k] = (k <= r) ? M[rb + k] : Real(0);
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const Real dj = readlane_<Real>(row[j], j);
    const Real sj = rsqrt_<Real>(dj);
    const Real lrj = (r >= j) ? row[j] * sj : Real(0);   // lanes above the diagonal contribute nothing
    row[j] = lrj;
    if (lane == j) sinv[j] = sj;
#pragma unroll
    for (int k = j + 1; k < NP; k++) row[k] -= lrj * readlane_<Real>(lrj, k);
  }
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NP; k++) if (k <= lane) M[rb + k] = row[k];
  }
  __syncthreads();
}
// one straight-line variant per padded size (only the one a model uses ever enters the instruction cache)
template <class Real>
__device__ __forceinline__ void sp_cholesky(Real* M, Real* sinv, int n, int lane) {
  const int np = sp_npad(n);
  if (np <= 8) sp_cholesky_t<Real, 8>(M, sinv, n, lane);
  else if (np <= 16) sp_cholesky_t<Real, 16>(M, sinv, n, lane);
  else if (np <= 24) sp_cholesky_t<Real, 24>(M, sinv, n, lane);
  else sp_cholesky_t<Real, 32>(M, sinv, n, lane);
}
// x <- L^-T x (backward) for one vector in LDS, column-oriented, lanes own entries
template <class Real>
__device__ __forceinline__ void sp_chol_backsolve(const Real* Lf, const Real* sinv, int n, Real* x, int lane) {
  for (int j = n - 1; j >= 0; j--) {
    __syncthreads();
    const Real xj = x[j] * sinv[j];
    if (lane == j) x[j] = xj;
    if (lane < j) x[lane] -= Lf[TL(j, lane)] * xj;
  }
  __syncthreads();
}

// x <- L^-1 x (forward), column-oriented like the back-substitution
template <class Real>
__device__ __forceinline__ void sp_chol_fwdsolve(const Real* Lf, const Real* sinv, int n, Real* x, int lane) {
  for (int j = 0; j < n; j++) {
    __syncthreads();
    const Real xj = x[j] * sinv[j];
    if (lane == j) x[j] = xj;
    if (lane > j && lane < n) x[lane] -= Lf[TL(lane, j)] * xj;
  }
  __syncthreads();
}

// Stable-PD torque of DartWalker3dSPD-v1 (walker3d_spd.py:40-55), evaluated before every world step once M (in S.H with
// the integrator's diagonal terms), the bias forces c (S.b) and the previous step's constraint forces (S.cf) are known:
//   qdd = (M + Kd dt_env)^-1 (-c + p + d + cf),  tau = p + d - Kd qdd dt_env,  root dofs zeroed, |tau| <= limit.
// S.tau holds the target pose; the torque goes straight into the right-hand side.  Workspace: S.A (factor), S.r (1/L_jj),
// S.lo (the solve) -- all idle until the constraint phase.
template <class Real>
__device__ __forceinline__ void sp_spd_torque(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane) {
  const int n = Md.n, np = sp_npad(n);
  Real pd = Real(0), kd = Real(0);
  if (lane < np) {
    for (int k = 0; k <= lane; k++) S.A[TL(lane, k)] = S.H[TL(lane, k)];
    if (lane < n) {
      kd = Md.spd_kd[lane];
      S.A[TL(lane, lane)] += kd * Md.envdt - lc.d_diag;
      const Real p = -Md.spd_kp[lane] * (S.q[lane] + S.dq[lane] * Md.envdt - S.tau[lane]);
      const Real d = -kd * S.dq[lane];
      pd = p + d;
      S.lo[lane] = -S.b[lane] + p + d + S.cf[lane];
    }
  }
  __syncthreads();
  sp_cholesky<Real>(S.A, S.r, n, lane);
  sp_chol_fwdsolve<Real>(S.A, S.r, n, S.lo, lane);
  sp_chol_backsolve<Real>(S.A, S.r, n, S.lo, lane);
  if (lane < n) {
    Real tq = pd - kd * S.lo[lane] * Md.envdt;
    const int k = lane - Md.act_dof0;
    if (k < 0 || k >= Md.act_dim) tq = Real(0);
    else if (fabs(tq) > Md.act_scale[k]) tq = (tq > Real(0) ? Real(1) : Real(-1)) * Md.act_scale[k];
    S.rhs[lane] += tq;
  }
  __syncthreads();
}

// Boxed LCP by block principal pivoting, one wavefront per problem (rows = lanes).  F/U are wave-uniform bit masks.
template <class Real>
__device__ __forceinline__ void sp_blcp(SpLds<Real>& S, int m, uint64_t pinmask, uint64_t& F, uint64_t& U, int max_iter,
                                       int pgs_sweeps, unsigned long long* stats, int lane, const bool ZERO_BOUNDS) {
  if (lane < m) S.x0[lane] = S.x[lane];   // solution of the previous stage (zeros before the first): PGS fallback start
  Real bmax = Real(0);
  for (int i = 0; i < m; i++) bmax = fmax(bmax, fabs(S.b[i]));
  const Real tol = tol_<Real>() * (Real(1) + bmax);
  int best = m + 1, patience = 3;
  const bool row = lane < m;
  bool converged = false;
  int it = 0;
  const int rbase = TL(lane, 0);
  for (; it < max_iter; ++it) {
    const bool fi = row && ((F >> lane) & 1ull), ui = row && ((U >> lane) & 1ull);
    __syncthreads();
    if (row) S.x[lane] = fi ? Real(0) : (ui ? S.hi[lane] : S.lo[lane]);   // xb
    __syncthreads();
    // rhs and masked copy of A
    if (row) {
      Real t = S.b[lane];
      if (!ZERO_BOUNDS) {
        for (int j = 0; j <= lane; j++) t -= S.A[rbase + j] * S.x[j];
        int jl = TL(lane + 1, lane);
        for (int j = lane + 1; j < m; j++) { t -= S.A[jl] * S.x[j]; jl += j + 1; }
      }
      S.r[lane] = fi ? t : S.x[lane];
      for (int j = 0; j < lane; j++) {
        const bool fj = (F >> j) & 1ull;
        S.Lw[rbase + j] = (fi && fj) ? S.A[rbase + j] : Real(0);
      }
      S.Lw[rbase + lane] = fi ? S.A[rbase + lane] : Real(1);
    }
    __syncthreads();
    // LDL^T restricted to the free columns (non-free columns are identity: nothing to eliminate); column j is
    // read-only while it is eliminated (unscaled entries u_kj = l_kj d_j), so one barrier per column
    {
      uint64_t Fr = F;
      while (Fr) {
        const int j = __builtin_ctzll(Fr);
        Fr &= Fr - 1;
        __syncthreads();
        if (row && lane > j && fi) {
          const Real lij = S.Lw[rbase + j] * rcp_<Real>(S.Lw[TL(j, j)]);
          int kj = TL(j + 1, j);
          for (int k = j + 1; k <= lane; k++) { S.Lw[rbase + k] -= lij * S.Lw[kj]; kj += k + 1; }
        }
      }
    }
    __syncthreads();
    // 1/d_j, then solve L D L^T x = r over the free rows (column oriented; l_ij = u_ij / d_j)
    const Real invd_own = row ? rcp_<Real>(S.Lw[rbase + lane]) : Real(1);
    {
      uint64_t Fr = F;
      while (Fr) {
        const int j = __builtin_ctzll(Fr);
        Fr &= Fr - 1;
        __syncthreads();
        const Real xj = S.r[j] * rcp_<Real>(S.Lw[TL(j, j)]);
        if (row && lane > j && fi) S.r[lane] -= S.Lw[rbase + j] * xj;
      }
    }
    __syncthreads();
    if (fi) S.r[lane] *= invd_own;
    {
      uint64_t Fr = F;
      while (Fr) {
        const int j = 63 - __builtin_clzll(Fr);
        Fr &= ~(1ull << j);
        __syncthreads();
        const Real xj = S.r[j];
        if (row && lane < j && fi) S.r[lane] -= S.Lw[TL(j, lane)] * invd_own * xj;
      }
    }
    __syncthreads();
    // feasibility of every row
    bool inf = false, gt = false;
    if (row) {
      Real w = -S.b[lane];
      for (int j = 0; j <= lane; j++) w += S.A[rbase + j] * S.r[j];
      int jl = TL(lane + 1, lane);
      for (int j = lane + 1; j < m; j++) { w += S.A[jl] * S.r[j]; jl += j + 1; }
      const Real ri = S.r[lane], lo = S.lo[lane], hi = S.hi[lane];
      const bool pinned = (pinmask >> lane) & 1ull;
      const bool over = ri > hi + tol * (Real(1) + fabs(hi)), under = ri < lo - tol * (Real(1) + fabs(lo));
      const bool wbad = ui ? (w > tol) : (w < -tol);
      inf = fi ? (over || under) : (wbad && !pinned);
      gt = ri > hi;
    }
    const uint64_t B = __ballot(inf), GT = __ballot(gt);
    if (B == 0ull) { converged = true; break; }
    const int ninf = __popcll(B);
    const bool improved = ninf < best;
    const bool single = !improved && patience == 0;
    best = improved ? ninf : best;
    patience = improved ? 3 : (patience > 0 ? patience - 1 : 0);
    const uint64_t Bs = single ? (1ull << (63 - __clzll((long long)B))) : B;
    const uint64_t toBound = Bs & F, toFree = Bs & ~F;
    F = (F & ~toBound) | toFree;
    U = (U & ~(toFree | toBound)) | (toBound & GT);
  }
  __syncthreads();
  if (stats && lane == 0) { atomicAdd(&stats[it < 31 ? it : 31], 1ull); atomicAdd(&stats[33], 1ull); }
  if (converged) {
    if (row) S.x[lane] = fmin(fmax(S.r[lane], S.lo[lane]), S.hi[lane]);
  } else {
    // The pivoting loop did not settle (degenerate, redundant-contact LCP): projected Gauss-Seidel from the previous
    // stage's impulses -- always in the box, monotone in the QP energy.  Row dot products are spread over the lanes.
    if (stats && lane == 0) atomicAdd(&stats[32], 1ull);
    if (row) S.x[lane] = fmin(fmax(S.x0[lane], S.lo[lane]), S.hi[lane]);
    __syncthreads();
    for (int sw = 0; sw < pgs_sweeps; ++sw)
      for (int i = 0; i < m; i++) {
        if ((pinmask >> i) & 1ull) continue;
        Real part = row ? S.A[TI(i, lane)] * S.x[lane] : Real(0);
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) {
          const Real xn = S.x[i] + (S.b[i] - part) / S.A[TI(i, i)];
          S.x[i] = fmin(fmax(xn, S.lo[i]), S.hi[i]);
        }
        __syncthreads();
      }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ DART FreeJoint root (dog.skel)
// Public coordinates: q[0:3] rotation vector, q[3:6] translation, dq[0:6] = twist of the child joint frame in that frame;
// DART integrates the pose as Q <- Q * [exp(w dt), v dt].  The dynamics run on the internal chain (translation x y z,
// rotations about x y z in a chart centred on the current orientation); S.root keeps R, p and the twist, the internal
// coordinates are re-derived from them before every world
// step and the new velocities are mapped back with the exact instantaneous Jacobian -- only the parametrisation
// differs from DART, not the integrator.
template <class Real>
__device__ __forceinline__ void sp_so3_exp(V3<Real> r, Real* R) {
  const Real th2 = dot(r, r), th = sqrt(th2);
  Real a, b;
  if (th < Real(1e-4)) { a = Real(1) - th2 / Real(6); b = Real(0.5) - th2 / Real(24); }
  else { Real sn, cs; sincos_<Real>(th, sn, cs); a = sn / th; b = (Real(1) - cs) / th2; }
  const Real K[9] = {0, -r.z, r.y, r.z, 0, -r.x, -r.y, r.x, 0};
  Real K2[9];
  mulRR(K, K, K2);
  for (int k = 0; k < 9; k++) R[k] = ((k % 4 == 0) ? Real(1) : Real(0)) + a * K[k] + b * K2[k];
}
// log map through the unit quaternion (largest-pivot extraction, then 2 atan2(|v|, w)): well conditioned at every angle,
// including rotations by pi where acos(trace) and R - R^T lose half of the digits -- the pose makes this round trip once per
// env step because the public state is DART's rotation vector.
template <class Real>
__device__ __forceinline__ V3<Real> sp_so3_log(const Real* R) {
  const Real tr = R[0] + R[4] + R[8];
  Real w, x, y, z;
  if (tr > Real(0)) {
    const Real s = sqrt(tr + Real(1)) * Real(2);
    w = s * Real(0.25); x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const Real s = sqrt(Real(1) + R[0] - R[4] - R[8]) * Real(2);
    w = (R[7] - R[5]) / s; x = s * Real(0.25); y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const Real s = sqrt(Real(1) + R[4] - R[0] - R[8]) * Real(2);
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = s * Real(0.25); z = (R[5] + R[7]) / s;
  } else {
    const Real s = sqrt(Real(1) + R[8] - R[0] - R[4]) * Real(2);
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = s * Real(0.25);
  }
  if (w < Real(0)) { w = -w; x = -x; y = -y; z = -z; }   // angle in [0, pi]
  const Real nv = sqrt(x * x + y * y + z * z);
  const Real k = nv < Real(1e-6) ? Real(2) / w : Real(2) * atan2(nv, w) / nv;
  return v3<Real>(x * k, y * k, z * k);
}
// S.root -> internal coordinates of the six root links (lane 0).  The rotation chart is centred on the current orientation:
// joint rotation = Rx(a) Ry(b) Rz(c) R0 with R0 = S.root[0:9] and a = b = c = 0, so the rates are the angular velocity in
// the joint's parent frame (E = I) and the chart has no singularity however far the body turns; R0 enters the forward pass
// as part of the last root link's joint-to-child transform (sp_forward / sp_kinematics).
template <class Real>
__device__ __forceinline__ void sp_free_root_to_internal(SpLds<Real>& S) {
  const Real* R = S.root;
  const V3<Real> ww = mulR(R, ld3(S.root + 12)), pd = mulR(R, ld3(S.root + 15));
  S.q[0] = Real(0); S.q[1] = Real(0); S.q[2] = Real(0);
  S.q[3] = S.root[9]; S.q[4] = S.root[10]; S.q[5] = S.root[11];
  S.dq[0] = ww.x; S.dq[1] = ww.y; S.dq[2] = ww.z;
  S.dq[3] = pd.x; S.dq[4] = pd.y; S.dq[5] = pd.z;
}
// after the velocity update: new internal rates (at the old pose) -> new body twist; DART's pose update
template <class Real>
__device__ __forceinline__ void sp_free_root_advance(SpLds<Real>& S, Real dt) {
  Real* R = S.root;
  const V3<Real> ww = v3<Real>(S.dq[0], S.dq[1], S.dq[2]);
  const V3<Real> pd = v3<Real>(S.dq[3], S.dq[4], S.dq[5]);
  const V3<Real> wb = v3<Real>(R[0] * ww.x + R[3] * ww.y + R[6] * ww.z, R[1] * ww.x + R[4] * ww.y + R[7] * ww.z, R[2] * ww.x + R[5] * ww.y + R[8] * ww.z);
  const V3<Real> vb = v3<Real>(R[0] * pd.x + R[3] * pd.y + R[6] * pd.z, R[1] * pd.x + R[4] * pd.y + R[7] * pd.z, R[2] * pd.x + R[5] * pd.y + R[8] * pd.z);
  st3(S.root + 12, wb); st3(S.root + 15, vb);
  S.root[9] += dt * pd.x; S.root[10] += dt * pd.y; S.root[11] += dt * pd.z;   // p += R v_b dt = pdot dt
  Real dR[9], Rn[9];
  sp_so3_exp<Real>(wb * dt, dR);
  mulRR(R, dR, Rn);
  for (int k = 0; k < 9; k++) R[k] = Rn[k];
}
// DART integrates the BODY-FRAME twist: twist' = twist + dt twist_acc.  With dq_int = T(q) twist the accelerations map as
// qdd_int = T twist_acc + Tdot twist, so the internal velocity that corresponds to DART's unconstrained one is
// dq_int + dt qdd_int - dt Tdot twist:  rotation (E rates = R w_b): -Tdot twist = E^-1 Edot rates = (rb rc, -ra rc, ra rb) at
// the chart centre;  translation (pdot = R v_b): -Tdot twist = -(w x pdot).  Applied to S.dq[0:6] after the bias forces have
// been computed from the true velocities; every later use of S.dq in the world step is at velocity level.
template <class Real>
__device__ __forceinline__ void sp_free_root_velocity_correction(SpLds<Real>& S, Real dt) {
  const Real ra = S.dq[0], rb = S.dq[1], rc = S.dq[2];
  const V3<Real> wxp = cross(v3<Real>(ra, rb, rc), v3<Real>(S.dq[3], S.dq[4], S.dq[5]));
  S.dq[0] += dt * (rb * rc); S.dq[1] -= dt * (ra * rc); S.dq[2] += dt * (ra * rb);
  S.dq[3] -= dt * wxp.x; S.dq[4] -= dt * wxp.y; S.dq[5] -= dt * wxp.z;
}
// public root coordinates (already in S.q / S.dq[0:6]) -> S.root
template <class Real>
__device__ __forceinline__ void sp_free_root_load(SpLds<Real>& S) {
  sp_so3_exp<Real>(v3<Real>(S.q[0], S.q[1], S.q[2]), S.root);
  for (int k = 0; k < 3; k++) { S.root[9 + k] = S.q[3 + k]; S.root[12 + k] = S.dq[k]; S.root[15 + k] = S.dq[3 + k]; }
}
// S.root -> public coordinates in S.q / S.dq[0:6]
template <class Real>
__device__ __forceinline__ void sp_free_root_store(SpLds<Real>& S) {
  const V3<Real> r = sp_so3_log<Real>(S.root);
  S.q[0] = r.x; S.q[1] = r.y; S.q[2] = r.z;
  for (int k = 0; k < 3; k++) { S.q[3 + k] = S.root[9 + k]; S.dq[k] = S.root[12 + k]; S.dq[3 + k] = S.root[15 + k]; }
}

// ------------------------------------------------------------------ box-box contacts between two links
// ODE's dBoxBox (the routine behind DART's ODE detector for two boxes): separating-axis test over the 15 axes with the
// 1.05 preference for face axes; edge-edge -> one point midway between the closest points of the two edges; face case
// -> the incident face of the other box is clipped against the reference face's rectangle and the clipped vertices
// below the reference face are the contacts.  Returns the number of points (<= 8) written as (x, y, z, depth) to `out`
// (40 Reals of LDS workspace); `normal` points from the first box to the second.
template <class Real>
__device__ __forceinline__ int sp_clip_rect_quad(const Real* h, Real* p, Real* ret, Real* buffer) {
  int nq = 4, nr = 0;
  Real* q = p;
  Real* r = ret;
  for (int dir = 0; dir <= 1; dir++) {
    for (int sign = -1; sign <= 1; sign += 2) {
      Real* pq = q;
      Real* pr = r;
      nr = 0;
      bool full = false;
      for (int i = nq; i > 0 && !full; i--) {
        const bool in0 = Real(sign) * pq[dir] < h[dir];
        if (in0) {
          pr[0] = pq[0]; pr[1] = pq[1]; pr += 2; nr++;
          if (nr & 8) { full = true; break; }
        }
        Real* nextq = (i > 1) ? pq + 2 : q;
        const bool in1 = Real(sign) * nextq[dir] < h[dir];
        if (in0 != in1) {
          pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (Real(sign) * h[dir] - pq[dir]);
          pr[dir] = Real(sign) * h[dir];
          pr += 2; nr++;
          if (nr & 8) { full = true; break; }
        }
        pq += 2;
      }
      q = r;
      if (full) { dir = 2; break; }
      r = (q == ret) ? buffer : ret;
      nq = nr;
    }
  }
  if (q != ret) for (int i = 0; i < 2 * nr; i++) ret[i] = q[i];
  return nr;
}

template <class Real>
__device__ __forceinline__ int sp_box_box(const SpatialModel<Real>& Md, SpLds<Real>& S, int sa, int sb, Real* out, V3<Real>& normal,
                                          int& la, int& lb) {
  const Real eps = sizeof(Real) == 4 ? Real(1.1920929e-7) : Real(2.220446049250313e-16);
  la = Md.sh_link[sa]; lb = Md.sh_link[sb];
  V3<Real> u[3], v[3], p1, p2, A, B;
  {
    const Real* La = S.link + la * SP_LINKF;
    const Real* Lb = S.link + lb * SP_LINKF;
    Real Ta[9], Tb[9], ra[9], rb[9];
    for (int k = 0; k < 9; k++) { ra[k] = Md.sh_R[sa][k]; rb[k] = Md.sh_R[sb][k]; }
    mulRR(La + LK_R, ra, Ta);
    mulRR(Lb + LK_R, rb, Tb);
    p1 = ld3(La + LK_P) + mulR(La + LK_R, ld3(Md.sh_p[sa]));
    p2 = ld3(Lb + LK_P) + mulR(Lb + LK_R, ld3(Md.sh_p[sb]));
    for (int j = 0; j < 3; j++) { u[j] = v3<Real>(Ta[j], Ta[3 + j], Ta[6 + j]); v[j] = v3<Real>(Tb[j], Tb[3 + j], Tb[6 + j]); }
    A = ld3(Md.sh_size[sa]) * Real(0.5); B = ld3(Md.sh_size[sb]) * Real(0.5);
  }
  const V3<Real> p = p2 - p1;
  const Real pp[3] = {dot(u[0], p), dot(u[1], p), dot(u[2], p)};
  const Real Av[3] = {A.x, A.y, A.z}, Bv[3] = {B.x, B.y, B.z};
  Real R[3][3], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = dot(u[i], v[j]); Q[i][j] = fabs(R[i][j]); }
  Real s = -inf_<Real>();
  V3<Real> nC = v3<Real>(0, 0, 0);
  int code = 0;
  bool invert = false, sep = false;
  // face axes of box 1, then of box 2
  for (int i = 0; i < 3; i++) {
    const Real e1 = pp[i], s2 = fabs(e1) - (Av[i] + Bv[0] * Q[i][0] + Bv[1] * Q[i][1] + Bv[2] * Q[i][2]);
    sep = sep || (s2 > Real(0));
    if (s2 > s) { s = s2; invert = e1 < Real(0); code = i + 1; }
  }
  for (int j = 0; j < 3; j++) {
    const Real e1 = dot(v[j], p), s2 = fabs(e1) - (Av[0] * Q[0][j] + Av[1] * Q[1][j] + Av[2] * Q[2][j] + Bv[j]);
    sep = sep || (s2 > Real(0));
    if (s2 > s) { s = s2; invert = e1 < Real(0); code = j + 4; }
  }
  if (sep) return 0;
  // edge axes u_i x v_j (i = 0: (0,-R2j,R1j), i = 1: (R2j,0,-R0j), i = 2: (-R1j,R0j,0)), Q padded by 1e-5 like ODE
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] += Real(1.0e-5);
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const Real e1 = pp[i2] * R[i1][j] - pp[i1] * R[i2][j];
      Real s2 = fabs(e1) - (Av[i1] * Q[i2][j] + Av[i2] * Q[i1][j] + Bv[j1] * Q[i][j2] + Bv[j2] * Q[i][j1]);
      sep = sep || (s2 > eps);
      Real nv[3] = {0, 0, 0};
      nv[i1] = -R[i2][j]; nv[i2] = R[i1][j];
      const Real l = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
      if (!sep && l > eps) {
        s2 /= l;
        if (s2 * Real(1.05) > s) { s = s2; nC = v3<Real>(nv[0] / l, nv[1] / l, nv[2] / l); invert = e1 < Real(0); code = 7 + 3 * i + j; }
      }
    }
  }
  if (sep || code == 0) return 0;
  if (code <= 3) normal = u[code - 1];
  else if (code <= 6) normal = v[code - 4];
  else normal = u[0] * nC.x + u[1] * nC.y + u[2] * nC.z;
  if (invert) normal = normal * Real(-1);
  const Real depth = -s;
  if (code > 6) {   // edge-edge
    V3<Real> pa = p1, pb = p2;
    for (int j = 0; j < 3; j++) {
      pa = pa + u[j] * ((dot(normal, u[j]) > Real(0) ? Real(1) : Real(-1)) * Av[j]);
      pb = pb + v[j] * ((dot(normal, v[j]) > Real(0) ? Real(-1) : Real(1)) * Bv[j]);
    }
    const int ia = (code - 7) / 3, ib = (code - 7) % 3;
    const V3<Real> ua = ia == 0 ? u[0] : (ia == 1 ? u[1] : u[2]), ub = ib == 0 ? v[0] : (ib == 1 ? v[1] : v[2]);
    const V3<Real> d3 = pb - pa;
    const Real uaub = dot(ua, ub), q1 = dot(ua, d3), q2 = -dot(ub, d3);
    Real d = Real(1) - uaub * uaub, alpha = Real(0), beta = Real(0);
    if (d > Real(1e-4)) { d = Real(1) / d; alpha = (q1 + uaub * q2) * d; beta = (uaub * q1 + q2) * d; }
    const V3<Real> mid = ((pa + ua * alpha) + (pb + ub * beta)) * Real(0.5);
    out[0] = mid.x; out[1] = mid.y; out[2] = mid.z; out[3] = depth;
    return 1;
  }
  // face case: the reference face belongs to box a (box 1 for codes 1..3, box 2 otherwise)
  const bool first = code <= 3;
  V3<Real> Ra[3], Rb[3];
  for (int j = 0; j < 3; j++) { Ra[j] = first ? u[j] : v[j]; Rb[j] = first ? v[j] : u[j]; }
  const V3<Real> pa = first ? p1 : p2, pb = first ? p2 : p1;
  const Real* Sa = first ? Av : Bv;
  const Real* Sb = first ? Bv : Av;
  const V3<Real> normal2 = first ? normal : normal * Real(-1);
  const Real nr[3] = {dot(Rb[0], normal2), dot(Rb[1], normal2), dot(Rb[2], normal2)};
  const Real anr[3] = {fabs(nr[0]), fabs(nr[1]), fabs(nr[2])};
  int lanr, a1, a2;
  if (anr[1] > anr[0]) { if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  else { if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  auto pick = [](const V3<Real>* M3, int k) -> V3<Real> { return k == 0 ? M3[0] : (k == 1 ? M3[1] : M3[2]); };
  auto pickr = [](const Real* a3, int k) -> Real { return k == 0 ? a3[0] : (k == 1 ? a3[1] : a3[2]); };
  const V3<Real> Rbl = pick(Rb, lanr), Rb1 = pick(Rb, a1), Rb2 = pick(Rb, a2);
  const V3<Real> center = pb - pa + Rbl * ((pickr(nr, lanr) < Real(0) ? Real(1) : Real(-1)) * pickr(Sb, lanr));
  const int codeN = first ? code - 1 : code - 4;
  const int code1 = codeN == 0 ? 1 : 0, code2 = codeN == 2 ? 1 : 2;
  const V3<Real> Ra1 = pick(Ra, code1), Ra2 = pick(Ra, code2);
  const Real c1 = dot(center, Ra1), c2 = dot(center, Ra2);
  Real m11 = dot(Ra1, Rb1), m12 = dot(Ra1, Rb2), m21 = dot(Ra2, Rb1), m22 = dot(Ra2, Rb2);
  Real* quad = out + 32;
  Real* ret = out;
  Real* buffer = out + 16;
  {
    const Real k1 = m11 * pickr(Sb, a1), k2 = m21 * pickr(Sb, a1), k3 = m12 * pickr(Sb, a2), k4 = m22 * pickr(Sb, a2);
    quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4; quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
    quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4; quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
  }
  const Real rect[2] = {pickr(Sa, code1), pickr(Sa, code2)};
  const int nq = sp_clip_rect_quad<Real>(rect, quad, ret, buffer);
  if (nq < 1) return 0;
  Real rx[8], ry[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { rx[j] = j < nq ? ret[2 * j] : Real(0); ry[j] = j < nq ? ret[2 * j + 1] : Real(0); }
  const Real det1 = Real(1) / (m11 * m22 - m12 * m21);
  m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
  const Real SaN = pickr(Sa, codeN);
  int cnum = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (j < nq) {
      const Real k1 = m22 * (rx[j] - c1) - m12 * (ry[j] - c2), k2 = -m21 * (rx[j] - c1) + m11 * (ry[j] - c2);
      const V3<Real> pt = center + Rb1 * k1 + Rb2 * k2;
      const Real dep = SaN - dot(normal2, pt);
      if (dep >= Real(0)) {
        const V3<Real> pos = first ? pt + pa : pt + pa - normal * dep;
        out[4 * cnum + 0] = pos.x; out[4 * cnum + 1] = pos.y; out[4 * cnum + 2] = pos.z; out[4 * cnum + 3] = dep;
        cnum++;
      }
    }
  }
  return cnum;
}

// ------------------------------------------------------------------ one world step for the env owned by this wavefront
#define SP_TICK(ph)                                                                              \
  do {                                                                                            \
    if (Md.stats && lane == 0) {                                                                  \
      const unsigned long long t1_ = __builtin_readcyclecounter();                                \
      S.ticks[ph] += t1_ - t0_;                                                                   \
      t0_ = t1_;                                                                                  \
    }                                                                                             \
  } while (0)

// PAIRS: link-link contacts (box pairs, general contact normals); EXTRAS: snake fluid forces, external body force, Coulomb
// joint friction rows.  Models that need neither run the lean instantiation (HumanWalker: 8 % faster than the full one).
template <class Real, bool PAIRS, bool EXTRAS, bool REPORT = false>
__device__ __forceinline__ void sp_world_step(const SpatialModel<Real>& Md, const LinkConst<Real>& lc, SpLds<Real>& S, int lane,
                                              int* contact_flags, bool report = false) {
  const int n = Md.n, nl = Md.nl;
  unsigned long long t0_ = Md.stats ? __builtin_readcyclecounter() : 0ull;
  if (EXTRAS && Md.free_root) {
    if (lane == 0) sp_free_root_to_internal<Real>(S);
    __syncthreads();
  }
  // tree recursions level by level: links of equal depth are independent, one lane each
  if (lane == 0) sp_root_offset<Real>(Md, S);
  sp_forward<Real, EXTRAS>(lc, Md, S, lane, (int64_t)blockIdx.x);   // lane i owns link i
  __syncthreads();
  for (int lv = Md.n_group_levels - 1; lv >= 0; lv--) {
    if (lane < nl && lc.group_level == lv) sp_gather_children<Real>(lc, Md, S, lane);
    __syncthreads();
  }
  if (lane < nl) sp_link_rhs<Real, EXTRAS>(lc, Md, S, lane);
  __syncthreads();
  if (EXTRAS && Md.free_root) {
    if (lane == 0) sp_free_root_velocity_correction<Real>(S, Md.dt);
    __syncthreads();
  }
  SP_TICK(0);
  if (lane < n) sp_mass_row<Real>(lc, Md, S, lane);
  else if (lane < sp_npad(n)) { for (int k = 0; k < lane; k++) S.H[TL(lane, k)] = Real(0); S.H[TL(lane, lane)] = Real(1); }
  __syncthreads();
  if (EXTRAS && Md.task == 12) sp_spd_torque<Real>(lc, Md, S, lane);
  SP_TICK(1);
  sp_cholesky<Real>(S.H, S.sinv, n, lane);
  SP_TICK(2);

  // ---- contact points and active limits, in parallel: lane s tests collision shape s, lane d tests the limits of
  // dof d; ballots give every hit its slot (shape order, then vertex order -- the serial order of the oracle)
  const V3<Real> roff = ld3(S.misc);
  int ncp, m;
  {
    const bool has_shape = lane < Md.nshapes;
    const int s = has_shape ? lane : 0;
    const int slink = Md.sh_link[s], stype = Md.sh_type[s];
    Real sR[9];
    for (int k = 0; k < 9; k++) sR[k] = Md.sh_R[s][k];
    const V3<Real> sp = ld3(Md.sh_p[s]), ssz = ld3(Md.sh_size[s]);
    const Real* L = S.link + slink * SP_LINKF;
    Real Ts[9];
    mulRR(L + LK_R, sR, Ts);
    const V3<Real> pc = ld3(L + LK_P) + mulR(L + LK_R, sp);
    V3<Real> P[4];
    Real dep[4];
    bool hit[4] = {false, false, false, false};
    if (stype == 0) {   // capsule: lowest segment endpoint, ODE sphere-sphere contact position
      const Real rad = ssz.x, hl = Real(0.5) * ssz.y;
      const V3<Real> zc = v3<Real>(Ts[2], Ts[5], Ts[8]);
      const V3<Real> p1 = pc + zc * hl, p2 = pc - zc * hl;
      const V3<Real> pe = (p2.y < p1.y) ? p2 : p1;
      const Real d = pe.y + roff.y - Md.ground_y;
      hit[0] = has_shape && d <= rad;
      P[0] = v3<Real>(pe.x, pe.y - Real(0.5) * (rad + d), pe.z); dep[0] = rad - d;
      for (int v = 1; v < 4; v++) { P[v] = P[0]; dep[v] = Real(0); }
    } else {            // box: vertices of the face that looks down, the ones below the floor
      const V3<Real> c0 = v3<Real>(Ts[0], Ts[3], Ts[6]), c1 = v3<Real>(Ts[1], Ts[4], Ts[7]), c2 = v3<Real>(Ts[2], Ts[5], Ts[8]);
      int k = 0;
      Real bestv = fabs(c0.y);
      if (fabs(c1.y) > bestv) { bestv = fabs(c1.y); k = 1; }
      if (fabs(c2.y) > bestv) k = 2;
      const V3<Real> ek = k == 0 ? c0 : (k == 1 ? c1 : c2), e1 = k == 0 ? c1 : (k == 1 ? c2 : c0), e2 = k == 0 ? c2 : (k == 1 ? c0 : c1);
      const Real hk = Real(0.5) * (k == 0 ? ssz.x : (k == 1 ? ssz.y : ssz.z)), h1 = Real(0.5) * (k == 0 ? ssz.y : (k == 1 ? ssz.z : ssz.x)),
                 h2 = Real(0.5) * (k == 0 ? ssz.z : (k == 1 ? ssz.x : ssz.y));
      const Real sgn = ek.y > Real(0) ? Real(-1) : Real(1);
      const V3<Real> base = pc + ek * (sgn * hk);
      const Real sg1[4] = {1, -1, -1, 1}, sg2[4] = {1, 1, -1, -1};
      for (int v = 0; v < 4; v++) {
        P[v] = base + e1 * (sg1[v] * h1) + e2 * (sg2[v] * h2);
        dep[v] = Md.ground_y - (P[v].y + roff.y);
        hit[v] = has_shape && dep[v] >= Real(0);
      }
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t hm[4];
    int before = 0, total = 0;
    for (int v = 0; v < 4; v++) { hm[v] = __ballot(hit[v]); before += __popcll(hm[v] & lt); total += __popcll(hm[v]); }
    int idx = before;
    for (int v = 0; v < 4; v++) {
      if (hit[v]) {
        if (idx < Md.maxcp) {
          S.cpP[4 * idx + 0] = P[v].x; S.cpP[4 * idx + 1] = P[v].y; S.cpP[4 * idx + 2] = P[v].z; S.cpP[4 * idx + 3] = dep[v];
          S.cpN[3 * idx + 0] = Real(0); S.cpN[3 * idx + 1] = Real(1); S.cpN[3 * idx + 2] = Real(0);
          S.cplink[idx] = slink; S.cplinkB[idx] = -1;
        }
        idx++;
      }
    }
    ncp = total < Md.maxcp ? total : Md.maxcp;
    // foot-contact flags of the observation (human_walker.py:97-106): any contact on aux_link[2], aux_link[3]
    const bool anyhit = hit[0] || hit[1] || hit[2] || hit[3];
    const uint64_t f0 = __ballot(anyhit && slink == Md.aux_link[2]), f1 = __ballot(anyhit && slink == Md.aux_link[3]);
    if (lane == 0) { contact_flags[0] = f0 != 0ull; contact_flags[1] = f1 != 0ull; }
    // link-link contacts (walker3d.py:26): lane p tests shape pair p; the points follow the ground contacts, pair by pair
    if (PAIRS && Md.npairs > 0) {
      const bool has_pair = lane < Md.npairs;
      Real* scratch = S.A + lane * 40;          // A / Lw are idle in this phase: 40 Reals of clipping workspace per lane
      int k = 0, la = 0, lb = 0;
      V3<Real> nrm = v3<Real>(0, 1, 0);
      if (has_pair) k = sp_box_box<Real>(Md, S, Md.pair_a[lane], Md.pair_b[lane], scratch, nrm, la, lb);
      int before = 0, total = 0;
      for (int v = 0; v < 8; v++) { const uint64_t hm8 = __ballot(v < k); before += __popcll(hm8 & lt); total += __popcll(hm8); }
      for (int v = 0; v < k; v++) {
        const int id2 = ncp + before + v;
        if (id2 < Md.maxcp) {
          for (int t = 0; t < 4; t++) S.cpP[4 * id2 + t] = scratch[4 * v + t];
          S.cpN[3 * id2 + 0] = -nrm.x; S.cpN[3 * id2 + 1] = -nrm.y; S.cpN[3 * id2 + 2] = -nrm.z;   // into the first link
          S.cplink[id2] = la; S.cplinkB[id2] = lb;
        }
      }
      ncp = (ncp + total) < Md.maxcp ? (ncp + total) : Md.maxcp;
    }
    // contact rows: normal, two tangents
    if (lane < 3 * ncp) { S.rdof[lane] = -1; S.rfidx[lane] = (lane % 3 == 0) ? -1 : (lane - lane % 3); }
    // joint-limit rows
    const Real qd = lane < n ? S.q[lane] : Real(0);
    const bool low = lane < n && lc.d_limited && qd <= lc.d_lower;
    const bool up = lane < n && lc.d_limited && !low && qd >= lc.d_upper;
    const uint64_t lm = __ballot(low || up);
    const int row = 3 * ncp + __popcll(lm & lt);
    if ((low || up) && row < Md.maxm) {
      const Real viol = low ? qd - lc.d_lower : qd - lc.d_upper;
      const Real bounce = fmin(fmax(-viol * Md.limit_erp_dt, -Md.max_erv), Md.max_erv);
      S.rdof[row] = lane; S.rfidx[row] = -1;
      S.b[row] = bounce - S.dq[lane];   // the dt * W_i . y part (unconstrained acceleration) is added after the W solve
      S.lo[row] = low ? Real(0) : -inf_<Real>();
      S.hi[row] = low ? inf_<Real>() : Real(0);
    }
    m = 3 * ncp + __popcll(lm);
    if (EXTRAS && Md.has_joint_friction) {   // DART JointCoulombFrictionConstraint rows: joint velocity -> 0, impulse within +-mu dt
      const bool fr = lane < n && lc.d_fric > Real(0);
      const uint64_t fm = __ballot(fr);
      const int frow = m + __popcll(fm & lt);
      if (fr && frow < Md.maxm) {
        S.rdof[frow] = lane; S.rfidx[frow] = -1;
        S.b[frow] = -S.dq[lane];
        S.lo[frow] = -lc.d_fric; S.hi[frow] = lc.d_fric;
      }
      m += __popcll(fm);
    }
    m = m < Md.maxm ? m : Md.maxm;
  }
  __syncthreads();
  SP_TICK(3);
  {
    // ---- Jacobian rows (lane per row) + the generalized-force row (index m), bias part of b for contact rows
    if (lane == m) for (int k = 0; k < n; k++) S.W[m * n + k] = S.rhs[k];
    if (lane < m) {
      Real* Jr = S.W + lane * n;
      for (int k = 0; k < n; k++) Jr[k] = Real(0);
      const int d = S.rdof[lane];
      if (d >= 0) {
        Jr[d] = Real(1);
      } else {
        const int cidx = lane / 3, kind = lane % 3;
        // DART ContactConstraint tangent basis: t1 = normalize(z x n) (x x n when z and n are parallel), t2 = n x t1
        V3<Real> dir;
        if (PAIRS) {
          const V3<Real> nn = ld3(S.cpN + 3 * cidx);
          V3<Real> t1 = cross(v3<Real>(0, 0, 1), nn);
          if (dot(t1, t1) < Real(1e-12)) t1 = cross(v3<Real>(1, 0, 0), nn);
          t1 = t1 * (Real(1) / sqrt(dot(t1, t1)));
          dir = kind == 0 ? nn : (kind == 1 ? t1 : cross(nn, t1));
        } else {   // ground contacts only: n = +y, t1 = z x n = -x, t2 = n x t1 = +z
          dir = kind == 0 ? v3<Real>(0, 1, 0) : (kind == 1 ? v3<Real>(-1, 0, 0) : v3<Real>(0, 0, 1));
        }
        const V3<Real> P = ld3(S.cpP + 4 * cidx);
        Real rel = Real(0);
        for (int side = 0; side < (PAIRS ? 2 : 1); side++) {   // J = J_a - J_b for a link-link contact
          const Real sg = side == 0 ? Real(1) : Real(-1);
          for (int j = side == 0 ? S.cplink[cidx] : S.cplinkB[cidx]; j >= 0;) {
            const int w = S.topo[j];
            const int dj = topo_dof(w), jcur = j;
            j = topo_parent(w);
            if (dj < 0) continue;
            const Real* Lj = S.link + jcur * SP_LINKF;
            const V3<Real> aj = ld3(Lj + LK_A);
            const Real v = sg * ((topo_jtype(w) == 2) ? dot(dir, cross(aj, P - ld3(Lj + LK_JO))) : dot(dir, aj));
            Jr[dj] += v;
            rel += v * S.dq[dj];
          }
        }
        const Real depth = S.cpP[4 * cidx + 3];
        S.b[lane] = (kind == 0 ? fmin(depth * Md.erp_dt, Md.max_erv) : Real(0)) - rel;
        S.lo[lane] = Real(0);
        S.hi[lane] = kind == 0 ? inf_<Real>() : Real(0);   // friction rows pinned during the frictionless stage
      }
    }
    __syncthreads();
    SP_TICK(4);
    // ---- W = L^-1 [J^T | rhs] : every lane forward-substitutes its own row; row m becomes y = L^-1 rhs
    // The row lives in registers and both loops are fully unrolled (SP_MAXN x SP_MAXN / 2 predicated steps, uniform
    // `k < n` branches): every factor entry is one LDS read at an immediate offset, no index arithmetic.
    if (lane <= m) {
      Real* yrow = S.W + lane * n;
      Real y[SP_MAXN];
#pragma unroll
      for (int k = 0; k < SP_MAXN; k++) y[k] = (k < n) ? yrow[k] : Real(0);
#pragma unroll
      for (int k = 0; k < SP_MAXN; k++) {
        if (k < n) {
          Real t = y[k];
#pragma unroll
          for (int j = 0; j < k; j++) t -= S.H[TL(k, j)] * y[j];
          y[k] = t * S.sinv[k];
        }
      }
#pragma unroll
      for (int k = 0; k < SP_MAXN; k++) if (k < n) yrow[k] = y[k];
    }
    __syncthreads();
    // b_i = bounce_i - J_i (dq + dt H^-1 rhs) = bias_i - dt W_i . y
    if (lane < m) {
      const Real* wi = S.W + lane * n;
      const Real* y = S.W + m * n;
      Real t = Real(0);
      for (int k = 0; k < n; k++) t += wi[k] * y[k];
      S.b[lane] -= Md.dt * t;
    }
    __syncthreads();
    SP_TICK(5);
  }
  if (m > 0) {
    // ---- A = W W^T (lower), cfm on the diagonal.  The m(m+1)/2 entries are dealt round-robin to the 64 lanes
    // (row-per-lane would leave the last lane with m dot products and the first with one).
    {
      const int ntri = m * (m + 1) / 2;
      const Real cfm1 = Md.cfm1, ccfm1 = Md.ccfm1;
      int i = 0, base = 0;   // entry e = base + k with base = i(i+1)/2
      for (int e = lane; e < ntri; e += 64) {
        while (base + i + 1 <= e) { base += i + 1; i++; }
        const int k = e - base;
        const Real* wi = S.W + i * n;
        const Real* wk = S.W + k * n;
        Real t = Real(0);
        for (int j = 0; j < n; j++) t += wi[j] * wk[j];
        if (k == i) t *= (S.rdof[i] >= 0) ? cfm1 : ccfm1;
        S.A[e] = t;   // TI(i, k) == e for k <= i
      }
    }
    __syncthreads();
    SP_TICK(6);
    // ---- stage 1 (frictionless), stage 2 (friction bounds from the stage-1 normal impulses)
    uint64_t pinmask = 0, F = 0, U = 0;
    {
      Real bm = Real(0);
      for (int i = 0; i < m; i++) bm = fmax(bm, fabs(S.b[i]));
      const Real tol0 = tol_<Real>() * (Real(1) + bm);
      bool pinned = false, upper = false, startf = false;
      if (lane < m) {
        pinned = !(S.lo[lane] < S.hi[lane]);
        upper = !(S.lo[lane] == Real(0));
        startf = !pinned && (upper ? (S.b[lane] < -tol0) : (S.b[lane] > tol0));
      }
      pinmask = __ballot(pinned); F = __ballot(startf); U = __ballot(upper && !startf);
    }
    if (lane < m) S.x[lane] = Real(0);
    __syncthreads();
    // one inlined copy of the solver serves both stages (instruction-cache footprint)
    for (int stage = 0; stage < 2; stage++) {
      if (stage == 1) {
        SP_TICK(7);
        if (ncp == 0) break;
        bool isf = false, pinned = false;
        if (lane < m && S.rfidx[lane] >= 0) {
          const Real hb = fabs(Md.mu * S.x[S.rfidx[lane]]);
          // a direction the skeleton cannot move in (planar model, z tangent) has A_ii = 0: keep that row out
          isf = true; pinned = !(hb > Real(0)) || !(S.A[TI(lane, lane)] > Real(1e-12));
          S.hi[lane] = hb; S.lo[lane] = -hb;
        }
        __syncthreads();
        const uint64_t fr = __ballot(isf), pf = __ballot(isf && pinned);
        pinmask = (pinmask & ~fr) | pf;
        F = (F & ~fr) | (fr & ~pf);
        U &= ~fr;
      }
      sp_blcp<Real>(S, m, pinmask, F, U, Md.solver_iters, Md.pgs_fallback_sweeps, Md.stats, lane, stage == 0 && !(EXTRAS && Md.has_joint_friction));
    }
    SP_TICK(8);
    if (Md.dbg) {
      double* D = Md.dbg + (size_t)blockIdx.x * 160;
      if (lane == 0) { D[0] = m; D[1] = ncp; }
      if (lane < m) { D[2 + lane] = (double)S.x[lane]; D[42 + lane] = (double)S.b[lane]; D[82 + lane] = (double)S.hi[lane]; D[122 + lane] = (double)S.A[TI(lane, lane)]; }
    }
  }
  if (REPORT && report) {   // world.collision_result.contacts (walker2d.py:38-41, human_walker.py:97-106): point, force on the first body
    if (lane == 0) Md.creport_count[blockIdx.x] = ncp;
    if (lane < ncp) {
      Real* out = Md.creport + ((size_t)blockIdx.x * Md.maxcp + lane) * 8;
      const V3<Real> nn = ld3(S.cpN + 3 * lane);
      V3<Real> t1 = cross(v3<Real>(0, 0, 1), nn);
      if (dot(t1, t1) < Real(1e-12)) t1 = cross(v3<Real>(1, 0, 0), nn);
      t1 = t1 * (Real(1) / sqrt(dot(t1, t1)));
      const V3<Real> t2 = cross(nn, t1);
      // a tangent the skeleton cannot move along (planar model: z) has A_ii = 0 and stays pinned at a bound: no force
      const int r1 = 3 * lane + 1, r2 = 3 * lane + 2;
      const Real l0 = S.x[3 * lane], l1 = S.A[TI(r1, r1)] > Real(1e-12) ? S.x[r1] : Real(0),
                 l2 = S.A[TI(r2, r2)] > Real(1e-12) ? S.x[r2] : Real(0), idt = Real(1) / Md.dt;
      const int lb = S.cplinkB[lane];
      out[0] = (Real)Md.link_body[S.cplink[lane]]; out[1] = lb >= 0 ? (Real)Md.link_body[lb] : Real(-1);
      st3(out + 2, ld3(S.cpP + 4 * lane) + roff);
      st3(out + 5, (nn * l0 + t1 * l1 + t2 * l2) * idt);
    }
  }
  // ---- new velocity: vs = dq + L^-T (dt y + W^T lambda)
  if (lane < n) {
    Real u = Md.dt * S.W[m * n + lane], ul = Real(0);
    for (int i = 0; i < m; i++) { const Real t = S.W[i * n + lane] * S.x[i]; u += t; ul += t; }
    if ((EXTRAS && Md.task == 12) || (REPORT && report)) S.lo[lane] = ul;   // W^T lambda = L^-1 J^T lambda
    S.rhs[lane] = u;
  }
  __syncthreads();
  if ((EXTRAS && Md.task == 12) || (REPORT && report)) {   // constraint_forces() of this step: J^T lambda / dt = L (W^T lambda) / dt
    if (lane < n) {
      Real t = Real(0);
      for (int k = 0; k <= lane; k++) t += S.H[TL(lane, k)] * S.lo[k];
      S.cf[lane] = t / Md.dt;
    }
    __syncthreads();
    if (REPORT && report && lane < n) {
      Real v = S.cf[lane];
      if (EXTRAS && Md.free_root && lane < 6) {   // internal root coordinates are world-frame, DART's body-frame: tau_b = R^T tau_w
        const int g = lane < 3 ? 0 : 3, a = lane - g;
        v = S.root[a] * S.cf[g] + S.root[3 + a] * S.cf[g + 1] + S.root[6 + a] * S.cf[g + 2];
      }
      Md.cf_report[(size_t)blockIdx.x * n + lane] = v;
    }
  }
  sp_chol_backsolve<Real>(S.H, S.sinv, n, S.rhs, lane);
  SP_TICK(9);
  if (lane < n) { const Real vnew = S.dq[lane] + S.rhs[lane]; S.dq[lane] = vnew; S.q[lane] += Md.dt * vnew; }
  __syncthreads();
  if (EXTRAS && Md.free_root) {   // the six root entries just advanced are placeholders: the pose lives in S.root
    if (lane == 0) sp_free_root_advance<Real>(S, Md.dt);
    __syncthreads();
  }
  (void)nl;
}

// ------------------------------------------------------------------ task epilogues (lane 0, after a fresh kinematics pass)
// HumanWalker: reward / done / obs (human_walker.py:75-149).  Returns done.
template <class Real>
__device__ __forceinline__ bool sp_humanwalker_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before,
                                                        Real abs_a_sum, Real init_height, const int* cflags,
                                                        Real& reward_out) {
  const V3<Real> roff = ld3(S.misc);
  const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
  const Real* Lh = S.link + Md.aux_link[1] * SP_LINKF;
  const Real pos_after = Lb[LK_C] + roff.x;
  const Real height = Lh[LK_C + 1] + roff.y, side = Lh[LK_C + 2] + roff.z;
  const Real* R = Lh + LK_R;
  const V3<Real> up = v3<Real>(R[1], R[4], R[7]), fw = v3<Real>(R[0], R[3], R[6]);
  const Real ang_u = acos(fmin(fmax(up.y / sqrt(dot(up, up)), Real(-1)), Real(1)));
  const Real ang_f = acos(fmin(fmax(fw.x / sqrt(dot(fw, fw)), Real(-1)), Real(1)));
  const Real vel = (pos_after - pos_before) * Md.inv_envdt;
  const Real tv = Md.aux_real[0];
  Real rew = Real(2) * (tv - fabs(tv - vel)) + Md.aux_real[1] - Md.aux_real[2] * abs_a_sum - Md.aux_real[3] * fabs(side);
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  const Real dh = height - init_height;
  ok = ok && (dh > Md.aux_real[4]) && (dh < Md.aux_real[5]) && (fabs(ang_u) < Md.aux_real2[1]) && (fabs(ang_f) < Md.aux_real2[1]) &&
       (fabs(S.q[3]) < Md.aux_real[6]) && (fabs(S.q[5]) < Md.aux_real[7]) && (fabs(side) < Md.aux_real2[0]);
  if (!ok) rew = Real(0);
  reward_out = rew;
  (void)cflags;
  return !ok;
}

// Walker3d: reward / done (walker3d.py:44-97).  Progress, height, side deviation and the up / forward angles are those
// of bodynodes[0] (aux_link[0]); aux_link[1..2] are the two penalised dofs; aux_real = {alive, ctrl_cost, limit_penalty,
// deviation_pen, height_lo, height_hi, penalty_margin}.  Returns done.
template <class Real>
__device__ __forceinline__ bool sp_walker3d_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before,
                                                     Real sq_a_sum, Real& reward_out) {
  const V3<Real> roff = ld3(S.misc);
  const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
  const Real pos_after = Lb[LK_C] + roff.x, height = Lb[LK_C + 1] + roff.y, side = Lb[LK_C + 2] + roff.z;
  const Real* R = Lb + LK_R;
  const V3<Real> up = v3<Real>(R[1], R[4], R[7]), fw = v3<Real>(R[0], R[3], R[6]);
  const Real ang_u = acos(fmin(fmax(up.y / sqrt(dot(up, up)), Real(-1)), Real(1)));
  const Real ang_f = acos(fmin(fmax(fw.x / sqrt(dot(fw, fw)), Real(-1)), Real(1)));
  Real pen = Real(0);
  for (int k = 1; k <= 2; k++) {
    const int j = Md.aux_link[k];
    if (j < 0) continue;
    if ((Md.lower[j] - S.q[j]) > -Md.aux_real[6]) pen += Real(1.5);
    if ((Md.upper[j] - S.q[j]) < Md.aux_real[6]) pen += Real(1.5);
  }
  Real rew = Md.aux_real2[2] * ((pos_after - pos_before) * Md.inv_envdt) + Md.aux_real[0];   // weight 1 (Walker3d) / 0.45 (SPD)
  rew -= Md.aux_real[1] * sq_a_sum;
  rew -= Md.aux_real[2] * pen;
  rew -= Md.aux_real[3] * fabs(side);
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  ok = ok && (height > Md.aux_real[4]) && (height < Md.aux_real[5]) && (fabs(ang_u) < Md.aux_real2[1]) && (fabs(ang_f) < Md.aux_real2[1]);
  if (!ok && Md.task == 3) rew = Real(0);   // the SPD variant (task 12) keeps the reward of the terminal step
  reward_out = rew;
  return !ok;
}

// Hopper / Walker2d task logic for cards the planar kernels do not take (e.g. every capsule collidable): hopper.py:36-65,
// walker2d.py:22-62.  aux_link = {height body link, penalty dof or -1}; aux_real = {alive, ctrl_cost, limit_penalty, -,
// height_lo, height_hi, penalty_margin}.  Returns done; *height_out feeds observation[0].
template <class Real>
__device__ __forceinline__ bool sp_planar_task_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before, Real sq_a_sum,
                                                        Real& reward_out) {
  const Real height = S.link[Md.aux_link[0] * SP_LINKF + LK_C + 1] + S.misc[1];
  Real pen = Real(0);
  const int j = Md.aux_link[1];
  if (j >= 0) {
    if ((Md.lower[j] - S.q[j]) > -Md.aux_real[6]) pen += Real(1.5);
    if ((Md.upper[j] - S.q[j]) < Md.aux_real[6]) pen += Real(1.5);
  }
  Real rew = (S.q[0] - pos_before) * Md.inv_envdt;
  rew += Md.aux_real[0];
  rew -= Md.aux_real[1] * sq_a_sum;
  rew -= Md.aux_real[2] * pen;
  reward_out = rew;
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  ok = ok && (height > Md.aux_real[4]) && (height < Md.aux_real[5]) && (fabs(S.q[2]) < Md.aux_real2[1]);
  return !ok;
}

// Reacher tip: to_world(aux body, aux_real[0..2]) in absolute coordinates (no floating base in these models)
template <class Real>
__device__ __forceinline__ V3<Real> sp_reacher_tip(const SpatialModel<Real>& Md, SpLds<Real>& S) {
  const Real* L = S.link + Md.aux_link[0] * SP_LINKF;
  return ld3(L + LK_P) + mulR(L + LK_R, v3<Real>(Md.aux_real[0], Md.aux_real[1], Md.aux_real[2])) + ld3(S.misc);
}

// CartPole (cart_pole.py:12-24): reward 1, done when the observation is not finite or |q[1]| > angle_max.
// HalfCheetah (half_cheetah.py:43-63): aux_real = {alive, ctrl_cost}; reward zeroed when the state broke.
template <class Real>
__device__ __forceinline__ bool sp_simple_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before, Real sq_a_sum,
                                                   Real& reward_out) {
  bool fin = true, bounded = true;
  for (int i = 0; i < Md.n; i++) {
    fin = fin && isfinite(S.q[i]) && isfinite(S.dq[i]);
    bounded = bounded && (fabs(S.dq[i]) < Md.s_max) && (i < 2 || fabs(S.q[i]) < Md.s_max);
  }
  if (Md.task == 5) {
    reward_out = Md.aux_real[0];
    return !(fin && fabs(S.q[1]) <= Md.aux_real2[1]);
  }
  if (Md.task == 9) {   // snake (snake_7link.py:72-84); aux_real = {alive, ctrl_cost, deviation cost, fluid k}
    Real rew = (S.q[0] - pos_before) * Md.inv_envdt;
    rew += Md.aux_real[0];
    rew -= Md.aux_real[1] * sq_a_sum;
    rew -= fabs(S.q[2]) * Md.aux_real[2];
    reward_out = rew;
    return !(fin && bounded && fabs(S.q[2]) < Md.aux_real2[1]);
  }
  if (Md.task == 7) {   // cart-pole swing-up (cartpole_swingup.py:22-31); sq_a_sum = a^2 of the single action
    reward_out = Md.aux_real[0] - fabs(S.q[1]) - Md.aux_real[1] * sq_a_sum - Md.aux_real[2] * fabs(S.q[0]);
    return (fabs(S.q[1]) > Md.aux_real[3]) || (fabs(S.dq[1]) > Md.aux_real[4]) || (fabs(S.q[0]) > Md.aux_real[5]);
  }
  if (Md.task == 8) {   // double inverted pendulum (inverted_double_pendulum.py:27-42): tip height above the cart
    const Real base = S.link[Md.aux_link[0] * SP_LINKF + LK_P + 1], raw = S.link[Md.aux_link[1] * SP_LINKF + LK_P + 1];
    const Real height = Real(2) * (raw - base - Md.aux_real[4]) / Md.aux_real[5];
    const Real dist_pen = Md.aux_real[1] * (S.q[0] * S.q[0]) + (height - Real(2)) * (height - Real(2));
    const Real vel_pen = Md.aux_real[2] * (S.dq[1] * S.dq[1]) + Md.aux_real[3] * (S.dq[2] * S.dq[2]);
    reward_out = Md.aux_real[0] - dist_pen - vel_pen;
    return height <= Real(1);
  }
  const bool ok = fin && bounded;
  Real rew = (S.q[0] - pos_before) * Md.inv_envdt + Md.aux_real[0];
  rew -= Md.aux_real[1] * sq_a_sum;
  reward_out = ok ? rew : Real(0);
  return !(ok && fabs(S.q[2]) < Md.aux_real2[1]);
}

template <class Real>
__device__ __forceinline__ void sp_write_obs(const SpatialModel<Real>& Md, SpLds<Real>& S, const int* cflags, float* __restrict__ o, int lane) {
  const int n = Md.n;
  if (Md.task == 10 || Md.task == 11) {   // reachers: cos q, sin q, target (2-D: x, z), dq, tip - target (reacher.py:38-42)
    const V3<Real> tgt = ld3(S.misc + 4);       // staged by the caller from the per-env task state
    int o0 = 2 * n;
    if (lane < n) { Real sn, cs; sincos_<Real>(S.q[lane], sn, cs); o[lane] = (float)cs; o[n + lane] = (float)sn; }
    if (lane == 0) {
      if (Md.task == 10) { o[o0] = (float)tgt.x; o[o0 + 1] = (float)tgt.z; }
      else { o[o0] = (float)tgt.x; o[o0 + 1] = (float)tgt.y; o[o0 + 2] = (float)tgt.z; }
    }
    o0 += (Md.task == 10) ? 2 : 3;
    if (lane < n) o[o0 + lane] = (float)S.dq[lane];
    if (lane == 0) {
      const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - tgt;
      o[o0 + n] = (float)vec.x; o[o0 + n + 1] = (float)vec.y; o[o0 + n + 2] = (float)vec.z;
    }
    return;
  }
  if (Md.task == 8) {   // double pendulum: [q0, sin q1, sin q2, cos q1, cos q2, dq] (inverted_double_pendulum.py:45-51)
    if (lane == 0) o[0] = (float)S.q[0];
    if (lane == 1 || lane == 2) { Real sn, cs; sincos_<Real>(S.q[lane], sn, cs); o[lane] = (float)sn; o[lane + 2] = (float)cs; }
    if (lane < 3) o[5 + lane] = (float)S.dq[lane];
    return;
  }
  if (Md.task == 0 || Md.task == 5 || Md.task == 7) {   // physics only, CartPole, swing-up: [q, dq]
    if (lane < n) { o[lane] = (float)S.q[lane]; o[n + lane] = (float)S.dq[lane]; }
    return;
  }
  if (lane >= 1 && lane < n) o[lane - 1] = (float)S.q[lane];
  if (lane < n) o[n - 1 + lane] = (float)fmin(fmax(S.dq[lane], -Md.v_clip), Md.v_clip);
  if (Md.task == 4 && lane < 2) o[2 * n - 1 + lane] = (float)cflags[lane];   // foot-contact flags (human_walker.py:146)
  if ((Md.task == 1 || Md.task == 2) && lane == 1)   // observation[0] = COM height of the root body (hopper.py:72)
    o[0] = (float)(S.link[Md.aux_link[0] * SP_LINKF + LK_C + 1] + S.misc[1]);
}

// ------------------------------------------------------------------ kernels: one wavefront (64 threads) per env
// REPORT: the contact-report variant (dart_get_contacts); only the most general instantiation <true, true, true> is built --
// the mere presence of the reporting code costs the lean kernels 2.5 % (register allocation), so they do not carry it.
template <class Real, bool PAIRS, bool EXTRAS, bool REPORT = false>
__global__ void __launch_bounds__(64, 3) sp_step_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                      Real* __restrict__ qs, Real* __restrict__ dqs, Real* __restrict__ tstate,
                                                      int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                      const float* __restrict__ actions, float* __restrict__ obs,
                                                      float* __restrict__ reward, uint8_t* __restrict__ done,
                                                      uint8_t* __restrict__ truncated, int autoreset, uint64_t seed,
                                                      uint64_t env_offset) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  if (e >= n_envs) return;
  const int n = Md.n;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, Md.nl, n, Md.maxm, Md.maxcp);
  int* cflags = S.imisc + 2;
  Real* sh_scal = S.misc + 8;
  if (lane < n) { S.q[lane] = qs[e * n + lane]; S.dq[lane] = dqs[e * n + lane]; S.tau[lane] = Real(0); }
  if (lane < Md.nl) S.topo[lane] = (Md.parent[lane] + 1) | ((Md.dof[lane] + 1) << 8) | (Md.jtype[lane] << 16);
  if (Md.stats && lane < 10) S.ticks[lane] = 0ull;
  if (EXTRAS && Md.task == 12 && lane < n) S.cf[lane] = Md.cf_store[e * n + lane];
  __syncthreads();
  Real abs_sum = Real(0), sq_sum = Real(0);
  if (lane == 0) {
    for (int k = 0; k < Md.act_dim; k++) {
      const Real a = (Real)actions[e * Md.act_dim + k];
      abs_sum += fabs(a); sq_sum += a * a;
      Real cl = (a > Md.act_hi[k]) ? Md.act_hi[k] : a;
      cl = (cl < Md.act_lo[k]) ? Md.act_lo[k] : cl;
      const int dd = Md.act_dof0 + k;
      if (EXTRAS && Md.task == 12)   // SPD: the action is a target pose inside the joint's limits (walker3d_spd.py:68-70)
        S.tau[dd] = (cl + Real(1)) / Real(2) * (Md.upper[dd] - Md.lower[dd]) + Md.lower[dd];
      else
        S.tau[dd] = cl * Md.act_scale[k];
    }
    if (EXTRAS && Md.free_root) { sp_free_root_load<Real>(S); sp_free_root_to_internal<Real>(S); }   // S.q / S.dq: internal from here
    // link poses are needed before the step only by the tasks that measure progress on a body (3, 4) or a tip (11), after
    // it only by the tasks whose reward / done / observation read a body pose
    if (Md.task == 3 || Md.task == 4 || Md.task == 11 || Md.task == 12 || Md.task == 13) sp_kinematics<Real>(Md, S);
    sh_scal[0] = (Md.task == 3 || Md.task == 4 || Md.task == 12 || Md.task == 13) ? S.link[Md.aux_link[0] * SP_LINKF + LK_C] + S.misc[0] : S.q[0];   // posbefore
    if (Md.task == 11) {   // DartReacher3d: distance to the target BEFORE the step, and sum tau^2 (reacher.py:23-27)
      const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - ld3(tstate + 4 * e);
      sh_scal[0] = sqrt(dot(vec, vec));
      Real st2 = Real(0);
      for (int k = 0; k < Md.act_dim; k++) st2 += S.tau[Md.act_dof0 + k] * S.tau[Md.act_dof0 + k];
      sh_scal[2] = st2;
    }
    cflags[0] = 0; cflags[1] = 0;
  }
  __syncthreads();
  LinkConst<Real> lc;
  sp_load_link_const<Real>(Md, lane < Md.nl ? lane : 0, lc);
  for (int f = 0; f < Md.frame_skip; ++f) sp_world_step<Real, PAIRS, EXTRAS, REPORT>(Md, lc, S, lane, cflags, REPORT && Md.creport != nullptr && f == Md.frame_skip - 1);
  if (Md.stats && lane < 10) atomicAdd(&Md.stats[40 + lane], S.ticks[lane]);
  bool dn = false, tr = false;
  const bool pose_last = Md.task == 1 || Md.task == 2 || Md.task == 3 || Md.task == 4 || Md.task == 8 || Md.task >= 10;
  const bool pose_reset = pose_last && Md.task != 13;   // the dog's observation holds no body pose
  if (lane == 0) {
    if (EXTRAS && Md.free_root) sp_free_root_to_internal<Real>(S);   // internal coordinates of the final pose
    if (pose_last) sp_kinematics<Real>(Md, S);
    Real rew = Real(0);
    bool task_done = false;
    Real dog_x = Real(0), dog_h = Real(0), dog_side = Real(0);
    if (Md.task == 13) {
      const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
      dog_x = Lb[LK_C] + S.misc[0]; dog_h = Lb[LK_C + 1] + S.misc[1]; dog_side = fabs(Lb[LK_C + 2] + S.misc[2]);
    }
    if (EXTRAS && Md.free_root) sp_free_root_store<Real>(S);          // S.q / S.dq: public coordinates again
    if (Md.task == 13) {   // DartDogEnv.step (dog.py:28-46)
      rew = Md.aux_real[1] * (dog_x - sh_scal[0]) * Md.inv_envdt;
      rew += Md.aux_real[0];
      rew -= Md.aux_real[2] * sq_sum;
      bool ok = true;
      for (int i = 0; i < Md.n; i++) {
        ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
        if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
      }
      task_done = !(ok && dog_h > Md.aux_real[4] && dog_h < Md.aux_real[5] && dog_side < Md.aux_real[3]);
    }
    if (Md.task == 4) task_done = sp_humanwalker_epilogue<Real>(Md, S, sh_scal[0], abs_sum, tstate[4 * e], cflags, rew);
    else if (Md.task == 3 || Md.task == 12) task_done = sp_walker3d_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    else if (Md.task >= 5 && Md.task != 13) task_done = sp_simple_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    else if (Md.task == 1 || Md.task == 2) task_done = sp_planar_task_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    if (Md.task == 10 || Md.task == 11) {   // reachers: reward from the tip-target distance (2-D: after, 3-D: before the step)
      const V3<Real> tgt = ld3(tstate + 4 * e);
      bool fin = true;
      for (int i = 0; i < Md.n; i++) fin = fin && isfinite(S.q[i]) && isfinite(S.dq[i]);
      if (Md.task == 11) {
        const Real dist0 = sh_scal[0];
        rew = -dist0 + -(sh_scal[2] * Md.aux_real[3]);
        task_done = !(fin && (dist0 > Md.aux_real[4]));
      } else {
        const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - tgt;
        rew = -sqrt(dot(vec, vec)) + -sq_sum;
        task_done = false;
      }
    }
    int el = elapsed[e] + 1;
    const bool trunc = (Md.max_steps > 0) && (el >= Md.max_steps);
    dn = task_done || trunc; tr = trunc && !task_done;
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = tr ? 1 : 0;
    elapsed[e] = (autoreset && dn) ? 0 : el;
    sh_scal[1] = dn ? Real(1) : Real(0);
  }
  __syncthreads();
  dn = sh_scal[1] != Real(0);
  if (autoreset && dn) {
    const uint32_t ep = episode[e] + 1;
    // Philox reset noise, same stream as the planar kernels / the oracle: u[0..n) positions, u[n..2n) velocities
    if (lane < n) {
      const int iq = lane, iv = n + lane;
      uint32_t o[4];
      philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iq / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
      const Real uq = Real(o[iq % 4] >> 8) * Real(1.0 / 16777216.0);
      philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iv / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
      const Real uv = Real(o[iv % 4] >> 8) * Real(1.0 / 16777216.0);
      S.q[lane] = Md.q0[lane] + (-Md.noise + Real(2) * Md.noise * uq);
      S.dq[lane] = Md.dq0[lane] + (-Md.noise_v + Real(2) * Md.noise_v * uv);
    }
    __syncthreads();
    if (lane == 0) {
      episode[e] = ep;
      if (pose_reset) sp_kinematics<Real>(Md, S);
      if (Md.task == 4) tstate[4 * e] = S.link[Md.aux_link[1] * SP_LINKF + LK_C + 1] + S.misc[1];
      cflags[0] = 0; cflags[1] = 0;
    }
    __syncthreads();
  }
  if (lane < n) { qs[e * n + lane] = S.q[lane]; dqs[e * n + lane] = S.dq[lane]; }
  if (EXTRAS && Md.task == 12 && lane < n) Md.cf_store[e * n + lane] = (autoreset && dn) ? Real(0) : S.cf[lane];
  if ((Md.task == 10 || Md.task == 11) && lane < 3) S.misc[4 + lane] = tstate[4 * e + lane];
  __syncthreads();
  sp_write_obs<Real>(Md, S, cflags, obs + e * Md.obs_dim, lane);
}

// Dynamics quantities of the CURRENT state (pydart2's skel.M and skel.c, reference gym/envs/dart/walker3d_spd.py:40-55):
// mass matrix (n x n, symmetric, without the implicit damping / stiffness terms) and Coriolis + gravity forces.
// One wavefront per env; `soa` tells how the owning implementation stores its state (planar kernels: q[n][N]).
template <class Real>
__global__ void __launch_bounds__(64) sp_dynamics_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                          const Real* __restrict__ qs, const Real* __restrict__ dqs, int soa,
                                                          double* __restrict__ mass_out, double* __restrict__ bias_out,
                                                          double* __restrict__ pose_out, int nbodies) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  if (e >= n_envs) return;
  const int n = Md.n, nl = Md.nl;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, nl, n, Md.maxm, Md.maxcp);
  if (lane < n) {
    const int64_t at = soa ? (int64_t)lane * n_envs + e : e * n + lane;
    S.q[lane] = qs[at]; S.dq[lane] = dqs[at]; S.tau[lane] = Real(0);
  }
  if (lane < nl) S.topo[lane] = (Md.parent[lane] + 1) | ((Md.dof[lane] + 1) << 8) | (Md.jtype[lane] << 16);
  __syncthreads();
  if (pose_out) {   // bodynode world transforms / COMs: R (9), origin (3), com (3) per card body; cold path, serial kinematics
    if (lane == 0) {
      if (Md.free_root) { sp_free_root_load<Real>(S); sp_free_root_to_internal<Real>(S); }
      sp_kinematics<Real>(Md, S);
    }
    __syncthreads();
    if (lane < nl && Md.link_body[lane] >= 0) {
      const Real* L = S.link + lane * SP_LINKF;
      double* o = pose_out + ((size_t)e * nbodies + Md.link_body[lane]) * 15;
      // the root translation joints are factored out of the link records (S.misc): a carrier body of that chain has only
      // the slides up to its own joint behind it, every other body all of them
      V3<Real> off = v3<Real>(0, 0, 0);
      for (int j = 0; j < nl; j++)
        if (Md.root_trans[j] && (j <= lane || !Md.root_trans[lane])) off = off + ld3(S.link + j * SP_LINKF + LK_A) * S.q[Md.dof[j]];
      const Real offv[3] = {off.x, off.y, off.z};
      for (int k = 0; k < 9; k++) o[k] = (double)L[LK_R + k];
      for (int k = 0; k < 3; k++) { o[9 + k] = (double)(L[LK_P + k] + offv[k]); o[12 + k] = (double)(L[LK_C + k] + offv[k]); }
    }
    if (!mass_out && !bias_out) return;
    __syncthreads();
    if (lane < n) { const int64_t at = soa ? (int64_t)lane * n_envs + e : e * n + lane; S.q[lane] = qs[at]; S.dq[lane] = dqs[at]; }
    __syncthreads();
  }
  LinkConst<Real> lc;
  sp_load_link_const<Real>(Md, lane < nl ? lane : 0, lc);
  if (lane == 0) sp_root_offset<Real>(Md, S);
  sp_forward<Real>(lc, Md, S, lane);
  __syncthreads();
  for (int lv = Md.n_group_levels - 1; lv >= 0; lv--) {
    if (lane < nl && lc.group_level == lv) sp_gather_children<Real>(lc, Md, S, lane);
    __syncthreads();
  }
  if (lane < nl) sp_link_rhs<Real>(lc, Md, S, lane);
  __syncthreads();
  if (bias_out && lane < nl && lc.dof >= 0) {
    const Real* L = S.link + lane * SP_LINKF;
    const V3<Real> a = ld3(L + LK_A);
    bias_out[e * n + lc.dof] = (double)((lc.jtype == 2) ? dot(a, ld3(L + LK_N)) : dot(a, ld3(L + LK_F)));
  }
  if (mass_out) {
    if (lane < n) sp_mass_row<Real>(lc, Md, S, lane);
    __syncthreads();
    if (lane < n) {
      double* Mo = mass_out + e * n * n;
      for (int k = 0; k <= lane; k++) {
        double v = (double)S.H[TL(lane, k)];
        if (k == lane) v -= (double)lc.d_diag;
        Mo[lane * n + k] = v; Mo[k * n + lane] = v;
      }
    }
  }
}

// per-env task state (reach targets): masked copy of (N, 4) doubles
template <class Real>
__global__ void sp_task_state_kernel(int64_t n_envs, const uint8_t* __restrict__ mask, const double* __restrict__ values, Real* __restrict__ tstate) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs || (mask && !mask[e])) return;
  for (int k = 0; k < 4; k++) tstate[4 * e + k] = (Real)values[4 * e + k];
}

// masked reset: q = init + noise (host rows or Philox), elapsed = 0, init height, obs
template <class Real>
__global__ void __launch_bounds__(64) sp_reset_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                       Real* __restrict__ qs, Real* __restrict__ dqs, Real* __restrict__ tstate,
                                                       int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                       const uint8_t* __restrict__ mask, const double* __restrict__ qnoise,
                                                       const double* __restrict__ vnoise, float* __restrict__ obs,
                                                       uint64_t seed, uint64_t env_offset, int obs_masked_only) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  if (e >= n_envs) return;
  const int n = Md.n;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, Md.nl, n, Md.maxm, Md.maxcp);
  int* cflags = S.imisc + 2;
  const bool m = (mask == nullptr) || mask[e];
  if (lane < n) {
    if (m) {
      if (qnoise) { S.q[lane] = (Real)qnoise[e * n + lane]; S.dq[lane] = (Real)vnoise[e * n + lane]; }
      else {
        const uint32_t ep = episode[e] + 1;
        const int iq = lane, iv = n + lane;
        uint32_t o[4];
        philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iq / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const Real uq = Real(o[iq % 4] >> 8) * Real(1.0 / 16777216.0);
        philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iv / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const Real uv = Real(o[iv % 4] >> 8) * Real(1.0 / 16777216.0);
        S.q[lane] = Md.q0[lane] + (-Md.noise + Real(2) * Md.noise * uq);
        S.dq[lane] = Md.dq0[lane] + (-Md.noise_v + Real(2) * Md.noise_v * uv);
      }
    } else { S.q[lane] = qs[e * n + lane]; S.dq[lane] = dqs[e * n + lane]; }
  }
  __syncthreads();
  if (lane == 0) {
    cflags[0] = 0; cflags[1] = 0;
    if (m) {
      if (!qnoise) episode[e] = episode[e] + 1;
      elapsed[e] = 0;
    }
    if (m || (obs && !obs_masked_only && (Md.task == 1 || Md.task == 2 || Md.task == 10 || Md.task == 11))) {   // these observations need the pose
      sp_kinematics<Real>(Md, S);
      if (Md.task == 4) tstate[4 * e] = S.link[Md.aux_link[1] * SP_LINKF + LK_C + 1] + S.misc[1];
    }
  }
  __syncthreads();
  if (m && lane < n) { qs[e * n + lane] = S.q[lane]; dqs[e * n + lane] = S.dq[lane]; }
  if (m && Md.task == 12 && Md.cf_store && lane < n) Md.cf_store[e * n + lane] = Real(0);   // world.reset() clears them
  if ((Md.task == 10 || Md.task == 11) && lane < 3) S.misc[4 + lane] = tstate[4 * e + lane];
  __syncthreads();
  if (obs && (m || !obs_masked_only)) sp_write_obs<Real>(Md, S, cflags, obs + e * Md.obs_dim, lane);
}

// (N, n) doubles <-> the kernel's AoS state
template <class Real>
__global__ void sp_state_io_kernel(int64_t count, Real* __restrict__ qs, Real* __restrict__ dqs, double* __restrict__ qh,
                                   double* __restrict__ dqh, int to_device) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (to_device) { qs[i] = (Real)qh[i]; dqs[i] = (Real)dqh[i]; }
  else { qh[i] = (double)qs[i]; dqh[i] = (double)dqs[i]; }
}

}  // namespace dartk
