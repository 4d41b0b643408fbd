// spatial_dynamics.hpp -- tree recursions of the tree kernel: kinematics, log-depth forward pass, group-level backward pass, bias forces, mass-matrix rows.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include "spatial_model.hpp"
#include "tree_patterns.hpp"

namespace dartk {

// ------------------------------------------------------------------ tree recursions
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
  // pattern kernels (skyline storage of H, tree_patterns.hpp): of factor row `lane` -- base offset, mask of its structural columns -- and
  // the offset of the diagonal entry of dof `lane` (row n-1-lane); filled by sp_load_pattern_const, unused by the dense kernels
  int h_rb; uint32_t h_mask; int h_diag;
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
  c.h_rb = 0; c.h_mask = 0u; c.h_diag = 0;
}
// Per-lane constants of a pattern's skyline H storage: the tables are compile-time, the index is the lane -- looked up ONCE per kernel
// (a constant-memory load per lane) and kept in registers, like the rest of LinkConst.
template <class PAT, class Real>
__device__ __forceinline__ void sp_load_pattern_const(int lane, LinkConst<Real>& c) {
  if constexpr (!PAT::dense) {
    const int r = lane < PAT::n ? lane : 0;
    c.h_rb = PAT::hbase(r); c.h_mask = lane < PAT::n ? PAT::row(r) : 0u;
    const int rd = lane < PAT::n ? PAT::n - 1 - lane : 0;
    c.h_diag = PAT::hbase(rd) + rd;
  }
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
// POSE_ONLY: steps 1-2 only (link frames, joint origins / axes, COMs go to LDS) -- the pose the task code reads before and
// after the world steps, which the first version computed link after link on lane 0.
// PAT: a pattern kernel's compile-time model dimensions (tree_patterns.hpp); DensePattern = read them from the model.
template <class Real, bool EXTRAS = false, bool POSE_ONLY = false, class PAT = DensePattern>
__device__ __forceinline__ void sp_forward(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane,
                                           int64_t env = 0) {
  const bool live = lane < (PAT::dense ? Md.nl : PAT::nl);
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
    if (rev) sincos_remat_<Real>(qv, sn, cs);   // (constants rematerialised at their use: planar_kernel.hpp)
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
  const int nr = PAT::dense ? Md.nrounds : PAT::nrounds;
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
  if constexpr (POSE_ONLY) {
    if (live) {
      Real* L = S.link + lane * SP_LINKF;
      for (int k = 0; k < 9; k++) L[LK_R + k] = R[k];
      st3(L + LK_P, p); st3(L + LK_JO, pj); st3(L + LK_A, a); st3(L + LK_C, c);
    }
    return;
  }
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
  Real* D = S.ldyn + lane * SP_LDYN;
  st3(D + LD_F, f);
  st3(D + LD_N, nj);
  D[LD_MC] = m;
  st3(D + LD_H, dj * m);
  const Real d2 = dot(dj, dj);
  D[LD_IC + 0] = Iw[0] + m * (d2 - dj.x * dj.x);
  D[LD_IC + 1] = Iw[1] - m * dj.x * dj.y;
  D[LD_IC + 2] = Iw[2] - m * dj.x * dj.z;
  D[LD_IC + 3] = Iw[4] + m * (d2 - dj.y * dj.y);
  D[LD_IC + 4] = Iw[5] - m * dj.y * dj.z;
  D[LD_IC + 5] = Iw[8] + m * (d2 - dj.z * dj.z);
}

// Link poses of the current S.q for the task code (all 64 lanes call; ends with a barrier): what sp_kinematics computes
// serially, in O(log depth) wave steps.  A free root must already be in internal coordinates (sp_free_root_to_internal).
template <class Real, bool EXTRAS, class PAT = DensePattern>
__device__ __forceinline__ void sp_pose_pass(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane);

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

template <class Real, bool EXTRAS, class PAT>
__device__ __forceinline__ void sp_pose_pass(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane) {
  __syncthreads();
  if (lane == 0) sp_root_offset<Real>(Md, S);
  sp_forward<Real, EXTRAS, true, PAT>(lc, Md, S, lane);
  __syncthreads();
}

// parent-centric backward step for group leader i (all child groups are complete): gather their wrenches and composite
// bodies (lc.children holds the child groups' leaders)
template <class Real>
__device__ __forceinline__ void sp_gather_children(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int i) {
  Real* Dp = S.ldyn + i * SP_LDYN;
  V3<Real> F = ld3(Dp + LD_F), N = ld3(Dp + LD_N), H = ld3(Dp + LD_H);
  Real mcp = Dp[LD_MC];
  Real I0 = Dp[LD_IC + 0], I1 = Dp[LD_IC + 1], I2 = Dp[LD_IC + 2], I3 = Dp[LD_IC + 3], I4 = Dp[LD_IC + 4], I5 = Dp[LD_IC + 5];
  const V3<Real> jop = ld3(S.link + i * SP_LINKF + LK_JO);
  for (int ci = 0; ci < lc.nchild; ci++) {
    const int c = (int)((lc.children >> (8 * ci)) & 0xffull);
    const Real* D = S.ldyn + c * SP_LDYN;
    const V3<Real> o = ld3(S.link + c * SP_LINKF + LK_JO) - jop, Fc = ld3(D + LD_F);
    F = F + Fc;
    N = N + ld3(D + LD_N) + cross(o, Fc);
    const Real mc = D[LD_MC];
    const V3<Real> h = ld3(D + LD_H);
    const Real diag = Real(2) * dot(o, h) + mc * dot(o, o);
    I0 += D[LD_IC + 0] + diag - Real(2) * h.x * o.x - mc * o.x * o.x;
    I1 += D[LD_IC + 1] - (h.x * o.y + o.x * h.y) - mc * o.x * o.y;
    I2 += D[LD_IC + 2] - (h.x * o.z + o.x * h.z) - mc * o.x * o.z;
    I3 += D[LD_IC + 3] + diag - Real(2) * h.y * o.y - mc * o.y * o.y;
    I4 += D[LD_IC + 4] - (h.y * o.z + o.y * h.z) - mc * o.y * o.z;
    I5 += D[LD_IC + 5] + diag - Real(2) * h.z * o.z - mc * o.z * o.z;
    H = H + h + o * mc;
    mcp += mc;
  }
  st3(Dp + LD_F, F); st3(Dp + LD_N, N); st3(Dp + LD_H, H);
  Dp[LD_MC] = mcp;
  Dp[LD_IC + 0] = I0; Dp[LD_IC + 1] = I1; Dp[LD_IC + 2] = I2; Dp[LD_IC + 3] = I3; Dp[LD_IC + 4] = I4; Dp[LD_IC + 5] = I5;
}
// every link of a group takes the leader's composite (same joint origin, massless carriers), then emits its rhs entry
template <class Real, bool EXTRAS = false>
__device__ __forceinline__ void sp_link_rhs(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int i) {
  const Real* L = S.link + i * SP_LINKF;
  Real* D = S.ldyn + i * SP_LDYN;
  if (lc.group_leader != i) {
    const Real* G = S.ldyn + lc.group_leader * SP_LDYN;
    for (int k = 0; k < 16; k++) D[k] = G[k];   // (the 17th Real is padding)
  }
  const int d = lc.dof;
  if (d >= 0) {
    const V3<Real> a = ld3(L + LK_A);
    const Real Cb = (lc.jtype == 2) ? dot(a, ld3(D + LD_N)) : dot(a, ld3(D + LD_F));
    if (EXTRAS && Md.task == 12) {   // SPD: S.tau holds the target pose; the torque is added once M and c are known
      S.b[d] = Cb;
      S.rhs[d] = -Cb - lc.damp * S.dq[d] - lc.stiff * (S.q[d] + Md.dt * S.dq[d] - lc.rest);
    } else {
      S.rhs[d] = S.tau[d] - Cb - lc.damp * S.dq[d] - lc.stiff * (S.q[d] + Md.dt * S.dq[d] - lc.rest);
    }
  }
}

// Entries (d, ancestors of d) of the mass matrix: one lane per dof walks its ancestor chain.  STORAGE ORDER IS REVERSED: dof d lives
// at row / column n-1-d of S.H, so that the Cholesky factor eliminates leaves first and the trunk last (Featherstone's LTL order:
// no fill-in, L_ij != 0 only where dof(i) is an ancestor of dof(j)); everything between the Jacobian rows and the final
// back-substitution works in storage order.  The caller has zero-filled S.H (and barriered) before.
// add_diag: M + dt D + dt^2 K instead of M.
template <class Real>
__device__ __forceinline__ void sp_mass_row(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int d, bool add_diag) {
  const int i = lc.d_link;
  const Real* L = S.link + i * SP_LINKF;
  const Real* D = S.ldyn + i * SP_LDYN;
  const V3<Real> a = ld3(L + LK_A), h = ld3(D + LD_H), jo = ld3(L + LK_JO);
  V3<Real> Lm, K;
  if (topo_jtype(S.topo[i]) == 2) {
    Lm = cross(a, h);
    const Real* I = D + LD_IC;
    K = v3<Real>(I[0] * a.x + I[1] * a.y + I[2] * a.z, I[1] * a.x + I[3] * a.y + I[4] * a.z, I[2] * a.x + I[4] * a.y + I[5] * a.z);
  } else {
    Lm = a * D[LD_MC];
    K = cross(h, a);
  }
  const int n1 = Md.n - 1;
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
    if (dj == d && add_diag) v += lc.d_diag;   // the implicit damping / spring terms E = dt D + dt^2 K (A3: only when the impulse pass runs on M + E)
    S.H[HI(n1 - d, n1 - dj)] = v;   // symmetric index: a free root's rotation dofs (0..2) hang below its translation dofs (3..5)
  }
}

// The same entries without the walk (round 4): the model lists the structurally non-zero pairs (d, dj) -- SpatialModel::mpairs, 251 for
// HumanWalker -- and the 64 lanes take them round-robin, every entry from LDS reads that depend on nothing but the entry: ~4 entries
// per lane instead of a 12-hop dependent chain through LDS on 29 lanes (fp64 HumanWalker: 9.5 k -> ~2 k cycles per world step).
// The integrator's diagonal terms (add_diag: the A3 knob at 0) are added by the caller after its barrier.
template <class Real, class PAT = DensePattern>
__device__ __forceinline__ void sp_mass_entries(const SpatialModel<Real>& Md, SpLds<Real>& S, int lane) {
  const int n1 = (PAT::dense ? Md.n : PAT::n) - 1, np_ = PAT::dense ? Md.n_mpairs : PAT::n_mpairs;
  for (int e = lane; e < np_; e += 64) {
    const uint32_t pr = Md.mpairs[e];
    const int i = pr & 0xff, jl = (pr >> 8) & 0xff, d = (pr >> 16) & 0xff, dj = pr >> 24;
    const Real* L = S.link + i * SP_LINKF;
    const Real* D = S.ldyn + i * SP_LDYN;
    const V3<Real> a = ld3(L + LK_A), h = ld3(D + LD_H), jo = ld3(L + LK_JO);
    V3<Real> Lm, K;
    if (topo_jtype(S.topo[i]) == 2) {
      Lm = cross(a, h);
      const Real* I = D + LD_IC;
      K = v3<Real>(I[0] * a.x + I[1] * a.y + I[2] * a.z, I[1] * a.x + I[3] * a.y + I[4] * a.z, I[2] * a.x + I[4] * a.y + I[5] * a.z);
    } else {
      Lm = a * D[LD_MC];
      K = cross(h, a);
    }
    const Real* Lj = S.link + jl * SP_LINKF;
    const V3<Real> aj = ld3(Lj + LK_A);
    const Real v = (topo_jtype(S.topo[jl]) == 2) ? dot(aj, K + cross(jo - ld3(Lj + LK_JO), Lm)) : dot(aj, Lm);
    // where the entry goes depends on the layout of S.H: the padded rows of the dense kernels, a pattern kernel's skyline -- the host
    // tabulated it for the kernel it launches (SpatialModel::mpair_off; dense: HI(n1 - d, n1 - dj))
    (void)n1; (void)d; (void)dj;
    S.H[Md.mpair_off[e]] = v;
  }
}

}  // namespace dartk
