// spatial_model.hpp -- model constants (SpatialModel), small vector helpers and the per-env LDS block (SpLds) of the tree kernel.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
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
// built, the per-link pose records only before: the two share LDS, the block is sized for the larger one (sp_link_reals).
__device__ __host__ constexpr bool sp_lw_aliases_links(int nl, int maxm) { return true; }
__device__ __host__ constexpr int sp_npad(int n) { return (n + 7) & ~7; }   // H is stored padded with identity rows to a multiple of 8
__device__ __host__ constexpr int TL(int i, int j) { return i * (i + 1) / 2 + j; }   // caller guarantees i >= j
// The mass matrix / its Cholesky factor is stored lower-triangular with rows padded to multiples of 4 entries and 16-byte
// aligned row starts, so that a row is read with 128-bit LDS loads (the triangular solves read every factor entry once per
// constraint row): rows 4g .. 4g+3 hold 4(g+1) entries each.
__device__ __host__ constexpr int HR(int i) { return 8 * (i >> 2) * ((i >> 2) + 1) + 4 * ((i >> 2) + 1) * (i - 4 * (i >> 2)); }   // row offset
__device__ __host__ constexpr int HL(int i, int j) { return HR(i) + j; }   // caller guarantees i >= j
__device__ __host__ constexpr int HI(int i, int j) { return i >= j ? HR(i) + j : HR(j) + i; }
constexpr int SP_LINKF = 21;  // Reals of a link's POSE record in LDS (S.link): R 9, origin 3, joint origin 3, joint axis 3, COM 3
constexpr int SP_LDYN = 17;   // Reals of its DYNAMICS record (S.ldyn): wrench F 3, N 3, composite mass 1, first moment 3, inertia 6 -- and one of
                              // padding: lane i works on record i, and a stride of 16 Reals puts every lane on the same LDS banks (measured: Dog +4 %)
// the link block holds the pose records and, once they are dead, the pivoting solver's matrix / workspace + its start vector
__device__ __host__ constexpr int sp_link_reals(int nl, int maxm) { return nl * SP_LINKF > sp_tri(maxm) + maxm ? nl * SP_LINKF : sp_tri(maxm) + maxm; }
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
  // entry-parallel assembly (round 4): no ancestor pointer chasing inside a world step
  uint32_t anc_dofs[SP_MAXL];        // bit d: dof d moves link i (d belongs to a link on the path i -> root, i included)
  int n_mpairs;                      // structurally non-zero entries (d, dj) of the mass matrix, dj on d's path to the root
  uint32_t mpairs[SP_MAXN * (SP_MAXN + 1) / 2];   // link(d) | link(dj) << 8 | d << 16 | dj << 24
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
  int impulse_M;                     // card.impulse_inertia (A3): 1 = the impulse pass runs on M (DART 6), 0 = on M + dt D + dt^2 K
  int has_implicit;                  // some dof has damping or a spring: with impulse_M the forward dynamics needs its own factor of M + E
  int fd_passes;                     // 2: trip count of sp_world_step's factorisation loop (a model constant the compiler cannot see, so the loop stays one)
  int free_root;                     // 1: body 0 hangs on a DART FreeJoint (public q[0:3] rotation vector, dq[0:6] body twist)
  int free_link;                     // the last of the six root links (carries the body); its joint rotation is Rz(c) R0
  int maxm, maxcp;                   // LCP rows / contact points this model's LDS block is carved for
  int reg_lcp;                       // 1: the model runs the BIG kernels with <= 40 LCP rows and no link-link contacts: the LDS solver's
                                     // workspace is never touched and A (written after the link records' last use) takes its alias slot
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
  // launch order (dart_debug_sched): workgroup b steps env sched_perm[b] (null: env b) and leaves its duration in s_memtime ticks in
  // sched_cost[env] -- the most expensive envs of the previous step can be dispatched first (longest-processing-time-first packing)
  const int* sched_perm;
  unsigned int* sched_cost;
  // Storage of H / its factor in LDS (round 5).  hreals: Reals of the block -- HR(sp_npad(n)) in the padded dense row layout, a pattern's
  // skyline size (tree_patterns.hpp: PAT::hreals) when the step kernel the host will launch is that pattern's; every kernel that carves the
  // block (step, reset) must agree on it, so the host states it here.  mpair_off[e]: where entry e of `mpairs` goes in THAT layout.
  int hreals;
  uint16_t mpair_off[SP_MAXN * (SP_MAXN + 1) / 2];
};

// 128-bit LDS access of `width` consecutive Reals
template <class Real> struct sp_vec128;
template <> struct sp_vec128<float> { using type = float4; static constexpr int width = 4; };
template <> struct sp_vec128<double> { using type = double2; static constexpr int width = 2; };

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
enum { LK_R = 0, LK_P = 9, LK_JO = 12, LK_A = 15, LK_C = 18 };
// ... and of its dynamics record.  Round 4: the two were one 37-Real record; the dynamics part is dead once the mass matrix and the
// right-hand side are assembled, before anything is written to the Jacobian block W -- it lives THERE now (S.ldyn = S.W), and the link
// block shrinks to what the pivoting solver's matrix needs anyway (it aliases the pose records): HumanWalker fp64 26 928 -> 22 776 B of
// LDS per env = seven workgroups per CU instead of six.
enum { LD_F = 0, LD_N = 3, LD_MC = 6, LD_H = 7, LD_IC = 10 };

// Reals of the Jacobian block W: maxm rows of n (round 5: the spare row that carried the right-hand side through the substitution until
// round 3 is gone -- no kernel touches row maxm any more), and room for the forward dynamics' own factor of M + E (padded rows, the
// reciprocal diagonal and one vector) that lives there before the Jacobian rows are written (sp_world_step, A3)
__device__ __host__ constexpr int sp_w_reals(int n, int maxm, int nl = 0) {   // nl: ... and for the links' dynamics records
  const int w = maxm * n > HR(sp_npad(n)) + 2 * sp_npad(n) ? maxm * n : HR(sp_npad(n)) + 2 * sp_npad(n);
  return w > nl * SP_LDYN ? w : nl * SP_LDYN;
}

template <class Real>
struct SpLds {
  Real* link;    // [nl][SP_LINKF] pose records
  Real* ldyn;    // [nl][SP_LDYN] dynamics records (= W: dead before the Jacobian block is written)
  Real* q; Real* dq; Real* tau; Real* rhs;   // [n]
  Real* H;       // [n(n+1)/2] packed lower triangle -> Cholesky factor
  Real* W;       // [maxm][n]: constraint Jacobian rows, then W = L^-1 J^T
  Real* A;       // [tri(maxm)] packed symmetric
  Real* Lw;      // [tri(maxm)] packed lower
  Real* b; Real* lo; Real* hi; Real* x; Real* r; Real* x0;   // [maxm]
  int* rdof;     // [maxm] limit rows: dof index, contact rows: -1
  int* rfidx;    // [maxm] friction rows: index of their normal row, else -1
  Real* cpP;     // [maxcp][4]: contact point (relative coords) + depth
  Real* cpN;     // [maxcp][3]: contact normal, pointing into the first link (ground: +y)
  int* cplink;   // [maxcp] first link
  int* cplinkB;  // [maxcp] second link of a link-link contact, -1 for the ground
  Real* sinv;    // [sp_npad(n)]: 1 / L_jj of the mass-matrix Cholesky factor
  Real* root;    // [24] free root joint: R (9), p (3), body twist w v (6); 6 spare
  Real* cf;      // [n]: J^T lambda / dt of the previous world step (pydart2 constraint_forces(), SPD task only)
  Real* misc;    // [16]: roff(3), scalars
  int* imisc;    // [8]: ncp, m, contact flags
  unsigned long long* ticks;   // [10] phase cycle counters of this env-step (diagnostics, only touched when stats are on)
  int* topo;     // [nl]: (parent + 1) | (dof + 1) << 8 | jtype << 16 -- ancestor walks read this instead of global memory
  int* ancd;     // [nl]: SpatialModel::anc_dofs, the dofs that move each link (Jacobian rows are built entry by entry from it)
};
__device__ __forceinline__ int topo_parent(int w) { return (w & 0xff) - 1; }
__device__ __forceinline__ int topo_dof(int w) { return ((w >> 8) & 0xff) - 1; }
__device__ __forceinline__ int topo_jtype(int w) { return (w >> 16) & 0xff; }

// hreals: Reals of the H block (SpatialModel::hreals; 0 = the padded dense layout HR(sp_npad(n)))
__device__ __host__ constexpr int sp_h_reals(int n, int hreals) { return hreals > 0 ? hreals : HR(sp_npad(n)); }
template <class Real>
__device__ __forceinline__ SpLds<Real> sp_carve(Real* base, int nl, int n, int maxm, int maxcp, int reg_lcp, int hreals = 0) {
  SpLds<Real> S;
  Real* p = base;
  S.link = p; p += sp_link_reals(nl, maxm);
  S.q = p; p += n; S.dq = p; p += n; S.tau = p; p += n; S.rhs = p; p += n;
  p = base + (((p - base) + 3) & ~3);   // 16-byte aligned rows
  S.H = p; p += sp_h_reals(n, hreals);
  S.W = p; S.ldyn = p; p += sp_w_reals(n, maxm, nl);
  const bool alias = sp_lw_aliases_links(nl, maxm);
  if (reg_lcp && alias) { S.A = S.link; S.x0 = S.link + sp_tri(maxm); S.Lw = nullptr; }
  else {
    S.A = p; p += sp_tri(maxm);
    if (alias) { S.Lw = S.link; S.x0 = S.link + sp_tri(maxm); }
    else { S.Lw = p; p += sp_tri(maxm); S.x0 = p; p += maxm; }
  }
  S.b = p; p += maxm; S.lo = p; p += maxm; S.hi = p; p += maxm; S.x = p; p += maxm;
  // reg_lcp (register LCP solver: no pairs, no extras, no free root, no report, no SPD -- SpatialImplT::choose_lds) leaves out what only
  // those paths touch: the LDS solver's iterate (r), the previous step's constraint forces (cf), the free root's pose (root).  712 B in
  // fp64 for HumanWalker: 26 928 B instead of 27 640 = six workgroups per CU instead of five (round 4).
  if (reg_lcp) S.r = nullptr; else { S.r = p; p += maxm; }
  S.cpP = p; p += maxcp * 4;
  S.cpN = p; p += maxcp * 3;
  S.misc = p; p += 16;
  S.sinv = p; p += sp_npad(n);   // the factorisation writes the padding columns' entries too
  if (reg_lcp) { S.cf = nullptr; S.root = nullptr; } else { S.cf = p; p += n; S.root = p; p += 24; }
  S.rdof = (int*)p; p += maxm * sizeof(int) / sizeof(Real) + 1;
  S.rfidx = (int*)p; p += maxm * sizeof(int) / sizeof(Real) + 1;
  S.cplink = (int*)p; p += maxcp * sizeof(int) / sizeof(Real) + 1;
  S.cplinkB = (int*)p; p += maxcp * sizeof(int) / sizeof(Real) + 1;
  S.imisc = (int*)p;
  S.topo = S.imisc + 8;
  S.ancd = S.topo + nl;
  S.ticks = (unsigned long long*)(((size_t)(S.ancd + nl) + 7) & ~(size_t)7);
  return S;
}
__host__ __device__ inline size_t sp_lds_bytes(int nl, int n, size_t real_bytes, int maxm, int maxcp, int reg_lcp, int hreals = 0) {
  const bool alias = sp_lw_aliases_links(nl, maxm);
  const size_t lw = alias ? 0 : (size_t)sp_tri(maxm) + maxm;
  const size_t a = (reg_lcp && alias) ? 0 : (size_t)sp_tri(maxm);
  size_t reals = (size_t)sp_link_reals(nl, maxm) + (reg_lcp ? 4 : 5) * n + sp_npad(n) + (size_t)sp_h_reals(n, hreals) + 3 + (size_t)sp_w_reals(n, maxm, nl) + a + lw +
                 (reg_lcp ? 4 : 5) * maxm + maxcp * 7 + 16 + (reg_lcp ? 0 : 24);
  return reals * real_bytes + (2 * maxm + 2 * maxcp + 8 + 2 * nl) * sizeof(int) + 4 * real_bytes + 16 + 10 * sizeof(unsigned long long);
}

}  // namespace dartk
