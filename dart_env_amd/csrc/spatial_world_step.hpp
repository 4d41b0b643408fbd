// spatial_world_step.hpp -- one DART world step of the env owned by a wavefront: dynamics, contact / limit / friction rows, two-stage LCP, velocity and position update, contact report.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include <type_traits>
#include "spatial_dense.hpp"
#include "spatial_free_root.hpp"
#include "spatial_box_box.hpp"

namespace dartk {

// ------------------------------------------------------------------ one world step for the env owned by this wavefront
// 1: mass-matrix entries and Jacobian rows are assembled entry by entry on all 64 lanes (round 4); 0: one lane per row walking its
// ancestors (rounds 1-3; kept for A/B builds: tools/build_variant.sh <name> spatial_f64 -DSP_ENTRY_PARALLEL=0)
#ifndef SP_ENTRY_PARALLEL
#define SP_ENTRY_PARALLEL 1
#endif
// 1: friction rows start sticking or sliding according to the frictionless solve (round 4); 0: all of them start free (rounds 1-3)
#ifndef SP_FRICTION_START
#define SP_FRICTION_START 1
#endif
#ifndef SP_BAKE_DIMS
#define SP_BAKE_DIMS 1
#endif
#define SP_TICK(ph)                                                                              \
  do {                                                                                            \
    if (Md.stats && lane == 0) {                                                                  \
      const unsigned long long t1_ = __builtin_readcyclecounter();                                \
      S.ticks[ph] += t1_ - t0_;                                                                   \
      t0_ = t1_;                                                                                  \
    }                                                                                             \
  } while (0)

// BIG: register-resident LCP solver and back-substitution (the 20+-dof models).
// PAIRS: link-link contacts (box pairs, general contact normals); EXTRAS: snake fluid forces, external body force, Coulomb
// joint friction rows.  Models that need neither run the lean instantiation (HumanWalker: 8 % faster than the full one).
// PAT: compile-time sparsity of the mass-matrix factor (tree_patterns.hpp; DensePattern = any model).
template <class Real, bool PAIRS, bool EXTRAS, bool REPORT = false, bool BIG = false, class PAT = DensePattern>
__device__ __forceinline__ void sp_world_step(const SpatialModel<Real>& Md, const LinkConst<Real>& lc, SpLds<Real>& S, int lane,
                                              int* contact_flags, int64_t env, bool report = false) {
  // env: the env this wavefront steps -- NOT blockIdx.x once a launch order is in force (sp_step_kernel: e = sched_perm[blockIdx.x]);
  // every per-env buffer (external force, contact report, constraint forces, debug dump) is indexed with it
  // BK: a pattern kernel takes the model's dimensions from its pattern at compile time (tree_patterns.hpp; the library launches it
  // only for a model that carries exactly these values) -- the LDS carve offsets, row capacities and loop bounds become immediates
  constexpr bool BK = (SP_BAKE_DIMS != 0) && !PAT::dense;
  const int n = BK ? PAT::n : Md.n, nl = BK ? PAT::nl : Md.nl;
  const int maxm = BK ? PAT::maxm : Md.maxm, maxcp = BK ? PAT::maxcp : Md.maxcp, nshapes = BK ? PAT::nshapes : Md.nshapes;
  const bool impulse_M = BK ? (PAT::impulse_M != 0) : (Md.impulse_M != 0);
  unsigned long long t0_ = Md.stats ? __builtin_readcyclecounter() : 0ull;
  {
    // A world that has left the representable regime is frozen: a coordinate or velocity beyond 1e6 (or not finite) cannot come
    // back under the tasks' validity bound |s| < 100 within an env-step (hopper.py:60-62 and every other task's done condition:
    // the env reports done either way), but its LCPs -- rows with 1e10-sized right-hand sides -- would run the pivoting solver
    // into its cap and the PGS safety net in every remaining world step: milliseconds for one wave that the whole launch then
    // waits for (HumanWalker at 16 384 envs: 5 such envs in 25 launches took the average from 6.4 to 8.3 ms).  The oracle
    // freezes the same way (oracle_step).
    const bool gone = lane < n && !(fabs(S.q[lane]) < Real(1e6) && fabs(S.dq[lane]) < Real(1e6));
    if (__any(gone)) {   // wave-uniform: this wavefront owns the env
      if (REPORT && report) {   // a frozen world has no contacts and no constraint forces: do not leave the previous step's in the report
        if (lane == 0) Md.creport_count[env] = 0;
        if (lane < n) Md.cf_report[(size_t)env * n + lane] = Real(0);
      }
      return;
    }
  }
  if (EXTRAS && Md.free_root) {
    if (lane == 0) sp_free_root_to_internal<Real>(S);
    __syncthreads();
  }
  // tree recursions level by level: links of equal depth are independent, one lane each
  if (lane == 0) sp_root_offset<Real>(Md, S);
  sp_forward<Real, EXTRAS, false, typename std::conditional<BK, PAT, DensePattern>::type>(lc, Md, S, lane, env);   // lane i owns link i
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
  using HP = typename std::conditional<BK, PAT, DensePattern>::type;   // layout of S.H: a pattern's skyline needs its compile-time dimensions (BK)
  if constexpr (!HP::dense) { for (int e = lane; e < HP::hreals; e += 64) S.H[e] = Real(0); }   // skyline: structural entries only, no padding rows
  else if (lane < sp_npad(n)) { for (int k = 0; k < lane; k++) S.H[HL(lane, k)] = Real(0); if (lane >= n) S.H[HL(lane, lane)] = Real(1); }
  __syncthreads();
  // Entry-parallel assembly of M and of the Jacobian rows (round 4) for the 20+-dof models (BIG): measured on one box against the
  // ancestor walks (-DSP_ENTRY_PARALLEL=0), fp64, 16 384 envs: HumanWalker 16.43 -> 15.72 ms, Walker3d 9.85 -> 9.46 ms; the Dog
  // (22 dofs, shallow legs, up to 40 box-vertex rows, LDS solver) got 4 % SLOWER with it (4.31 -> 4.50 ms) and keeps the walks.
  constexpr bool EP = (SP_ENTRY_PARALLEL != 0) && BIG;
  if constexpr (EP) {
    sp_mass_entries<Real, typename std::conditional<BK, PAT, DensePattern>::type>(Md, S, lane);   // all 64 lanes, one structural entry at a time (reversed storage order, see sp_mass_row)
    __syncthreads();
    if (!impulse_M) {   // (wave-uniform) A3 knob at 0: the impulse pass runs on M + E
      if (lane < n) S.H[HP::dense ? HL(n - 1 - lane, n - 1 - lane) : lc.h_diag] += lc.d_diag;
      __syncthreads();
    }
  } else {
    if (lane < n) sp_mass_row<Real>(lc, Md, S, lane, !impulse_M);   // scatters into other lanes' rows (reversed storage order, see there)
    __syncthreads();
  }
  if (EXTRAS && Md.task == 12) sp_spd_torque<Real>(lc, Md, S, lane);
  SP_TICK(1);
  // ---- contact points and active limits, in parallel: lane s tests collision shape s, lane d tests the limits of
  // dof d; ballots give every hit its slot (shape order, then vertex order -- the serial order of the oracle)
  const V3<Real> roff = ld3(S.misc);
  int ncp, m, m1;   // contact points, LCP rows, rows of the frictionless stage (normals + limits + joint friction: a prefix)
  constexpr bool PREFIX = !PAIRS;
  {
    const bool has_shape = lane < nshapes;
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
        if (idx < maxcp) {
          S.cpP[4 * idx + 0] = P[v].x; S.cpP[4 * idx + 1] = P[v].y; S.cpP[4 * idx + 2] = P[v].z; S.cpP[4 * idx + 3] = dep[v];
          S.cpN[3 * idx + 0] = Real(0); S.cpN[3 * idx + 1] = Real(1); S.cpN[3 * idx + 2] = Real(0);
          S.cplink[idx] = slink; S.cplinkB[idx] = -1;
        }
        idx++;
      }
    }
    ncp = total < maxcp ? total : maxcp;
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
        if (id2 < maxcp) {
          for (int t = 0; t < 4; t++) S.cpP[4 * id2 + t] = scratch[4 * v + t];
          S.cpN[3 * id2 + 0] = -nrm.x; S.cpN[3 * id2 + 1] = -nrm.y; S.cpN[3 * id2 + 2] = -nrm.z;   // into the first link
          S.cplink[id2] = la; S.cplinkB[id2] = lb;
        }
      }
      ncp = (ncp + total) < maxcp ? (ncp + total) : maxcp;
    }
    // Row order: contact normals [0, ncp), joint limits, joint friction, then the 2 ncp contact tangents.  The rows the
    // frictionless stage can move are a PREFIX (m1 of them): that stage solves the leading m1 x m1 block of A alone (the tangent
    // rows are pinned at 0 there and contribute nothing), usually in a smaller solver variant.
    // (PREFIX = false, the link-link kernels: rows stay interleaved {n, t1, t2} per contact with the limits behind them -- the prefix
    // order made Walker3d 8 % slower.)
    if (PREFIX) { if (lane < ncp) { S.rdof[lane] = -1; S.rfidx[lane] = -1; } }
    else if (lane < 3 * ncp) { S.rdof[lane] = -1; S.rfidx[lane] = (lane % 3 == 0) ? -1 : (lane - lane % 3); }
    const int nfront = PREFIX ? ncp : 3 * ncp;   // contact rows in front of the limit rows
    // joint-limit rows
    const Real qd = lane < n ? S.q[lane] : Real(0);
    const bool low = lane < n && lc.d_limited && qd <= lc.d_lower;
    const bool up = lane < n && lc.d_limited && !low && qd >= lc.d_upper;
    const uint64_t lm = __ballot(low || up);
    const int row = nfront + __popcll(lm & lt);
    if ((low || up) && row < maxm) {
      const Real viol = low ? qd - lc.d_lower : qd - lc.d_upper;
      const Real bounce = fmin(fmax(-viol * Md.limit_erp_dt, -Md.max_erv), Md.max_erv);
      S.rdof[row] = lane; S.rfidx[row] = -1;
      S.b[row] = bounce;   // (- v*_d follows with the Jacobian rows, once the forward dynamics has run)
      S.lo[row] = low ? Real(0) : -inf_<Real>();
      S.hi[row] = low ? inf_<Real>() : Real(0);
    }
    m = nfront + __popcll(lm);
    if (EXTRAS && Md.has_joint_friction) {   // DART JointCoulombFrictionConstraint rows: joint velocity -> 0, impulse within +-mu dt
      const bool fr = lane < n && lc.d_fric > Real(0);
      const uint64_t fm = __ballot(fr);
      const int frow = m + __popcll(fm & lt);
      if (fr && frow < maxm) {
        S.rdof[frow] = lane; S.rfidx[frow] = -1;
        S.b[frow] = Real(0);   // (- v*_d follows with the Jacobian rows)
        S.lo[frow] = -lc.d_fric; S.hi[frow] = lc.d_fric;
      }
      m += __popcll(fm);
    }
    m1 = m < maxm ? m : maxm;
    if (PREFIX) {
      if (lane < 2 * ncp && m1 + lane < maxm) { S.rdof[m1 + lane] = -1; S.rfidx[m1 + lane] = lane >> 1; }   // tangents t1, t2 of contact lane / 2
      m = m1 + 2 * ncp;
    }
    m = m < maxm ? m : maxm;
    if (!PREFIX) m1 = m;
  }
  __syncthreads();
  SP_TICK(3);
  {
    // ---- Two passes through ONE copy of the factorisation + forward-substitution code (a second inlined copy thrashed the
    // instruction cache: HumanWalker 5.0 -> 8.0 ms instead of the ~6.3 the extra arithmetic is worth):
    //   pass 0  forward dynamics: M + E (E = dt D + dt^2 K, the implicit damping / spring terms) is factored in the still idle
    //           Jacobian block, the right-hand side (one entry per lane) is substituted column by column while the factor is in
    //           registers (sp_cholesky_t, xvec) -> qdd; S.dq becomes the unconstrained velocity v* = dq + dt qdd, which is all
    //           the constraint phase needs.
    //   pass 1  impulse inertia: Jacobian rows (lane per row; b = bounce - J v*), S.H is factored -- M alone under DART 6's rule
    //           (A3, card.impulse_inertia = 1), M + E with the knob at 0 -- and W = L^-1 J^T.
    // columns of W follow the factor's storage order: dof d sits in column n-1-d
    const int np = sp_npad(n);
    Real* const H2 = S.W; Real* const sinv2 = S.W + HR(np); Real* const xq = sinv2 + np;
#pragma nounroll
    for (int pass = 0; pass < Md.fd_passes; pass++) {   // (a run-time bound: the compiler must keep this a loop)
      Real* const Hm = pass ? S.H : H2; Real* const sv = pass ? S.sinv : sinv2; Real* const Wr = pass ? S.W : xq;
      const int nrows = pass ? m : 1;
      if (pass == 0) {
        for (int e = lane; e < (HP::dense ? HR(np) : HP::hreals); e += 64) H2[e] = S.H[e];   // the whole padded block, 64 entries per trip (was: lane r copying its row, 32 dependent trips for the last one)
        // (factoring M + E straight from S.H -- row loads from S.H plus the E entry, no copy, no barrier -- was built and measured in
        // round 4: SLOWER, HumanWalker fp64 13.45 -> 13.81 ms, fp32 5.98 -> 6.36, Dog 3.41 -> 3.59; profiles/r04_tree_kernel_ab.txt)
        __syncthreads();
        if (lane < n) { if (impulse_M) H2[HP::dense ? HL(n - 1 - lane, n - 1 - lane) : lc.h_diag] += lc.d_diag; xq[n - 1 - lane] = S.rhs[lane]; }
      } else if constexpr (EP) {
        // Entry-parallel Jacobian (round 4).  Lane (d, half) = (lane & 31, lane >> 5) holds dof d's joint axis / origin / type in
        // registers and fills column d of the rows half, half + 2, ...: J_id = dir_i . (a_d x (P_i - o_d)) (revolute) or dir_i . a_d
        // (prismatic) where dof d moves the contact's link (bit d of the link's ancestor-dof mask), 0 elsewhere -- no walk up the tree,
        // every entry from reads that depend on nothing but (i, d), 58 of 64 lanes busy for HumanWalker.  (Before: one lane per row
        // chasing parent pointers through LDS, <= 12 dependent hops: 11 % of the fp64 kernel's cycles.)
        // Two steps so that no lane chases a chain of dependent LDS reads inside the entry loop: (1) the lane that owns row i writes
        // the row's descriptor -- direction, point, the ancestor-dof masks of its link(s), or the dof of a limit row -- into the idle
        // Delassus block (S.A; it aliases the link records, which every lane has read its dof's axis / origin from before the
        // barrier); (2) the entry loop reads descriptors only (addresses depend on i alone), two rows per trip.
        {
          const int dcol = lane & 31, half = lane >> 5;
          const int dlk = __shfl(lc.d_link, dcol);          // link of dof dcol (lane dcol < n owns it)
          const bool dlive = dcol < n;
          const Real* Ld = S.link + (dlive ? dlk : 0) * SP_LINKF;
          const V3<Real> ad = ld3(Ld + LK_A), od = ld3(Ld + LK_JO);
          const bool drev = topo_jtype(S.topo[dlive ? dlk : 0]) == 2;
          const uint32_t dbit = 1u << dcol;
          Real* const rowd = S.A;                            // [m][8]: dir (3), P (3), then two ints in the last two slots
          int* const rowi = (int*)(S.A + 8 * maxm);       // [m][4]: limit dof (-1: contact row), mask a, mask b
          __syncthreads();                                   // every lane holds its dof's axis / origin: the link records may go
          if (lane < m) {
            const int i = lane, rd = S.rdof[i];
            V3<Real> dir = v3<Real>(0, 0, 0), P = dir;
            int ma = 0, mb = 0;
            if (rd < 0) {
              const int cidx = PREFIX ? (i < m1 ? i : ((i - m1) >> 1)) : i / 3, kind = PREFIX ? (i < m1 ? 0 : 1 + ((i - m1) & 1)) : i % 3;
              // DART ContactConstraint tangent basis: t1 = normalize(z x n) (x x n when z and n are parallel), t2 = n x t1
              if (PAIRS) {
                const V3<Real> nn = ld3(S.cpN + 3 * cidx);
                V3<Real> t1 = cross(v3<Real>(0, 0, 1), nn);
                if (dot(t1, t1) < Real(1e-12)) t1 = cross(v3<Real>(1, 0, 0), nn);
                t1 = t1 * (Real(1) / sqrt(dot(t1, t1)));
                dir = kind == 0 ? nn : (kind == 1 ? t1 : cross(nn, t1));
              } else {   // ground contacts only: n = +y, t1 = z x n = -x, t2 = n x t1 = +z
                dir = kind == 0 ? v3<Real>(0, 1, 0) : (kind == 1 ? v3<Real>(-1, 0, 0) : v3<Real>(0, 0, 1));
              }
              P = ld3(S.cpP + 4 * cidx);
              ma = S.ancd[S.cplink[cidx]];
              if (PAIRS) { const int lb = S.cplinkB[cidx]; mb = lb >= 0 ? S.ancd[lb] : 0; }   // J = J_a - J_b for a link-link contact
            }
            st3(rowd + 8 * i, dir); st3(rowd + 8 * i + 3, P);
            rowi[4 * i] = rd; rowi[4 * i + 1] = ma; rowi[4 * i + 2] = mb;
          }
          __syncthreads();
          auto entry = [&](int i) -> Real {
            const int rd = rowi[4 * i];
            const uint32_t ma = (uint32_t)rowi[4 * i + 1], mb = (uint32_t)rowi[4 * i + 2];
            const V3<Real> dir = ld3(rowd + 8 * i), P = ld3(rowd + 8 * i + 3);
            const Real jd = drev ? dot(dir, cross(ad, P - od)) : dot(dir, ad);
            Real v = (ma & dbit) ? jd : Real(0);
            if (PAIRS) v = (mb & dbit) ? v - jd : v;
            return rd >= 0 ? ((rd == dcol) ? Real(1) : Real(0)) : v;   // limit / joint-friction row: the unit vector of its dof
          };
          int i = half;
          for (; i + 2 < m; i += 4) {   // rows i and i + 2 of this half: their reads are independent
            const Real v0 = entry(i), v1 = entry(i + 2);
            if (dlive) { S.W[i * n + (n - 1 - dcol)] = v0; S.W[(i + 2) * n + (n - 1 - dcol)] = v1; }
          }
          if (i < m) { const Real v0 = entry(i); if (dlive) S.W[i * n + (n - 1 - dcol)] = v0; }
        }
        __syncthreads();
        if (lane < m) {   // right-hand side and bounds of row `lane`: b = bounce - J v*
          const Real* Jr = S.W + lane * n;
          const int d = S.rdof[lane];
          if (d >= 0) {
            S.b[lane] -= S.dq[d];   // limit / joint-friction row: bounce - v*_d
          } else {
            const int cidx = PREFIX ? (lane < m1 ? lane : ((lane - m1) >> 1)) : lane / 3, kind = PREFIX ? (lane < m1 ? 0 : 1 + ((lane - m1) & 1)) : lane % 3;
            Real rel = Real(0);
            for (int k = 0; k < n; k++) rel += Jr[k] * S.dq[n - 1 - k];
            const Real depth = S.cpP[4 * cidx + 3];
            S.b[lane] = (kind == 0 ? fmin(depth * Md.erp_dt, Md.max_erv) : Real(0)) - rel;
            S.lo[lane] = Real(0);
            S.hi[lane] = kind == 0 ? inf_<Real>() : Real(0);   // friction rows pinned during the frictionless stage
          }
        }
      } else {
        if (lane < m) {
          Real* Jr = S.W + lane * n;
          for (int k = 0; k < n; k++) Jr[k] = Real(0);
          const int d = S.rdof[lane];
          if (d >= 0) {
            Jr[n - 1 - d] = Real(1);
            S.b[lane] -= S.dq[d];   // limit / joint-friction row: bounce - v*_d
          } else {
            const int cidx = PREFIX ? (lane < m1 ? lane : ((lane - m1) >> 1)) : lane / 3, kind = PREFIX ? (lane < m1 ? 0 : 1 + ((lane - m1) & 1)) : lane % 3;
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
                Jr[n - 1 - dj] += v;
                rel += v * S.dq[dj];
              }
            }
            const Real depth = S.cpP[4 * cidx + 3];
            S.b[lane] = (kind == 0 ? fmin(depth * Md.erp_dt, Md.max_erv) : Real(0)) - rel;
            S.lo[lane] = Real(0);
            S.hi[lane] = kind == 0 ? inf_<Real>() : Real(0);   // friction rows pinned during the frictionless stage
          }
        }
      }
      __syncthreads();
      SP_TICK(4);
      if constexpr (!PAT::dense) {
        // Pattern kernels (HumanWalker): the substitution follows while the factor is still in registers -- L_kj reaches a
        // row's lane through v_readlane (sp_cholesky_t).  Measured 6.36 -> 6.19 ms; the dense kernels got slower with it
        // (Walker3d +4 %) and keep the LDS reads.
        static_assert(PAT::dense || BK, "a pattern kernel stores H as a skyline: it needs SP_BAKE_DIMS");
        sp_cholesky<Real, PAT>(Hm, sv, n, lane, pass ? Wr : nullptr, nrows - 1, pass ? nullptr : xq, lc.h_rb, lc.h_mask);
      } else {
        sp_cholesky<Real, PAT>(Hm, sv, n, lane, nullptr, -1, pass ? nullptr : xq);
        // The row lives in registers and both loops are fully unrolled (SP_MAXN x SP_MAXN / 2 predicated steps, uniform
        // `k < n` branches): every factor entry is one LDS read at an immediate offset, no index arithmetic.
        if (pass == 1 && lane < nrows) {
          Real* yrow = Wr + lane * n;
          Real y[SP_MAXN];
  #pragma unroll
          for (int k = 0; k < SP_MAXN; k++) y[k] = (k < n) ? yrow[k] : Real(0);
          // factor rows start 16-byte aligned and are padded to multiples of 4 entries (HR): 128-bit LDS loads, 4 (fp32) or 2 (fp64)
          // entries each; the entries a load brings in beyond column k - 1 are not used
          using Vec = typename sp_vec128<Real>::type;
          constexpr int VW = sp_vec128<Real>::width;
  #pragma unroll
          for (int k = 0; k < SP_MAXN; k++) {
            if (k < n) {
              Real t = y[k];
              const Vec* hr = reinterpret_cast<const Vec*>(Hm + HR(k));
  #pragma unroll
              for (int j = 0; j < k; j += VW) {
                const Vec h = hr[j / VW];
                const Real* hv = reinterpret_cast<const Real*>(&h);
  #pragma unroll
                for (int c = 0; c < VW; c++) if (j + c < k) t -= hv[c] * y[j + c];
              }
              y[k] = t * sv[k];
            }
          }
  #pragma unroll
          for (int k = 0; k < SP_MAXN; k++) if (k < n) yrow[k] = y[k];
        }
        __syncthreads();
      }
      SP_TICK(2);
      if (pass == 0) {
        sp_chol_backsolve<Real, BIG, BK ? 1 : 0, HP>(Hm, sv, n, xq, lane);
        if (lane < n) S.dq[lane] += Md.dt * xq[n - 1 - lane];
        __syncthreads();
      }
    }
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
      const Real bm = wave_max_nonneg<Real>(lane < m ? fabs(S.b[lane]) : Real(0));
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
        bool isf = false, pinned = false, slide_up = false, slide_dn = false;
        if (lane < m && S.rfidx[lane] >= 0) {
          const int nr = S.rfidx[lane];   // the contact's normal row
          const Real hb = fabs(Md.mu * S.x[nr]);
          // a direction the skeleton cannot move in (planar model, z tangent) has A_ii = 0: keep that row out
          const Real att0 = S.A[TI(lane, lane)];
          isf = true; pinned = !(hb > Real(0)) || !(att0 > Real(1e-12));
          S.hi[lane] = hb; S.lo[lane] = -hb;
#if SP_FRICTION_START
          // Start set of the friction row (round 4; the lane kernels' rule, planar_kernel.hpp): the impulse that would stop the
          // tangential velocity the frictionless solve left, with every other row held and the contact's own normal row free to respond
          // (tangential stiffness = Schur complement A_tt - A_tn^2 / A_nn) -- inside the bounds the row starts free (sticking), beyond
          // them on that bound (sliding).  Before, every friction row started free and a sliding foot cost the wave extra pivoting
          // iterations of ~10 k cycles each.  Only the path to the (unique) LCP solution changes.
          Real wt = -S.b[lane];
          for (int j = 0; j < m; j++) wt += S.A[lane >= j ? TI(lane, j) : TI(j, lane)] * S.x[j];   // x = 0 on the friction rows
          const Real atn = S.A[lane >= nr ? TI(lane, nr) : TI(nr, lane)];
          Real att = ((F >> nr) & 1ull) ? att0 - atn * atn / S.A[TI(nr, nr)] : att0;
          if (!(att > Real(1e-12) * att0)) att = att0;   // a tangent (nearly) dependent on its normal row: the Schur complement rounds to <= 0 -- start from the row's own stiffness
          const Real xe = -wt / att;
          slide_up = !pinned && xe > hb; slide_dn = !pinned && xe < -hb;
#endif
        }
        __syncthreads();
        const uint64_t fr = __ballot(isf), pf = __ballot(isf && pinned), su = __ballot(slide_up), sd = __ballot(slide_dn);
        pinmask = (pinmask & ~fr) | pf;
        F = (F & ~fr) | (fr & ~pf & ~su & ~sd);
        U = (U & ~fr) | su;
      }
      sp_blcp<Real, BIG, BK ? 1 : 0>(S, stage == 0 ? m1 : m, pinmask, F, U, Md.solver_iters, Md.pgs_fallback_sweeps, Md.stats, lane, stage == 0 && !(EXTRAS && Md.has_joint_friction),
                         0, (PREFIX && stage == 1) ? ncp : -1, m1);
    }
    SP_TICK(8);
    if (Md.dbg) {
      double* D = Md.dbg + (size_t)env * 160;
      if (lane == 0) { D[0] = m; D[1] = ncp; }
      if (lane < m) { D[2 + lane] = (double)S.x[lane]; D[42 + lane] = (double)S.b[lane]; D[82 + lane] = (double)S.hi[lane]; D[122 + lane] = (double)S.A[TI(lane, lane)]; }
    }
  }
  if (REPORT && report) {   // world.collision_result.contacts (walker2d.py:38-41, human_walker.py:97-106): point, force on the first body
    if (lane == 0) Md.creport_count[env] = ncp;
    if (lane < ncp) {
      Real* out = Md.creport + ((size_t)env * maxcp + lane) * 8;
      const V3<Real> nn = ld3(S.cpN + 3 * lane);
      V3<Real> t1 = cross(v3<Real>(0, 0, 1), nn);
      if (dot(t1, t1) < Real(1e-12)) t1 = cross(v3<Real>(1, 0, 0), nn);
      t1 = t1 * (Real(1) / sqrt(dot(t1, t1)));
      const V3<Real> t2 = cross(nn, t1);
      // a tangent the skeleton cannot move along (planar model: z) has A_ii = 0 and stays pinned at a bound: no force
      const int r0 = PREFIX ? lane : 3 * lane, r1 = PREFIX ? m1 + 2 * lane : 3 * lane + 1, r2 = r1 + 1;
      const Real l0 = S.x[r0], l1 = (r1 < m && S.A[TI(r1, r1)] > Real(1e-12)) ? S.x[r1] : Real(0),   // (rows beyond the capacity were dropped)
                 l2 = (r2 < m && S.A[TI(r2, r2)] > Real(1e-12)) ? S.x[r2] : Real(0), idt = Real(1) / Md.dt;
      const int lb = S.cplinkB[lane];
      out[0] = (Real)Md.link_body[S.cplink[lane]]; out[1] = lb >= 0 ? (Real)Md.link_body[lb] : Real(-1);
      st3(out + 2, ld3(S.cpP + 4 * lane) + roff);
      st3(out + 5, (nn * l0 + t1 * l1 + t2 * l2) * idt);
    }
  }
  // ---- new velocity: v = v* + L^-T (W^T lambda)   (storage order until the back-substitution is done)
  if (lane < n) {
    Real u = Real(0);
    for (int i = 0; i < m; i++) u += S.W[i * n + lane] * S.x[i];
    if ((EXTRAS && Md.task == 12) || (REPORT && report)) S.lo[lane] = u;   // W^T lambda = L^-1 J^T lambda
    S.rhs[lane] = u;
  }
  __syncthreads();
  if ((EXTRAS && Md.task == 12) || (REPORT && report)) {   // constraint_forces() of this step: J^T lambda / dt = L (W^T lambda) / dt
    if (lane < n) {
      Real t = Real(0);
      for (int k = 0; k <= lane; k++) t += S.H[HL(lane, k)] * S.lo[k];
      S.cf[n - 1 - lane] = t / Md.dt;
    }
    __syncthreads();
    if (REPORT && report && lane < n) {
      Real v = S.cf[lane];
      if (EXTRAS && Md.free_root && lane < 6) {   // internal root coordinates are world-frame, DART's body-frame: tau_b = R^T tau_w
        const int g = lane < 3 ? 0 : 3, a = lane - g;
        v = S.root[a] * S.cf[g] + S.root[3 + a] * S.cf[g + 1] + S.root[6 + a] * S.cf[g + 2];
      }
      Md.cf_report[(size_t)env * n + lane] = v;
    }
  }
  if (m > 0) sp_chol_backsolve<Real, BIG, BK ? 1 : 0, HP>(S.H, S.sinv, n, S.rhs, lane);   // (wave-uniform: nothing touching, no limit active -> v = v*)
  SP_TICK(9);
  if (lane < n) { const Real vnew = S.dq[lane] + S.rhs[n - 1 - lane]; S.dq[lane] = vnew; S.q[lane] += Md.dt * vnew; }
  __syncthreads();
  if (EXTRAS && Md.free_root) {   // the six root entries just advanced are placeholders: the pose lives in S.root
    if (lane == 0) sp_free_root_advance<Real>(S, Md.dt);
    __syncthreads();
  }
  (void)nl;
}

}  // namespace dartk
