// spatial_dense.hpp -- dense wave-parallel pieces of the tree kernel: systolic register Cholesky, triangular solves, SPD controller torque, the boxed-LCP solver (block principal pivoting + PGS).
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include "spatial_dynamics.hpp"
#include "tree_patterns.hpp"

// A wavefront executes in lock-step: between two barriers, an instruction of every lane completes before the next one starts, so "all
// lanes read x, then one lane overwrites it" needs no barrier on the device.  The fiber emulation of tests/kernel_emu/fake_wave_include
// runs a lane alone from one cross-lane operation to the next and defines this as a rendezvous; the device build defines it as nothing.
#ifndef DART_LOCKSTEP_FENCE
#define DART_LOCKSTEP_FENCE() ((void)0)
#endif

namespace dartk {

// ------------------------------------------------------------------ wave-parallel dense kernels (row-owner scheme)

// Cholesky of the lower triangle M (n <= 32, row layout HR / HL, padded to sp_npad(n) with identity rows), run as a
// systolic array over the wave: lane r holds row r in REGISTERS, both loops are fully unrolled, and the finished
// column entry L_kj travels from lane k to everybody through v_readlane (an SGPR operand of the FMA) -- no LDS traffic
// and no barrier inside the factorisation.  Per column j: d_j = readlane(row[j], j); L_rj = row[j] / sqrt(d_j);
// row[k] -= L_rj L_kj for k > j.  Updates beyond a lane's diagonal are garbage that nothing reads (kept finite by the
// identity padding).  The factor is written back to LDS once at the end, with sinv[j] = 1 / L_jj.
// PAT (tree_patterns.hpp): compile-time sparsity of the factor -- the update of row k by column j is skipped where L_kj is
// structurally zero (the entry stays the exact 0 the zero-filled H holds).
// W != nullptr: the forward substitution W_i <- L^-1 W_i of rows 0..m of W (lane per row, stride n) follows while the factor
// is still in registers: L_kj reaches the row's lane through v_readlane from lane k, so the 220-400 factor entries a row needs
// cost no LDS access at all (first version: ~110 128-bit LDS loads per row, the phase was LDS-latency bound at 2 waves / SIMD).
// xvec != nullptr: ONE right-hand side, entry k held by lane k (storage order), is forward-substituted column by column while the
// factor is in registers -- lane j > k holds L_jk in its own row, x_k arrives through v_readlane: 3 wave instructions per column
// instead of a row's ~400 multiply-adds (the forward dynamics' solve, sp_world_step pass 0).
// Pattern kernels store M / the factor as a SKYLINE (round 5; tree_patterns.hpp): row r keeps only its structural entries -- with the
// dofs numbered leaves first they are one contiguous run of columns ending at the diagonal -- at hrb + k; `hrb` / `hmask` (base offset and
// column mask of row `lane`, LinkConst::h_rb / h_mask) come from the caller, the identity padding rows beyond n are made up in registers.
template <class Real, int NP, class PAT = DensePattern>
__device__ __forceinline__ void sp_cholesky_t(Real* M, Real* sinv, int n, int lane, Real* W = nullptr, int m = -1, Real* xvec = nullptr,
                                              int hrb = 0, uint32_t hmask = 0u) {
  const int r = lane < NP ? lane : 0;   // spare lanes shadow lane 0 (convergent code, results discarded)
  const int rb = PAT::dense ? HR(r) : (lane < NP ? hrb : PAT::hbase(0));
  const uint32_t own = PAT::dense ? 0u : ((lane < PAT::n ? hmask : 0u) | ((lane < PAT::n || lane >= NP) ? (1u << r) : 0u));   // stored entries of this lane's row
  Real row[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) {
    if constexpr (PAT::dense) row[k] = (k <= r) ? M[rb + k] : Real(0);
    else row[k] = ((own >> k) & 1u) ? M[rb + k] : ((k == r) ? Real(1) : Real(0));   // (k == r: a padding row n <= r < NP = identity)
  }
  Real sown = Real(1);                  // lane j: 1 / L_jj
  // Columns beyond the model's dofs (identity padding up to NP) are left alone in a pattern kernel: nothing to factor, sinv = 1.
  constexpr int NC = PAT::dense ? NP : (PAT::n < NP ? PAT::n : NP);
  // 1 / sqrt(d_j) is SOFTWARE-PIPELINED (round 4): column j + 1's diagonal is final after the first update of column j, so its
  // v_rsq + Newton chain (7 dependent fp64 operations, ~100 cycles that a lone in-order wave would wait out) is started there and its
  // links are placed one by one between the remaining updates of column j, which do not depend on it.
  RsqStaged<Real> nx;
  nx.start(readlane_<Real>(row[0], 0));
#pragma unroll
  for (int s = 0; s < RsqStaged<Real>::NSTAGE; s++) nx.step(s);
#pragma unroll
  for (int j = 0; j < NC; j++) {
    const Real sj = nx.r;
    const Real lrj = (r >= j) ? row[j] * sj : Real(0);   // lanes above the diagonal contribute nothing
    row[j] = lrj;
    // (1 / L_jj goes to sinv[] once, after the loop: a store under `if (lane == j)` here splits the factorisation into one basic block
    // per column, and the compiler then SINKS every update row[k] -= L_rj L_kj into the block of column k while the v_readlane that
    // produced L_kj -- convergent, it cannot move -- stays in column j's: the broadcast values of a whole factorisation were live at
    // once, ~450 SGPRs spilled into VGPR lanes and read back, 2 x v_writelane + 2 x v_readlane extra per update; round 4, found in the
    // disassembly of the fp64 pattern kernel)
    sown = (lane == j) ? sj : sown;
    int st = RsqStaged<Real>::NSTAGE;   // (compile-time after unrolling) next link of the chain; NSTAGE: none pending
    if (j + 1 < NC) {
      if (PAT::nz(j + 1, j)) row[j + 1] -= lrj * readlane_<Real>(lrj, j + 1);
      nx.start(readlane_<Real>(row[j + 1], j + 1));
      st = 0;
    }
#pragma unroll
    for (int k = j + 2; k < NP; k++) {
      if (PAT::nz(k, j)) {
        row[k] -= lrj * readlane_<Real>(lrj, k);
        if (st < RsqStaged<Real>::NSTAGE) nx.step(st++);
      }
    }
#pragma unroll
    for (int s = 0; s < RsqStaged<Real>::NSTAGE; s++) if (s >= st) nx.step(s);
  }
  if (lane < NP) sinv[lane] = sown;   // (sinv has sp_npad(n) slots: the padding columns write theirs too)
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if constexpr (PAT::dense) { if (k <= lane) M[rb + k] = row[k]; }
      else { if ((own >> k) & 1u) M[rb + k] = row[k]; }
    }
  }
  if (xvec != nullptr) {   // wave-uniform
    Real yv = (lane < n) ? xvec[lane] : Real(0);
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if (k < n) {
        const Real xk = readlane_<Real>(yv * sown, k);   // x_k = y_k / L_kk, final once columns 0..k-1 have been applied
        yv = (lane == k) ? xk : ((lane > k && lane < NP) ? yv - row[k] * xk : yv);
      }
    }
    if (lane < n) xvec[lane] = yv;
  }
  if (W != nullptr) {   // wave-uniform
    // the factor entries are broadcast AGAIN below, on purpose: see DART_OPAQUE (wave_blcp.hpp)
#pragma unroll
    for (int k = 0; k < NP; k++) DART_OPAQUE(row[k]);
    DART_OPAQUE(sown);
    const bool has = lane <= m;
    Real* yrow = W + (has ? lane : 0) * n;
    Real y[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) y[k] = (k < n) ? yrow[k] : Real(0);
#pragma unroll
    for (int k = 0; k < NP; k++) {
      if (k < n) {
        Real t = y[k];
        // the broadcasts of step k may not be issued before step k - 1 is done (DART_TIE, wave_blcp.hpp): <= 12 SGPR pairs in flight
#pragma unroll
        for (int j = 0; j < k; j++) if (PAT::nz(k, j)) { if (k > 0) DART_TIE(row[j], y[k - 1]); t -= readlane_<Real>(row[j], k) * y[j]; }
        if (k > 0) DART_TIE(sown, y[k - 1]);
        y[k] = t * readlane_<Real>(sown, k);
      }
    }
    if (has) {
#pragma unroll
      for (int k = 0; k < NP; k++) if (k < n) yrow[k] = y[k];
    }
  }
  __syncthreads();
}
// one straight-line variant per padded size (only the one a model uses ever enters the instruction cache)
template <class Real, class PAT = DensePattern>
__device__ __forceinline__ void sp_cholesky(Real* M, Real* sinv, int n, int lane, Real* W = nullptr, int m = -1, Real* xvec = nullptr,
                                            int hrb = 0, uint32_t hmask = 0u) {
  if constexpr (!PAT::dense) { sp_cholesky_t<Real, sp_npad(PAT::n), PAT>(M, sinv, n, lane, W, m, xvec, hrb, hmask); return; }
  const int np = sp_npad(n);
  if (np <= 8) sp_cholesky_t<Real, 8>(M, sinv, n, lane, W, m, xvec);
  else if (np <= 16) sp_cholesky_t<Real, 16>(M, sinv, n, lane, W, m, xvec);
  else if (np <= 24) sp_cholesky_t<Real, 24>(M, sinv, n, lane, W, m, xvec);
  else sp_cholesky_t<Real, 32>(M, sinv, n, lane, W, m, xvec);
}
// x <- L^-T x (backward) for one vector in LDS, column-oriented, lanes own entries
template <class Real>
__device__ __forceinline__ void sp_chol_backsolve_lds(const Real* Lf, const Real* sinv, int n, Real* x, int lane) {
  for (int j = n - 1; j >= 0; j--) {
    __syncthreads();
    const Real xj = x[j] * sinv[j];
    DART_LOCKSTEP_FENCE();   // every lane has read x[j] before lane j overwrites it
    if (lane == j) x[j] = xj;
    if (lane < j) x[lane] -= Lf[HL(j, lane)] * xj;
  }
  __syncthreads();
}

// x <- L^-T x (backward) for one vector in LDS, lanes own entries.  The running vector stays in registers: column j's finished
// entry is broadcast with v_readlane, the factor entries L_ji (row j, contiguous over the lanes i) are independent LDS reads
// issued up front -- no barrier and no LDS write per column (the first version paid both).
template <class Real, int NP, int TAG = 0, class PAT = DensePattern>   // TAG: private copies per kernel family (see sp_blcp_t); PAT: skyline storage of the factor
static __device__ __attribute__((noinline)) void sp_chol_backsolve_t(const Real* __restrict__ Lf_, const Real* __restrict__ sinv_, int n, Real* __restrict__ x_, int lane) {
  const auto Lf = DART_LDS_PTR(const Real, Lf_), sinv = DART_LDS_PTR(const Real, sinv_);   // LDS with every caller (wave_blcp.hpp)
  const auto x = DART_LDS_PTR(Real, x_);
  const int i = lane < NP ? lane : 0;
  Real xi = (lane < n) ? x[lane] : Real(0);
  const Real si = (lane < n) ? sinv[lane] : Real(1);
  Real col[NP];   // col[j] = L_ji for j > i
#pragma unroll
  for (int j = 0; j < NP; j++) {
    if constexpr (PAT::dense) col[j] = (j > i && j < n && lane < n) ? Lf[HL(j, i)] : Real(0);
    else col[j] = (j < PAT::n && lane < j && ((PAT::row(j) >> i) & 1u)) ? Lf[PAT::hbase(j) + i] : Real(0);   // structural zeros are not stored
  }
#pragma unroll
  for (int j = NP - 1; j >= 0; j--) {
    if (j < n) {
      const Real xj = readlane_<Real>(xi * si, j);   // lane j's entry is final here: x_j = (x_j - sum) / L_jj
      xi = (lane == j) ? xj : xi - col[j] * xj;
    }
  }
  __syncthreads();
  if (lane < n) x[lane] = xi;
  __syncthreads();
}
template <class Real, bool BIG = false, int TAG = 0, class PAT = DensePattern>
__device__ __forceinline__ void sp_chol_backsolve(const Real* Lf, const Real* sinv, int n, Real* x, int lane) {
  if constexpr (!PAT::dense) { sp_chol_backsolve_t<Real, sp_npad(PAT::n), TAG, PAT>(Lf, sinv, n, x, lane); return; }
  if constexpr (!BIG) { sp_chol_backsolve_lds<Real>(Lf, sinv, n, x, lane); return; }
  const int np = sp_npad(n);
  if (np <= 8) sp_chol_backsolve_t<Real, 8, TAG>(Lf, sinv, n, x, lane);
  else if (np <= 16) sp_chol_backsolve_t<Real, 16, TAG>(Lf, sinv, n, x, lane);
  else if (np <= 24) sp_chol_backsolve_t<Real, 24, TAG>(Lf, sinv, n, x, lane);
  else sp_chol_backsolve_t<Real, 32, TAG>(Lf, sinv, n, x, lane);
}

// x <- L^-1 x (forward), column-oriented like the back-substitution
template <class Real>
__device__ __forceinline__ void sp_chol_fwdsolve(const Real* Lf, const Real* sinv, int n, Real* x, int lane) {
  for (int j = 0; j < n; j++) {
    __syncthreads();
    const Real xj = x[j] * sinv[j];
    DART_LOCKSTEP_FENCE();   // every lane has read x[j] before lane j overwrites it
    if (lane == j) x[j] = xj;
    if (lane > j && lane < n) x[lane] -= Lf[HL(lane, j)] * xj;
  }
  __syncthreads();
}

// Stable-PD torque of DartWalker3dSPD-v1 (walker3d_spd.py:40-55), evaluated before every world step once M (in S.H, with
// the integrator's diagonal terms when card.impulse_inertia = 0), the bias forces c (S.b) and the previous step's constraint forces (S.cf) are known:
//   qdd = (M + Kd dt_env)^-1 (-c + p + d + cf),  tau = p + d - Kd qdd dt_env,  root dofs zeroed, |tau| <= limit.
// S.tau holds the target pose; the torque goes straight into the right-hand side.  Workspace: S.A (factor), S.r (1/L_jj),
// S.lo (the solve) -- all idle until the constraint phase.
template <class Real>
__device__ __forceinline__ void sp_spd_torque(const LinkConst<Real>& lc, const SpatialModel<Real>& Md, SpLds<Real>& S, int lane) {
  const int n = Md.n, np = sp_npad(n);
  Real pd = Real(0), kd = Real(0);
  if (lane < np) for (int k = 0; k <= lane; k++) S.A[HL(lane, k)] = S.H[HL(lane, k)];
  __syncthreads();
  if (lane < n) {   // dof `lane` sits at row n-1-lane of the factor's storage order (sp_mass_row)
    const int rv = n - 1 - lane;
    kd = Md.spd_kd[lane];
    S.A[HL(rv, rv)] += kd * Md.envdt - (Md.impulse_M ? Real(0) : lc.d_diag);   // S.H holds M, or M + E with the A3 knob at 0
    const Real p = -Md.spd_kp[lane] * (S.q[lane] + S.dq[lane] * Md.envdt - S.tau[lane]);
    const Real d = -kd * S.dq[lane];
    pd = p + d;
    S.lo[rv] = -S.b[lane] + p + d + S.cf[lane];
  }
  __syncthreads();
  sp_cholesky<Real>(S.A, S.r, n, lane);
  sp_chol_fwdsolve<Real>(S.A, S.r, n, S.lo, lane);
  sp_chol_backsolve<Real>(S.A, S.r, n, S.lo, lane);
  if (lane < n) {
    Real tq = pd - kd * S.lo[n - 1 - lane] * Md.envdt;
    const int k = lane - Md.act_dof0;
    if (k < 0 || k >= Md.act_dim) tq = Real(0);
    else if (fabs(tq) > Md.act_scale[k]) tq = (tq > Real(0) ? Real(1) : Real(-1)) * Md.act_scale[k];
    S.rhs[lane] += tq;
  }
  __syncthreads();
}

// Boxed LCP by block principal pivoting, one wavefront per problem (rows = lanes).  F/U are wave-uniform bit masks.
// LDS version: the masked LDL^T lives in S.Lw, one barrier per eliminated column.  Serves problems with more than 40 rows
// (link-link contacts) and hosts the PGS safety net; smaller problems take sp_blcp_t below.
template <class Real>
__device__ __forceinline__ void sp_blcp_lds(SpLds<Real>& S, int m, uint64_t pinmask, uint64_t& F, uint64_t& U, int max_iter,
                                       int pgs_sweeps, unsigned long long* stats, int lane, const bool ZERO_BOUNDS, int pf_ncp = -1, int pf_m1 = 0) {
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
    // Sweep order = the oracle's row order (oracle_step: {n, t1, t2} per contact, then limits, then joint friction), whatever the
    // storage order: a Gauss-Seidel iterate after a FIXED number of sweeps depends on it.  pf_ncp >= 0: the rows are stored in
    // prefix order (normals [0, ncp), limits / joint friction [ncp, m1), tangents [m1, m1 + 2 ncp) -- sp_world_step).
    // (the sweep positions run over the UNtruncated row count of the prefix layout: when the model's LCP capacity clipped m, the limit /
    // joint-friction rows sit at positions beyond m and must still be visited -- ADVICE r4; rows that were dropped are skipped below)
    const int npos = pf_ncp >= 0 ? pf_m1 + 2 * pf_ncp : m;
    for (int sw = 0; sw < pgs_sweeps; ++sw)
      for (int p = 0; p < npos; p++) {
        int i = p;
        if (pf_ncp >= 0) i = p < 3 * pf_ncp ? ((p % 3 == 0) ? p / 3 : pf_m1 + 2 * (p / 3) + (p % 3 - 1)) : pf_ncp + (p - 3 * pf_ncp);
        if (i >= m) continue;            // (rows beyond the model's LCP capacity were dropped)
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


#ifndef SP_BLCP_MAXREG
#define SP_BLCP_MAXREG 40
#endif
// Dispatch.  BIG kernels (the 20+-dof models: HumanWalker, Walker3d): register solver for problems of up to 40 rows, the LDS
// solver beyond; a register solve that hits its iteration cap re-enters the LDS solver with no pivoting iterations = its PGS
// safety net.  The small-model kernels keep the LDS solver: measured, the register arrays cost them more (occupancy, call
// overhead) than the saved barriers give back.
template <class Real, bool BIG = false, int TAG = 0>
__device__ __forceinline__ void sp_blcp(SpLds<Real>& S, int m, uint64_t pinmask, uint64_t& F, uint64_t& U, int max_iter,
                                       int pgs_sweeps, unsigned long long* stats, int lane, const bool ZERO_BOUNDS, int mv = 0,
                                       int pf_ncp = -1, int pf_m1 = 0) {
  // mv: row count that picks the solver variant (>= m): both stages of a world step can share one variant's code (instruction cache)
  // pf_ncp / pf_m1: the prefix row layout of the caller, for the PGS sweeps' order (sp_blcp_lds)
  mv = mv > m ? mv : m;
  if (!BIG || mv > SP_BLCP_MAXREG || max_iter == 0) { sp_blcp_lds<Real>(S, m, pinmask, F, U, max_iter, pgs_sweeps, stats, lane, ZERO_BOUNDS, pf_ncp, pf_m1); return; }
  if (lane < m) S.x0[lane] = S.x[lane];   // solution of the previous stage (zeros before the first): PGS fallback start
  BlcpSets r;
  if (mv <= 8) r = sp_blcp_t<Real, 8, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
  else if (mv <= 12) r = sp_blcp_t<Real, 12, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
  else if (mv <= 16) r = sp_blcp_t<Real, 16, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
  else if (mv <= 24) r = sp_blcp_t<Real, 24, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
  else if (mv <= 32) r = sp_blcp_t<Real, 32, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
#if SP_BLCP_MAXREG > 24
  else r = sp_blcp_t<Real, 40, false, TAG>(S.A, S.b, S.lo, S.hi, S.x, m, pinmask, F, U, max_iter, stats, lane, ZERO_BOUNDS);
#else
  else r = BlcpSets{F, U, false, 0};
#endif
  F = r.F; U = r.U;
  const bool ok = r.ok;
  __syncthreads();
  if (!ok) {
    if (stats && lane == 0) atomicAdd(&stats[32], 1ull);
    if (lane < m) S.x[lane] = S.x0[lane];
    __syncthreads();
    sp_blcp_lds<Real>(S, m, pinmask, F, U, 0, pgs_sweeps, nullptr, lane, ZERO_BOUNDS, pf_ncp, pf_m1);
  }
}

}  // namespace dartk
