// spatial_dense.hpp -- dense wave-parallel pieces of the tree kernel: systolic register Cholesky, triangular solves, SPD controller torque, the boxed-LCP solver (block principal pivoting + PGS).
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include "spatial_dynamics.hpp"

namespace dartk {

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

}  // namespace dartk
