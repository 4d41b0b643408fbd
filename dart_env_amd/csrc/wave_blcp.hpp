// wave_blcp.hpp -- wave-cooperative boxed LCP (block principal pivoting) with the whole iteration in registers: one problem per
// wavefront, lane i owns row i.  Used by the tree kernel for every constraint solve (spatial_dense.hpp: sp_blcp) and by the lane
// kernels for the rare env with more contacts than their register tiers hold (planar_kernel.hpp: wave_constraints).
// Included by planar_kernel.hpp after TI / tol_ / rcp_ are defined; device code only (readlane, DPP).
#pragma once

// Value-identity fence for the systolic (v_readlane-broadcast) algorithms: `x` comes back as a value the compiler knows nothing about
// (an empty asm with a read-write VGPR operand: no instruction).  Why: sp_cholesky_t broadcasts L_kj = readlane(row[j], k) once in
// the factorisation and once more, much later, in the W = L^-1 J^T substitution; the compiler recognises the two as the same value
// and keeps all 222 broadcast pairs of a HumanWalker factor alive in between -- 444 SGPRs in a file of ~100 -- i.e. it spills each
// into a VGPR lane (2 x v_writelane) and reads it back (2 x v_readlane) where simply broadcasting again costs 2 x v_readlane
// (round 4, from the disassembly of the fp64 pattern kernel: 1 078 SGPR spills).  Host builds (tests/kernel_emu) define it away.
// DART_TIE(x, after): the same fence with an input -- `x` becomes a value that exists only once `after` has been computed, so a
// v_readlane of x cannot be scheduled before that point.  Why: where every broadcast source is ready at the top of a long unrolled
// region (the W = L^-1 J^T substitution below the factorisation: 222 broadcasts of finished factor entries), the scheduler hoists ALL
// the v_readlanes to the top of the region "to cover latency" and the register allocator then parks each SGPR pair in a VGPR lane
// (2 x v_writelane + s_nops) and fetches it back before use -- found in the disassembly, round 4: 409 v_writelane in that region.
#ifndef DART_OPAQUE
#if defined(__HIP_DEVICE_COMPILE__)
#define DART_OPAQUE(x) asm volatile("" : "+v"(x))
#define DART_TIE(x, after) asm volatile("" : "+v"(x) : "v"(after))
#else
#define DART_OPAQUE(x) ((void)0)
#define DART_TIE(x, after) ((void)0)
#endif
#endif

// A pointer the CALLER knows to be LDS, re-typed as such inside a noinline function: its generic-pointer arguments would otherwise
// compile to flat_load / flat_store (the vector-memory path, aperture check first) instead of ds_read / ds_write.
#if defined(__HIP_DEVICE_COMPILE__)
#define DART_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#else
#define DART_LDS_PTR(T, p) (p)
#endif

namespace dartk {

template <class Real> __device__ __forceinline__ Real readlane_(Real x, int l);
template <> __device__ __forceinline__ float readlane_<float>(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
template <> __device__ __forceinline__ double readlane_<double>(double x, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------ wave-level sum through DPP (no LDS, ~8 VALU instructions)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Sum of x over the 64 lanes, returned to every lane.  Must be called by the whole wavefront (convergent code).
// row_shr:1,2,4,8 build an inclusive prefix sum inside each row of 16 lanes (lanes shifted in from outside a row read the
// `old` operand = 0), row_bcast:15 / row_bcast:31 carry the row totals forward; lane 63 ends up with the total.
template <class Real>
__device__ __forceinline__ Real wave_sum(Real x) {
  x = dpp_add_<0x111, 0xf>(x);   // row_shr:1
  x = dpp_add_<0x112, 0xf>(x);   // row_shr:2
  x = dpp_add_<0x114, 0xf>(x);   // row_shr:4
  x = dpp_add_<0x118, 0xf>(x);   // row_shr:8
  x = dpp_add_<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  x = dpp_add_<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
  return readlane_<Real>(x, 63);
}

// Maximum of NON-NEGATIVE x over the 64 lanes (same DPP ladder; lanes shifted in from outside a row read 0, the identity here).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, t));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return fmax(v, __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo));
}
template <class Real>
__device__ __forceinline__ Real wave_max_nonneg(Real x) {
  x = dpp_max_<0x111, 0xf>(x);
  x = dpp_max_<0x112, 0xf>(x);
  x = dpp_max_<0x114, 0xf>(x);
  x = dpp_max_<0x118, 0xf>(x);
  x = dpp_max_<0x142, 0xa>(x);
  x = dpp_max_<0x143, 0xc>(x);
  return readlane_<Real>(x, 63);
}

// Boxed LCP by block principal pivoting with the whole iteration in REGISTERS: lane i holds row i of A and row i of the working
// copy of the free block; a pivot row travels through v_readlane (an SGPR operand of the FMA).  No LDS traffic and no barrier inside
// the pivoting loop (the LDS version paid an LDS round trip + barrier per eliminated column).
// The free block is solved by GAUSS-JORDAN elimination on [A_FF | r] (round 4; before: masked LDL^T + two triangular solves).  With
// one row per lane an elimination step is the same instructions whether it clears the column below the pivot or below AND above it
// (every lane runs the update of its row either way), so eliminating both sides is free -- and the transposed solve L^T x = y, which
// in this layout needed one 20-instruction DPP wave sum per free row on the critical path, disappears: after the last pivot
// x_i = r_i / a_ii.  A_FF is symmetric positive definite (J M^-1 J^T + cfm), no pivoting needed.
// MP = compile-time row capacity (variants 8 ... 40: only the one a wave takes enters the instruction cache); rows >= m are inert padding.
// EXT (the lane kernels' callers): bmax_more, keep_last and the iteration count in the result are live; the tree kernel instantiates
// EXT = false.
struct BlcpSets { uint64_t F, U; bool ok; int iters; };
// TAG: a kernel that wants private copies of this function instantiates its own tag -- the register budget of a non-kernel function is
// the loosest one among the kernels that call it (waves per SIMD are a kernel attribute the compiler propagates to callees), so a
// kernel built for 2 waves per SIMD must not share its callees with kernels built for 1 (the tree kernel's fp64 pattern kernel).
template <class Real, int MP, bool EXT = false, int TAG = 0>
static __device__ __attribute__((noinline)) BlcpSets sp_blcp_t(const Real* __restrict__ Ap_, const Real* __restrict__ bp_, const Real* __restrict__ lop_,
                                                        const Real* __restrict__ hip_, Real* __restrict__ xp_, int m, uint64_t pinmask, uint64_t F,
                                                        uint64_t U, int max_iter, unsigned long long* stats, int lane, const bool ZERO_BOUNDS,
                                                        Real bmax_more = Real(0), bool keep_last = false) {
  // the problem lives in LDS with every caller (the tree kernel's SpLds block, the lane kernels' hand-off buffer)
  const auto Ap = DART_LDS_PTR(const Real, Ap_), bp = DART_LDS_PTR(const Real, bp_), lop = DART_LDS_PTR(const Real, lop_), hip = DART_LDS_PTR(const Real, hip_);
  const auto xp = DART_LDS_PTR(Real, xp_);
  // bmax_more: |b| of rows the caller left out of this solve (they enter the feasibility tolerance as in blcp_bpp)
  // keep_last: when the cap is reached the last iterate is written to xp, clamped into the box (default: xp is left alone)
  // a real function call (not inlined): its register arrays get their own allocation instead of raising the pressure of the
  // whole step kernel; the handful of loads / the one store below go through plain pointers
  const bool row = lane < m;
  Real Ar[MP];
#pragma unroll
  for (int j = 0; j < MP; j++) Ar[j] = (row && j < m) ? Ap[TI(lane, j)] : Real(0);
  const Real bi = row ? bp[lane] : Real(0), loi = row ? lop[lane] : Real(0), hii = row ? hip[lane] : Real(0);
  const Real bmax = wave_max_nonneg<Real>(EXT ? fmax(fabs(bi), bmax_more) : fabs(bi));
  const Real tol = tol_<Real>() * (Real(1) + bmax);
  int best = m + 1, patience = 3;
  bool converged = false;
  int it = 0;
  Real rr = Real(0);
  for (; it < max_iter; ++it) {
    const bool fi = row && ((F >> lane) & 1ull), ui = row && ((U >> lane) & 1ull);
    const Real xb = row ? (fi ? Real(0) : (ui ? hii : loi)) : Real(0);
    Real t = bi;
    if (!ZERO_BOUNDS) {
#pragma unroll
      for (int j = 0; j < MP; j++) t -= Ar[j] * readlane_<Real>(xb, j);
    }
    rr = fi ? t : xb;
    // working copy of the free block, full rows (non-free rows / columns are identity)
    Real L[MP];
#pragma unroll
    for (int j = 0; j < MP; j++) {
      const bool fj = (F >> j) & 1ull;
      L[j] = (fi && fj) ? Ar[j] : ((j == lane) ? Real(1) : Real(0));
    }
    Real invd_own = Real(1);
#pragma unroll
    for (int j = 0; j < MP; j++) {
      if ((F >> j) & 1ull) {   // wave-uniform
        const Real inv = rcp_<Real>(readlane_<Real>(L[j], j));
        const Real mi = (lane != j) ? L[j] * inv : Real(0);   // this row's multiplier of pivot row j (0 in non-free rows: L[j] = 0 there)
        invd_own = (lane == j) ? inv : invd_own;
#pragma unroll
        for (int k = j + 1; k < MP; k++) L[k] -= mi * readlane_<Real>(L[k], j);   // lane j itself has mi = 0: the pivot row stays
        rr -= mi * readlane_<Real>(rr, j);
      }
    }
    rr = fi ? rr * invd_own : rr;
    // w = A x - b and the feasibility of every row
    Real w = -bi;
#pragma unroll
    for (int j = 0; j < MP; j++) w += Ar[j] * readlane_<Real>(rr, j);
    bool inf = false, gt = false;
    if (row) {
      const bool pinned = (pinmask >> lane) & 1ull;
      const bool over = rr > hii + tol * (Real(1) + fabs(hii)), under = rr < loi - tol * (Real(1) + fabs(loi));
      const bool wbad = ui ? (w > tol) : (w < -tol);
      inf = fi ? (over || under) : (wbad && !pinned);
      gt = rr > hii;
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
  if (stats && lane == 0) { atomicAdd(&stats[it < 31 ? it : 31], 1ull); atomicAdd(&stats[33], 1ull); }
  if ((converged || (EXT && keep_last)) && row) xp[lane] = fmin(fmax(rr, loi), hii);
  return BlcpSets{F, U, converged, EXT ? it : 0};
}

// ------------------------------------------------------------------ FOUR problems per wavefront: one per row of 16 lanes (round 5)
// The lane kernels' envs beyond the small register tier are 6-12-row problems: served one at a time (sp_blcp_t above) they keep at most
// 16 of the 64 lanes busy, and at one wave per SIMD that latency is the kernel's time (half cheetah: ~3 such lanes per wave and world
// step).  Here lane 16 g + i holds row i of group g's problem and the four problems go through the SAME instruction stream:
//   * a pivot row travels through DPP `row_newbcast:j` (gfx90a+: lane j of every 16-lane row to all lanes of that row) where the
//     one-problem solver uses v_readlane -- each group reads its own pivot;
//   * the sets F / U / pinned are 16-bit group values (the group's slice of the wave's ballot), group-uniform VGPRs;
//   * "column j is not free" differs between the groups, so it is not a branch: the multiplier of a non-free column is exactly 0 (its row
//     of the working copy is the identity row) and  x - 0 * y = x  for the finite y a bound row holds -- a column is skipped only
//     when NO group has it free (wave-uniform), which skips exact no-ops;
//   * a converged group keeps iterating on unchanged sets while the others finish: the same instructions on the same inputs, the same
//     result.  So a problem's solution does not depend on the group it sits in nor on its three neighbours (tests/test_gpu_wave_blcp.py).
// Same start sets, tolerances, patience and single-pivot rule as sp_blcp_t / blcp_bpp.  m <= 16 rows per group (0: an idle group).
// (device: the type-generic builtin with the value itself as `old` and bound_ctrl -- every lane has a source with row_newbcast -- compiles to
// ONE v_mov_b64_dpp / v_mov_b32_dpp; written with old = 0 on 32-bit halves it was v_mov 0 + v_mov_dpp per half, four instructions per double)
template <int J> __device__ __forceinline__ float row_bcast_(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true);
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, false));
#endif
}
template <int J> __device__ __forceinline__ double row_bcast_(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true);
#else
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + J, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + J, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
#endif
}
// maximum of NON-NEGATIVE x over the lane's row of 16, returned to every lane of the row
template <class Real> __device__ __forceinline__ Real row_max_nonneg(Real x) {
  x = dpp_max_<0x111, 0xf>(x);
  x = dpp_max_<0x112, 0xf>(x);
  x = dpp_max_<0x114, 0xf>(x);
  x = dpp_max_<0x118, 0xf>(x);
  return row_bcast_<15>(x);
}
// the lane's group's 16 bits of a wave ballot
__device__ __forceinline__ uint32_t row_ballot(bool p, int lane) { return (uint32_t)(__ballot(p) >> (lane & 48)) & 0xffffu; }

struct Blcp4Sets { uint32_t F, U; bool ok; int iters; };   // F, U, ok: of the lane's group; iters: of the wave (its slowest group)
template <class Real, int TAG = 0>
static __device__ __attribute__((noinline)) Blcp4Sets sp_blcp4_t(const Real* __restrict__ Ap_, const Real* __restrict__ bp_, const Real* __restrict__ lop_,
                                                          const Real* __restrict__ hip_, Real* __restrict__ xp_, int m, uint32_t pinmask, uint32_t F,
                                                          uint32_t U, int max_iter, int lane, const bool ZERO_BOUNDS, bool keep_last) {
  constexpr int MP = 16;
  // the lane's GROUP's operands (LDS, as with every caller of sp_blcp_t); m, pinmask, F, U: group-uniform
  const auto Ap = DART_LDS_PTR(const Real, Ap_), bp = DART_LDS_PTR(const Real, bp_), lop = DART_LDS_PTR(const Real, lop_), hip = DART_LDS_PTR(const Real, hip_);
  const auto xp = DART_LDS_PTR(Real, xp_);
  const int l = lane & 15;
  const bool row = l < m;
  Real Ar[MP];
#pragma unroll
  for (int j = 0; j < MP; j++) Ar[j] = (row && j < m) ? Ap[TI(l, j)] : Real(0);
  const Real bi = row ? bp[l] : Real(0), loi = row ? lop[l] : Real(0), hii = row ? hip[l] : Real(0);
  const Real bmax = row_max_nonneg<Real>(fabs(bi));
  const Real tol = tol_<Real>() * (Real(1) + bmax);
  int best = m + 1, patience = 3;
  bool converged = m == 0;
  int it = 0;
  Real rr = Real(0);
  for (; it < max_iter; ++it) {
    const bool fi = row && ((F >> l) & 1u), ui = row && ((U >> l) & 1u);
    const Real xb = row ? (fi ? Real(0) : (ui ? hii : loi)) : Real(0);
    Real t = bi;
    if (!ZERO_BOUNDS) sfor<0, MP>([&](auto Jc) { constexpr int j = Jc; t -= Ar[j] * row_bcast_<j>(xb); });
    rr = fi ? t : xb;
    Real L[MP];
#pragma unroll
    for (int j = 0; j < MP; j++) {
      const bool fj = (F >> j) & 1u;
      L[j] = (fi && fj) ? Ar[j] : ((j == l) ? Real(1) : Real(0));
    }
    Real invd_own = Real(1);
    sfor<0, MP>([&](auto Jc) {
      constexpr int j = Jc;
      if (__ballot((F >> j) & 1u) != 0ull) {   // wave-uniform: some group has column j free
        // (a group whose column j is NOT free: lane j's row is the identity row -> inv = 1, every multiplier of the column exactly 0)
        const Real inv = rcp_<Real>(row_bcast_<j>(L[j]));
        const Real mi = (l != j) ? L[j] * inv : Real(0);
        invd_own = (l == j) ? inv : invd_own;
        sfor<j + 1, MP>([&](auto Kc) { constexpr int k = Kc; L[k] -= mi * row_bcast_<j>(L[k]); });
        rr -= mi * row_bcast_<j>(rr);
      }
    });
    rr = fi ? rr * invd_own : rr;
    Real w = -bi;
    sfor<0, MP>([&](auto Jc) { constexpr int j = Jc; w += Ar[j] * row_bcast_<j>(rr); });
    bool inf = false, gt = false;
    if (row) {
      const bool pinned = (pinmask >> l) & 1u;
      const bool over = rr > hii + tol * (Real(1) + fabs(hii)), under = rr < loi - tol * (Real(1) + fabs(loi));
      const bool wbad = ui ? (w > tol) : (w < -tol);
      inf = fi ? (over || under) : (wbad && !pinned);
      gt = rr > hii;
    }
    const uint32_t B = row_ballot(inf, lane), GT = row_ballot(gt, lane);
    converged = B == 0u;   // (a group that has converged finds B == 0 again on every later iteration: unchanged sets)
    if (__ballot(!converged) == 0ull) break;
    if (!converged) {
      const int ninf = __popc(B);
      const bool improved = ninf < best;
      const bool single = !improved && patience == 0;
      best = improved ? ninf : best;
      patience = improved ? 3 : (patience > 0 ? patience - 1 : 0);
      const uint32_t Bs = single ? (1u << (31 - __clz((int)B))) : B;
      const uint32_t toBound = Bs & F, toFree = Bs & ~F;
      F = (F & ~toBound) | toFree;
      U = (U & ~(toFree | toBound)) | (toBound & GT);
    }
  }
  if ((converged || keep_last) && row) xp[l] = fmin(fmax(rr, loi), hii);
  return Blcp4Sets{F, U, converged, it};
}

}  // namespace dartk
