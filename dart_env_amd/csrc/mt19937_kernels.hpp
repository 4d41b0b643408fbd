// mt19937_kernels.hpp -- a bank of per-env MT19937 generators in HBM, so that reset noise is BIT-EXACT with the
// reference at any batch size without a host loop.
//
// The reference seeds each env with `seeding.np_random(seed)` (reference gym/utils/seeding.py:11-19: seed -> SHA-512 ->
// uint32 words -> numpy RandomState.seed(list) = MT19937 init_by_array) and draws
// `np_random.uniform(-r, r, ndofs)` for qpos, then for qvel (reference gym/envs/dart/hopper.py:78-79).  numpy's legacy
// uniform is  low + (high - low) * ((a >> 5) * 2^26 + (b >> 6)) / 2^53  with two successive 32-bit outputs a, b.
// State layout: mt[624][N] (env fastest -> coalesced when one lane serves one env), pos[N] = slot of the next word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cr_log.hpp"
#include "mt19937_draw.hpp"   // MtBankView + the one-env draw the lane kernels call from their epilogue (round 6)

namespace dartk {

// init_by_array(key[0..len)) for every env; keys: [N][2], len[N] in {1, 2}
__global__ void mt_seed_kernel(int64_t n_envs, uint32_t* __restrict__ mt, int32_t* __restrict__ pos,
                               const uint32_t* __restrict__ keys, const int32_t* __restrict__ key_len) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  auto M = [&](int i) -> uint32_t& { return mt[(int64_t)i * n_envs + e]; };
  uint32_t prev = 19650218u;
  M(0) = prev;
  for (int i = 1; i < 624; i++) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i; M(i) = prev; }
  const int len = key_len[e];
  const uint32_t k0 = keys[2 * e], k1 = keys[2 * e + 1];
  int i = 1, j = 0;
  for (int k = 624; k; k--) {
    const uint32_t p = M(i - 1);
    M(i) = (M(i) ^ ((p ^ (p >> 30)) * 1664525u)) + (j == 0 ? k0 : k1) + (uint32_t)j;
    i++; j++;
    if (i >= 624) { M(0) = M(623); i = 1; }
    if (j >= len) j = 0;
  }
  for (int k = 623; k; k--) {
    const uint32_t p = M(i - 1);
    M(i) = (M(i) ^ ((p ^ (p >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) { M(0) = M(623); i = 1; }
  }
  M(0) = 0x80000000u;
  pos[e] = 0;     // numpy regenerates before its first draw: the first output is x[624], produced over slot 0
}

enum { MT_EXTRA_NONE = 0, MT_EXTRA_SWINGUP = 1, MT_EXTRA_REACHER2D = 2, MT_EXTRA_REACHER3D = 3, MT_EXTRA_GAUSS_VEL = 4 };

// For every env with mask[e] != 0 (mask == nullptr: all): draw 2*ndofs doubles and emit
//   qn[e][d] = init_pos[d] + U(-r, r),  vn[e][d] = init_vel[d] + U(-rv, rv)     (row-major doubles, as dart_reset takes them)
//
// The generator runs in its incremental form: output k is the tempered  x[k+624] = x[k+397] ^ twist(x[k], x[k+1]),
// written over x[k]'s slot -- the same in-place order numpy's block regeneration uses, so the streams are identical, but
// a reset costs a fixed 4*ndofs word updates instead of an occasional 624-word loop that would stall the whole launch
// on the few lanes that hit it.  Words are produced 8 at a time so that the 17 loads of a chunk are in flight together
// (lanes of a wave are sparse here -- only the envs that just finished -- so this kernel is latency-, not bandwidth-bound).
__global__ void mt_draw_kernel(int64_t n_envs, int ndofs, uint32_t* __restrict__ mt, int32_t* __restrict__ pos,
                               const uint8_t* __restrict__ mask, double low_q, double range_q, double low_v,
                               double range_v, const double* __restrict__ init_pos, const double* __restrict__ init_vel,
                               double* __restrict__ qn, double* __restrict__ vn, int extra, double* __restrict__ tvals,
                               double* __restrict__ gauss_cache = nullptr, int32_t* __restrict__ has_gauss = nullptr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  if (mask && !mask[e]) return;
  auto M = [&](int i) -> uint32_t& { return mt[(int64_t)(i >= 624 ? i - 624 : i) * n_envs + e]; };
  // numpy rounds the product and the sum separately (no fma); the reference then adds the noise to q / dq.
  // (HIP's __dmul_rn is a plain `*` and still contracts, hence the pragma.)
  auto affine = [](double base, double low, double range, double u) -> double {
#pragma clang fp contract(off)
    const double prod = range * u;
    const double noise = low + prod;
    return base + noise;
  };
  auto temper = [](uint32_t y) -> uint32_t {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  };
  int p = pos[e];                       // slot of the next word to produce, 0..623
  // MT_EXTRA_GAUSS_VEL: only the positions are uniform; the velocities are `randn(ndofs) * range_v` (below)
  const int n_doubles = extra == MT_EXTRA_GAUSS_VEL ? ndofs : 2 * ndofs;
  for (int d0 = 0; d0 < n_doubles; d0 += 4) {
    uint32_t lo[9], hi[8], out[8];
#pragma unroll
    for (int j = 0; j < 9; j++) lo[j] = M(p + j);
#pragma unroll
    for (int j = 0; j < 8; j++) hi[j] = M(p + 397 + j);   // never a slot this chunk overwrites (397 + j - i != 0, 624)
    const int cnt = n_doubles - d0 < 4 ? n_doubles - d0 : 4;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t y = (lo[j] & 0x80000000u) | (lo[j + 1] & 0x7fffffffu);
      const uint32_t x = hi[j] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      out[j] = temper(x);
      if (j < 2 * cnt) M(p + j) = x;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (t < cnt) {
        const uint32_t a = out[2 * t] >> 5, b = out[2 * t + 1] >> 6;
        const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        const int d = d0 + t;
        if (d < ndofs) qn[e * ndofs + d] = affine(init_pos[d], low_q, range_q, u);
        else vn[e * ndofs + d - ndofs] = affine(init_vel[d - ndofs], low_v, range_v, u);
      }
    }
    p += 2 * cnt;
    if (p >= 624) p -= 624;
  }
  // ---- what some reset_model()s draw after the two noise vectors, from the same stream, one double at a time
  if (extra != MT_EXTRA_NONE) {
    auto next_word = [&]() -> uint32_t {
      const uint32_t a = M(p), b = M(p + 1);
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      const uint32_t x = M(p + 397) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      M(p) = x;
      p = p + 1 >= 624 ? 0 : p + 1;
      return temper(x);
    };
    auto next_double = [&]() -> double {
      const uint32_t a = next_word() >> 5, b = next_word() >> 6;
      return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    };
    if (extra == MT_EXTRA_GAUSS_VEL) {
      // inverted_double_pendulum.py:50-51: qvel = dq + np_random.randn(ndofs) * .1 -- numpy's LEGACY Gaussian (legacy-distributions.c:
      // legacy_gauss): polar Box-Muller, two uniforms per try, the second deviate of a pair cached in the generator across calls (and
      // across resets: gauss_cache / has_gauss live in HBM per env like the MT state).  Every operation is a single IEEE rounding as in
      // the C original (no contraction); log is dartk::log_cr (cr_log.hpp: the host libm's value in 99.92 % of the draws, 1 ulp otherwise).
      double cached = gauss_cache[e];
      int has = has_gauss[e];
      for (int d = 0; d < ndofs; d++) {
        double g;
        if (has) { g = cached; has = 0; cached = 0.0; }
        else {
          double x1, x2, r2;
          do {
#pragma clang fp contract(off)
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
          } while (r2 >= 1.0 || r2 == 0.0);
          double f;
          {
#pragma clang fp contract(off)
            const double num = -2.0 * log_cr(r2);
            f = sqrt(num / r2);
            cached = f * x1;
            g = f * x2;
          }
          has = 1;
        }
        double noise;
        {
#pragma clang fp contract(off)
          noise = g * range_v;                 // (range_v carries the scale of the Gaussian here: .1)
          vn[e * ndofs + d] = init_vel[d] + noise;
        }
      }
      gauss_cache[e] = cached; has_gauss[e] = has;
    } else if (extra == MT_EXTRA_SWINGUP) {
      // cartpole_swingup.py:42-45: `if np_random.uniform(0, 1, 1) > 0.5: qpos[1] += pi else: qpos[1] += -pi`
      const double u = affine(0.0, 0.0, 1.0, next_double());
      qn[e * ndofs + 1] += (u > 0.5) ? 3.141592653589793 : -3.141592653589793;
    } else {
      // reacher2d.py:51-55 / reacher.py:52-54: rejection-sample the target inside a disc / ball
      const bool planar = extra == MT_EXTRA_REACHER2D;
      const double low = planar ? -0.2 : -1.0, range = planar ? 0.4 : 2.0, rmax = planar ? 0.2 : 1.5;
      double t0, t1, t2;
      for (;;) {
        t0 = affine(0.0, low, range, next_double());
        t1 = affine(0.0, low, range, next_double());
        t2 = affine(0.0, low, range, next_double());
        if (planar) t1 = 0.0;
        double ss;
        {
#pragma clang fp contract(off)
          ss = t0 * t0 + t1 * t1 + t2 * t2;
        }
        if (sqrt(ss) < rmax) break;
      }
      if (planar) t1 = 0.01;
      tvals[4 * e + 0] = t0; tvals[4 * e + 1] = t1; tvals[4 * e + 2] = t2; tvals[4 * e + 3] = 0.0;
    }
  }
  pos[e] = p;
}

}  // namespace dartk
