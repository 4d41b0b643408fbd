// mt19937_draw.hpp -- the reset_model() draw of ONE env from the per-env MT19937 bank (mt19937_kernels.hpp), as a device function
// the lane kernels call from their epilogue (round 6: the reference-exact auto-reset fused into the step kernel).
//
// Replaces, for the envs that finished in this very launch:   self.np_random.uniform(low=-.005, high=.005, size=ndofs)  twice
// (reference gym/envs/dart/hopper.py:76-84, walker2d.py:76-84, ...) + set_state, which SyncVectorEnv.step_wait runs inside the step
// (sync_vector_env.py:77-78).  Rounds 1-5 ran it as two more launches behind the step kernel -- mt_draw_kernel over the done mask, then
// the masked reset kernel: 66.4 us against 44.5 us device-side per batched step of 65 536 hoppers.
#pragma once
#include <stdint.h>

namespace dartk {

// What a kernel needs of the bank, in device memory (dart_seed_mt19937 fills it; Extras::mt points at it, null = Philox / no bank).
struct MtBankView {
  uint32_t* mt;            // [624][n_envs], env fastest
  int32_t* pos;            // [n_envs] slot of the next word to produce
  const double* init_pos;  // [ndofs] world.reset() pose / velocity the noise is added to (doubles, as numpy adds them)
  const double* init_vel;
  double low_q, range_q, low_v, range_v;   // uniform(low, low + range)
};

// q[d] = init_pos[d] + U(low_q, low_q + range_q), dq[d] = init_vel[d] + U(low_v, low_v + range_v) for d < N, bit-exact with numpy's legacy
// uniform() on the env's RandomState: 4 N words in the generator's incremental form (output k is the tempered
// x[k+624] = x[k+397] ^ twist(x[k], x[k+1]), written over x[k]'s slot -- mt_draw_kernel's order).  ALL the words' inputs are loaded
// before the first is produced: the lanes that get here are the few that finished, nothing else hides an HBM round trip at the end of a
// kernel, and 4 N < 227 words never read a slot this call has already overwritten (x[k+397] of word k is word k-227's slot).
template <class Real, int N>
__device__ inline void mt_reset_draw(const MtBankView& B, int64_t n_envs, int64_t e, Real (&q)[N], Real (&dq)[N]) {
  constexpr int W = 4 * N;
  static_assert(W < 227, "mt_reset_draw: a word's x[k+397] input must not be a slot written earlier in the same call");
  uint32_t* const mt = B.mt + e;
  int p = B.pos[e];
  uint32_t lo[W + 1], hi[W];
#pragma unroll
  for (int j = 0; j <= W; j++) { int i = p + j; i = i >= 624 ? i - 624 : i; lo[j] = mt[(int64_t)i * n_envs]; }
#pragma unroll
  for (int j = 0; j < W; j++) { int i = p + 397 + j; i = i >= 624 ? i - 624 : i; i = i >= 624 ? i - 624 : i; hi[j] = mt[(int64_t)i * n_envs]; }
  const double low_q = B.low_q, range_q = B.range_q, low_v = B.low_v, range_v = B.range_v;
  uint32_t out[W];
#pragma unroll
  for (int j = 0; j < W; j++) {
    const uint32_t y = (lo[j] & 0x80000000u) | (lo[j + 1] & 0x7fffffffu);
    const uint32_t x = hi[j] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    int i = p + j; i = i >= 624 ? i - 624 : i;
    mt[(int64_t)i * n_envs] = x;
    uint32_t t = x;
    t ^= (t >> 11);
    t ^= (t << 7) & 0x9d2c5680u;
    t ^= (t << 15) & 0xefc60000u;
    t ^= (t >> 18);
    out[j] = t;
  }
  // numpy rounds the product and the sum separately (no fma), the reference then adds the noise to the reset state
  auto affine = [](double base, double low, double range, double u) -> double {
#pragma clang fp contract(off)
    const double prod = range * u;
    const double noise = low + prod;
    return base + noise;
  };
#pragma unroll
  for (int d = 0; d < 2 * N; d++) {
    const uint32_t a = out[2 * d] >> 5, b = out[2 * d + 1] >> 6;
    const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    if (d < N) q[d] = (Real)affine(B.init_pos[d], low_q, range_q, u);
    else dq[d - N] = (Real)affine(B.init_vel[d - N], low_v, range_v, u);
  }
  p += W;
  B.pos[e] = p >= 624 ? p - 624 : p;
}

}  // namespace dartk
