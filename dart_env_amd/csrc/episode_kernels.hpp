// episode_kernels.hpp -- per-env episode return / length accumulators kept in HBM, the batched counterpart of the
// reference's RecordEpisodeStatistics wrapper (reference gym/wrappers/record_episode_statistics.py:22-34): return += reward,
// length += 1 every step; when an env reports done its totals are latched into last_return / last_length (what the wrapper
// puts into info['episode']), added to the running sums of finished episodes, and the accumulators restart from zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dartk {

// totals: [0] sum of returns, [1] sum of lengths, [2] number of finished episodes
__global__ void episode_stats_kernel(int64_t n_envs, const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                     double* __restrict__ ep_ret, int32_t* __restrict__ ep_len,
                                     double* __restrict__ last_ret, int32_t* __restrict__ last_len,
                                     double* __restrict__ totals) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double r = 0.0, l = 0.0;
  bool fin = false;
  if (e < n_envs) {
    const double ret = ep_ret[e] + (double)reward[e];
    const int32_t len = ep_len[e] + 1;
    fin = done[e] != 0;
    if (fin) { last_ret[e] = ret; last_len[e] = len; r = ret; l = (double)len; }
    ep_ret[e] = fin ? 0.0 : ret;
    ep_len[e] = fin ? 0 : len;
  }
  // one atomic per wave for the running sums of finished episodes
  const uint64_t any = __ballot(fin);
  if (any) {
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o); l += __shfl_xor(l, o); }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&totals[0], r); atomicAdd(&totals[1], l); atomicAdd(&totals[2], (double)__popcll(any));
    }
  }
}

// wrapper.reset(): the accumulators of the masked envs (all when mask == nullptr) restart from zero
__global__ void episode_reset_kernel(int64_t n_envs, const uint8_t* __restrict__ mask, double* __restrict__ ep_ret,
                                     int32_t* __restrict__ ep_len) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs || (mask && !mask[e])) return;
  ep_ret[e] = 0.0; ep_len[e] = 0;
}

}  // namespace dartk
