// TEST INFRASTRUCTURE -- the wave-cooperative boxed-LCP solver (dart_env_amd/csrc/wave_blcp.hpp: the tree kernel's constraint solver,
// and the lane kernels' wave-served fallback) behind a plain C entry point, so that tests/test_gpu_wave_blcp.py can feed it random
// problems and check the complementarity conditions of what comes back.  One problem per 64-lane workgroup, operands staged
// from global memory into LDS.  Built by __graft_entry__.build() into tests/gpu_kernels/libwave_blcp_harness.so; nothing under dart_env_amd/ loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planar_kernel.hpp"   // tol_, rcp_, TI and wave_blcp.hpp itself

using namespace dartk;

template <class Real, int MP, bool EXT>
__global__ void __launch_bounds__(64) blcp_harness_kernel(int n_problems, int mcap, const Real* A, const Real* b, const Real* lo, const Real* hi,
                                                          Real* x, const int* m, const uint64_t* pin, uint64_t* F, uint64_t* U, int* ok, int* iters,
                                                          int max_iter, int zero_bounds, int keep_last) {
  const int p = blockIdx.x;
  if (p >= n_problems) return;
  const int tri = mcap * (mcap + 1) / 2;
  // the solver's operands live in LDS with every product caller (it re-types its pointers as LDS: DART_LDS_PTR): stage them there
  __shared__ Real sm[40 * 41 / 2 + 4 * 40];
  Real* sA = sm; Real* sb = sm + tri; Real* slo = sb + mcap; Real* shi = slo + mcap; Real* sx = shi + mcap;
  for (int e = threadIdx.x; e < tri; e += 64) sA[e] = A[(size_t)p * tri + e];
  for (int e = threadIdx.x; e < mcap; e += 64) {
    sb[e] = b[(size_t)p * mcap + e]; slo[e] = lo[(size_t)p * mcap + e]; shi[e] = hi[(size_t)p * mcap + e]; sx[e] = x[(size_t)p * mcap + e];
  }
  __syncthreads();
  BlcpSets r;
  if constexpr (EXT) r = sp_blcp_t<Real, MP, true>(sA, sb, slo, shi, sx, m[p], pin[p], F[p], U[p], max_iter, nullptr, (int)threadIdx.x, zero_bounds != 0, Real(0), keep_last != 0);
  else r = sp_blcp_t<Real, MP>(sA, sb, slo, shi, sx, m[p], pin[p], F[p], U[p], max_iter, nullptr, (int)threadIdx.x, zero_bounds != 0);
  __syncthreads();
  for (int e = threadIdx.x; e < mcap; e += 64) x[(size_t)p * mcap + e] = sx[e];
  if (threadIdx.x == 0) { F[p] = r.F; U[p] = r.U; ok[p] = r.ok ? 1 : 0; iters[p] = r.iters; }
}

// four problems per workgroup, one per row of 16 lanes (sp_blcp4_t): problem 4 * blockIdx.x + g in lanes 16 g ... 16 g + 15; `rot` rotates the
// assignment of problems to groups (a problem's result must not depend on the group it sits in, nor on its neighbours)
template <class Real>
__global__ void __launch_bounds__(64) blcp4_harness_kernel(int n_problems, int mcap, const Real* A, const Real* b, const Real* lo, const Real* hi,
                                                           Real* x, const int* m, const uint64_t* pin, uint64_t* F, uint64_t* U, int* ok, int* iters,
                                                           int max_iter, int zero_bounds, int keep_last, int rot) {
  const int lane = (int)threadIdx.x, g = lane >> 4, l = lane & 15;
  const int p = 4 * (int)blockIdx.x + ((g + rot) & 3);
  const bool live = p < n_problems;
  constexpr int W = 16 * 17 / 2 + 4 * 16;
  __shared__ Real sm[4 * W];
  Real* sA = sm + g * W; Real* sb = sA + 136; Real* slo = sb + 16; Real* shi = slo + 16; Real* sx = shi + 16;
  const int tri = mcap * (mcap + 1) / 2;
  const int mp = live ? m[p] : 0;
  if (live) {
    for (int i = l; i < 16; i += 16) for (int j = 0; j <= i; j++) sA[TI(i, j)] = (i < mcap) ? A[(size_t)p * tri + TI(i, j)] : Real(0);
    sb[l] = l < mcap ? b[(size_t)p * mcap + l] : Real(0); slo[l] = l < mcap ? lo[(size_t)p * mcap + l] : Real(0);
    shi[l] = l < mcap ? hi[(size_t)p * mcap + l] : Real(0); sx[l] = l < mcap ? x[(size_t)p * mcap + l] : Real(0);
  }
  __syncthreads();
  const Blcp4Sets r = sp_blcp4_t<Real>(sA, sb, slo, shi, sx, mp, live ? (uint32_t)pin[p] : 0u, live ? (uint32_t)F[p] : 0u, live ? (uint32_t)U[p] : 0u, max_iter,
                                       lane, zero_bounds != 0, keep_last != 0);
  __syncthreads();
  if (live && l < mcap) x[(size_t)p * mcap + l] = sx[l];
  if (live && l == 0) { F[p] = r.F; U[p] = r.U; ok[p] = r.ok ? 1 : 0; iters[p] = r.iters; }
}

template <class Real>
static int run(int n, int mp, int ext, int mcap, const Real* A, const Real* b, const Real* lo, const Real* hi, Real* x, const int* m, const uint64_t* pin,
               uint64_t* F, uint64_t* U, int* ok, int* iters, int max_iter, int zero_bounds, int keep_last) {
  const size_t tri = (size_t)mcap * (mcap + 1) / 2;
  Real *dA, *db, *dlo, *dhi, *dx; int *dm, *dok, *dit; uint64_t *dpin, *dF, *dU;
#define CK(e) do { if ((e) != hipSuccess) return -1; } while (0)
  CK(hipMalloc(&dA, n * tri * sizeof(Real))); CK(hipMalloc(&db, n * mcap * sizeof(Real))); CK(hipMalloc(&dlo, n * mcap * sizeof(Real)));
  CK(hipMalloc(&dhi, n * mcap * sizeof(Real))); CK(hipMalloc(&dx, n * mcap * sizeof(Real))); CK(hipMalloc(&dm, n * sizeof(int)));
  CK(hipMalloc(&dok, n * sizeof(int))); CK(hipMalloc(&dit, n * sizeof(int))); CK(hipMalloc(&dpin, n * 8)); CK(hipMalloc(&dF, n * 8)); CK(hipMalloc(&dU, n * 8));
  CK(hipMemcpy(dA, A, n * tri * sizeof(Real), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, n * mcap * sizeof(Real), hipMemcpyHostToDevice));
  CK(hipMemcpy(dlo, lo, n * mcap * sizeof(Real), hipMemcpyHostToDevice)); CK(hipMemcpy(dhi, hi, n * mcap * sizeof(Real), hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x, n * mcap * sizeof(Real), hipMemcpyHostToDevice)); CK(hipMemcpy(dm, m, n * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMemcpy(dpin, pin, n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dF, F, n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dU, U, n * 8, hipMemcpyHostToDevice));
#define LAUNCH(MP, EXT) hipLaunchKernelGGL((blcp_harness_kernel<Real, MP, EXT>), dim3(n), dim3(64), 0, 0, n, mcap, dA, db, dlo, dhi, dx, dm, dpin, dF, dU, dok, dit, max_iter, zero_bounds, keep_last)
  if (mcap > 40) return -2;
  if (ext >= 4) {   // ext = 4 + rot: the four-problems-per-wave solver (16 rows)
    if (mp != 16 || mcap > 16) return -2;
    hipLaunchKernelGGL((blcp4_harness_kernel<Real>), dim3((n + 3) / 4), dim3(64), 0, 0, n, mcap, dA, db, dlo, dhi, dx, dm, dpin, dF, dU, dok, dit, max_iter, zero_bounds, keep_last, ext - 4);
  } else
  if (mp == 16 && ext) LAUNCH(16, true); else if (mp == 24 && ext) LAUNCH(24, true); else if (mp == 16) LAUNCH(16, false);
  else if (mp == 8) LAUNCH(8, false); else if (mp == 12) LAUNCH(12, false);
  else if (mp == 24) LAUNCH(24, false); else if (mp == 32) LAUNCH(32, false); else if (mp == 40) LAUNCH(40, false); else return -2;
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(x, dx, n * mcap * sizeof(Real), hipMemcpyDeviceToHost)); CK(hipMemcpy(F, dF, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(U, dU, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ok, dok, n * sizeof(int), hipMemcpyDeviceToHost)); CK(hipMemcpy(iters, dit, n * sizeof(int), hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(db); hipFree(dlo); hipFree(dhi); hipFree(dx); hipFree(dm); hipFree(dok); hipFree(dit); hipFree(dpin); hipFree(dF); hipFree(dU);
  return 0;
}

extern "C" {
// n problems of up to `mcap` rows each (row count m[p]); A packed lower triangle (TI), row-major per problem with stride mcap(mcap+1)/2;
// mp = the register variant (8 / 12 / 16 / 24 / 32 / 40 rows), ext = 1: the lane kernels' instantiation (mp 16 or 24); ext = 4 + rot (mp 16, mcap <= 16): sp_blcp4_t,
// four problems per wave, problem 4 w + ((g + rot) & 3) in group g.  Returns 0, -1 (HIP error), -2 (variant).
int wave_blcp_run_f64(int n, int mp, int ext, int mcap, const double* A, const double* b, const double* lo, const double* hi, double* x, const int* m,
                      const uint64_t* pin, uint64_t* F, uint64_t* U, int* ok, int* iters, int max_iter, int zero_bounds, int keep_last) {
  return run<double>(n, mp, ext, mcap, A, b, lo, hi, x, m, pin, F, U, ok, iters, max_iter, zero_bounds, keep_last);
}
int wave_blcp_run_f32(int n, int mp, int ext, int mcap, const float* A, const float* b, const float* lo, const float* hi, float* x, const int* m,
                      const uint64_t* pin, uint64_t* F, uint64_t* U, int* ok, int* iters, int max_iter, int zero_bounds, int keep_last) {
  return run<float>(n, mp, ext, mcap, A, b, lo, hi, x, m, pin, F, U, ok, iters, max_iter, zero_bounds, keep_last);
}
}
