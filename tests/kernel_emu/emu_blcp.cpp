// TEST INFRASTRUCTURE -- the lane kernels' boxed-LCP solver (blcp_bpp / blcp_bpp_mixed in dart_env_amd/csrc/planar_kernel.hpp: one problem
// per lane, block principal pivoting on a packed Delassus matrix in registers) compiled for the host and exposed on its own, so that
// tests/test_lane_blcp.py can hold it against the complementarity conditions and an enumeration of active sets without any physics
// around it.  Built by tests/kernel_emu/Makefile into libdart_lane_blcp.so; nothing under dart_env_amd/ loads it.
#include <stdint.h>

#include "planar_kernel.hpp"

using namespace dartk;

template <class Real, int M, bool ZB, bool PRE>
static void solve_all(int n, const double* A, const double* b, const double* lo, const double* hi, const uint32_t* pin, uint32_t* F, uint32_t* U,
                      double* x, int max_iter) {
  constexpr int TR = M * (M + 1) / 2;
  for (int p = 0; p < n; p++) {
    threadIdx.x = (unsigned)(p & 63);
    Real Ar[TR], br[M], lor[M], hir[M], xr[M];
    for (int k = 0; k < TR; k++) Ar[k] = (Real)A[(size_t)p * TR + k];
    for (int i = 0; i < M; i++) { br[i] = (Real)b[(size_t)p * M + i]; lor[i] = (Real)lo[(size_t)p * M + i]; hir[i] = (Real)hi[(size_t)p * M + i]; xr[i] = Real(0); }
    uint32_t f = F[p], u = U[p];
    blcp_bpp_mixed<Real, M, ZB, PRE>(Ar, br, lor, hir, pin[p], f, u, xr, max_iter, nullptr, Real(0), nullptr, 0, 6);
    F[p] = f; U[p] = u;
    for (int i = 0; i < M; i++) x[(size_t)p * M + i] = (double)xr[i];
  }
}

extern "C" {
// n problems of exactly M rows (M in {5, 7, 10, 14}: the row counts of the Hopper / Walker2d / half-cheetah tiers); A packed lower
// triangle per problem; real: 64 / 32; zero_bounds: the frictionless stage's shortcut; presolve32: the fp64 big tier's fp32 search first.
int lane_blcp_run(int n, int M, int real, int zero_bounds, int presolve32, const double* A, const double* b, const double* lo, const double* hi,
                  const uint32_t* pin, uint32_t* F, uint32_t* U, double* x, int max_iter) {
#define CASE(R, MM)                                                                                              \
  if (M == MM) {                                                                                                 \
    if (zero_bounds) { if (presolve32) solve_all<R, MM, true, true>(n, A, b, lo, hi, pin, F, U, x, max_iter);    \
                       else solve_all<R, MM, true, false>(n, A, b, lo, hi, pin, F, U, x, max_iter); }            \
    else { if (presolve32) solve_all<R, MM, false, true>(n, A, b, lo, hi, pin, F, U, x, max_iter);               \
           else solve_all<R, MM, false, false>(n, A, b, lo, hi, pin, F, U, x, max_iter); }                       \
    return 0;                                                                                                    \
  }
  if (real == 64) { CASE(double, 5) CASE(double, 7) CASE(double, 10) CASE(double, 14) }
  else { CASE(float, 5) CASE(float, 7) CASE(float, 10) CASE(float, 14) }
#undef CASE
  return -1;
}
}
