// TEST INFRASTRUCTURE: dartk::log_cr and the legacy-Gaussian factor f = sqrt(-2 log(r2) / r2) evaluated ON THE DEVICE for a host array
// (tests/test_gpu_cr_log.py compares them with the host libm / decimal arithmetic).  Built by __graft_entry__.build().
#include <hip/hip_runtime.h>
#include "cr_log.hpp"
__global__ void crlog_kernel(const double* __restrict__ x, double* __restrict__ lg, double* __restrict__ f, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r2 = x[i];
  const double l = dartk::log_cr(r2);
  lg[i] = l;
  {
#pragma clang fp contract(off)
    const double num = -2.0 * l;
    f[i] = sqrt(num / r2);
  }
}
extern "C" int crlog_run(const double* x, double* lg, double* f, long n) {
  double *dx = nullptr, *dl = nullptr, *df = nullptr;
  if (hipMalloc((void**)&dx, 8 * n) != hipSuccess || hipMalloc((void**)&dl, 8 * n) != hipSuccess || hipMalloc((void**)&df, 8 * n) != hipSuccess) return -1;
  hipMemcpy(dx, x, 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(crlog_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dl, df, n);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  hipMemcpy(lg, dl, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(f, df, 8 * n, hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dl); hipFree(df);
  return 0;
}
