// TEST INFRASTRUCTURE -- a stand-in for <hip/hip_runtime.h> that lets g++ compile the planar DEVICE code
// (dart_env_amd/csrc/planar_kernel.hpp + planar_impl.hpp) for the host, one lane at a time.
//
// Why: the development container has no GPU; this makes the arithmetic of the one-env-per-lane kernels checkable against
// the fp64 oracle with `pytest -m "not gpu"`.  A lane of those kernels never reads another lane's data (wave votes only
// decide when a loop stops), so running the lanes one after the other is the same computation.
// Nothing under dart_env_amd/ includes or loads this; the product library is built by hipcc against the real header and
// has no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define DART_PIN_VGPR(x) asm volatile("" : "+g"(x))   // the device build pins the value in a VGPR

using std::fabs; using std::fmax; using std::fmin; using std::isfinite; using std::sqrt;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }
// "device" memory is host memory here
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

inline bool __all(bool p) { return p; }
// a lane's wave neighbours may vote a lane into the larger tier / the friction stage / the constraint phase although it
// has nothing there itself: DART_EMU_ANY=1 makes every __any() vote true so that those paths are exercised lane by lane
inline bool emu_any_true_() { static const bool v = getenv("DART_EMU_ANY") && getenv("DART_EMU_ANY")[0] == '1'; return v; }
inline bool __any(bool p) { return p || emu_any_true_(); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
// hardware approximations: model them a few bits short of exact so the Newton refinements in the kernels are exercised
inline float __builtin_amdgcn_rcpf(float x) { return (1.0f / x) * (1.0f + 5.9e-8f); }
inline float __builtin_amdgcn_rsqf(float x) { return (1.0f / sqrtf(x)) * (1.0f - 5.9e-8f); }
inline double __builtin_amdgcn_rcp(double x) { return (double)(float)(1.0 / x); }
inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }

// a kernel launch = the kernel body called for every (block, thread) in turn
template <class F> inline void emu_launch_(dim3 g, dim3 b, F&& f) {
  gridDim = g; blockDim = b;
  for (unsigned bx = 0; bx < g.x; bx++) for (unsigned tx = 0; tx < b.x; tx++) { blockIdx.x = bx; threadIdx.x = tx; f(); }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch_(grid, block, [&]() { kernel(__VA_ARGS__); })
