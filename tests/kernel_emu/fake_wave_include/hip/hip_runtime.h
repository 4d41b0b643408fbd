// TEST INFRASTRUCTURE -- a stand-in for <hip/hip_runtime.h> that lets g++ compile the WAVE-COOPERATIVE device code (the tree kernel:
// dart_env_amd/csrc/spatial_*.hpp + wave_blcp.hpp) for the host.  A workgroup is one 64-lane wavefront whose lanes share LDS and talk
// through barriers, ballots, shuffles, v_readlane and DPP, so the lanes cannot be run one after the other as fake_include/ does for the
// one-env-per-lane kernels: here every lane of a workgroup is a FIBER (its own stack, switched by a dozen instructions), and every
// cross-lane operation is a rendezvous -- each lane deposits its operand and yields; when all live lanes of the wave have arrived the
// results are computed as the hardware defines them and the lanes resume.  That is exact for code whose cross-lane operations are
// executed by all live lanes, and it follows the hardware's exec-mask semantics for the two divergent idioms the kernels use: a shuffle
// / ballot / readlane inside a lane-divergent branch is resolved among the lanes that reach it while the others wait at their next
// barrier (the reconvergence point), and lanes waiting at different barriers are released minority first (a single-wave workgroup's
// s_barrier never blocks; the lanes inside the divergent region have to catch up).  Lanes waiting at two different non-barrier
// operations abort with a message.  Between two cross-lane operations a lane runs alone, so "every lane reads x, then one lane
// overwrites it" -- safe in lock-step -- needs DART_LOCKSTEP_FENCE() in the kernel source (defined away in the device build).
// Nothing under dart_env_amd/ includes or loads this; the product library is built by hipcc against the real header and has no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__
#define DART_PIN_VGPR(x) asm volatile("" : "+g"(x))
#define DART_WAVE_EMU 1

using std::fabs; using std::fmax; using std::fmin; using std::isfinite; using std::sqrt;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }   // HIP's typed overload
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline float __builtin_amdgcn_rcpf(float x) { return (1.0f / x) * (1.0f + 5.9e-8f); }
inline float __builtin_amdgcn_rsqf(float x) { return (1.0f / sqrtf(x)) * (1.0f - 5.9e-8f); }
inline double __builtin_amdgcn_rcp(double x) { return (double)(float)(1.0 / x); }
inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
#define __builtin_readcyclecounter() 0ull

// ------------------------------------------------------------------ fibers: one per lane of the running workgroup
namespace wave_emu {
constexpr int MAXL = 1024;
#ifndef WAVE_EMU_STACK   // per lane of a 64-lane workgroup (the step kernels' unrolled register arrays; the counting-scalar build at -O1 needs > 1 MB)
#define WAVE_EMU_STACK (4u << 20)
#endif
inline size_t stack_bytes = 0;   // of the fibers as allocated: WAVE_EMU_STACK for a wavefront, 256 KB per lane for the plain wide kernels
struct Fiber { void* sp = nullptr; char* stack = nullptr; size_t cap = 0; bool done = true; };
inline Fiber fib[MAXL];
inline void* sched_sp = nullptr;
inline int cur = -1, nl = 0;
inline void (*body)() = nullptr;
// x86-64 System V context switch: push the callee-saved registers, swap the stack pointer, pop them
extern "C" void wave_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wave_emu_switch
.type wave_emu_switch,@function
wave_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wave_emu_switch,.-wave_emu_switch
)");
inline void yield() { const int me = cur; wave_emu_switch(&fib[me].sp, sched_sp); }
extern "C" inline void wave_emu_entry() {
  body();
  fib[cur].done = true;
  yield();
  abort();
}
inline void make_fiber(int l) {
  Fiber& f = fib[l];
  const size_t STACK = stack_bytes;
  if (f.cap < STACK) { free(f.stack); f.stack = (char*)aligned_alloc(64, STACK); f.cap = STACK; }
  f.done = false;
  // initial frame: [mxcsr/fpcw slot][r15 r14 r13 r12 rbx rbp][return address = entry][alignment]
  uintptr_t top = ((uintptr_t)(f.stack + f.cap)) & ~(uintptr_t)63;
  uint64_t* s = (uint64_t*)top;
  *--s = 0;                                  // keeps the stack 16-byte aligned at the entry's first instruction (as after a call)
  *--s = (uint64_t)(uintptr_t)&wave_emu_entry;
  for (int i = 0; i < 6; i++) *--s = 0;      // rbp rbx r12 r13 r14 r15
  uint32_t csr; uint16_t cw;
  asm volatile("stmxcsr %0" : "=m"(csr)); asm volatile("fnstcw %0" : "=m"(cw));
  --s; ((uint32_t*)s)[0] = csr; ((uint32_t*)s)[1] = cw;
  f.sp = s;
}
// ---- rendezvous state of the wave
enum Op { NONE = 0, BARRIER, BALLOT, ANY, SHFL, SHFLX, READLANE, DPP };
inline int op_of[MAXL], line_of[MAXL];
inline const char* file_of[MAXL];
inline uint64_t val[MAXL], aux[MAXL], res[MAXL];
struct DppArgs { int ctrl, row_mask, bank_mask, bound; uint32_t old; };
inline DppArgs dpp[MAXL];
inline uint64_t rendezvous(int op, uint64_t v, uint64_t a, const char* file = "", int line = 0) {
  op_of[cur] = op; val[cur] = v; aux[cur] = a; file_of[cur] = file; line_of[cur] = line;
  yield();
  return res[cur];
}
inline bool parked[MAXL];
inline void resolve(const dim3& bd) {
  // who waits where: barrier waiters (by call site) and the lanes at another cross-lane operation
  int op = NONE, first = -1, nbar = 0, nx = 0;
  for (int l = 0; l < nl; l++) if (!fib[l].done) {
    if (op_of[l] == BARRIER) { nbar++; continue; }
    nx++;
    if (op == NONE) { op = op_of[l]; first = l; }
    else if (op_of[l] != op) {
      fprintf(stderr, "wave_emu: lanes of one wave wait at different cross-lane operations: lane %d at op %d, lane %d at op %d\n", first, op, l, op_of[l]);
      abort();
    }
  }
  if (nx == 0) {   // everybody at a barrier: all at the same one -> release; else the minority call sites first (they are inside a divergent region)
    int best_line = -1, best_cnt = -1;
    for (int l = 0; l < nl; l++) if (!fib[l].done) {
      int c = 0;
      for (int k = 0; k < nl; k++) if (!fib[k].done && line_of[k] == line_of[l] && file_of[k] == file_of[l]) c++;
      if (c > best_cnt) { best_cnt = c; best_line = l; }
    }
    const bool same = best_cnt == nbar;
    for (int l = 0; l < nl; l++) if (!fib[l].done)
      parked[l] = same ? false : (line_of[l] == line_of[best_line] && file_of[l] == file_of[best_line]);
    return;
  }
  // a non-barrier operation: resolved among the lanes that reached it; barrier waiters stay where they are
  for (int l = 0; l < nl; l++) if (!fib[l].done) parked[l] = (op_of[l] == BARRIER);
  for (int w0 = 0; w0 < nl; w0 += 64) {            // per wavefront of the workgroup
    const int w1 = w0 + 64 < nl ? w0 + 64 : nl;
    uint64_t ballot = 0; bool any = false;
    for (int l = w0; l < w1; l++) if (!fib[l].done && !parked[l] && val[l]) { ballot |= 1ull << (l - w0); any = true; }
    for (int l = w0; l < w1; l++) {
      if (fib[l].done || parked[l]) continue;
      const int i = l - w0;
      switch (op) {
        case BALLOT: res[l] = ballot; break;
        case ANY: res[l] = any; break;
        case SHFL: res[l] = val[w0 + ((int)aux[l] & 63)]; break;
        case SHFLX: res[l] = val[w0 + ((i ^ (int)aux[l]) & 63)]; break;
        case READLANE: res[l] = val[w0 + ((int)aux[l] & 63)]; break;
        case DPP: {
          const DppArgs& d = dpp[l];
          const int row = i >> 4, pos = i & 15;
          int src = -1;                             // lane (within the wave) whose value this lane reads, -1: none
          if (d.ctrl >= 0x111 && d.ctrl <= 0x11f) { const int sh = d.ctrl - 0x110; src = pos - sh >= 0 ? i - sh : -1; }   // row_shr:n
          else if (d.ctrl == 0x142) src = (pos >= 0 && row >= 1) ? ((row - 1) << 4 | 15) : -1;   // row_bcast:15 -> next row
          else if (d.ctrl == 0x143) src = row >= 2 ? 31 : -1;                                 // row_bcast:31 -> rows 2, 3
          else if (d.ctrl >= 0x150 && d.ctrl <= 0x15f) src = (row << 4) | (d.ctrl - 0x150);       // row_newbcast:n (gfx90a+): lane n of the lane's own row
          else { fprintf(stderr, "wave_emu: DPP control 0x%x not modelled\n", d.ctrl); abort(); }
          const bool row_en = (d.row_mask >> row) & 1, bank_en = (d.bank_mask >> (pos >> 2)) & 1;
          uint32_t r = d.old;
          if (row_en && bank_en) {
            if (src >= 0) r = (uint32_t)val[w0 + src];
            else if (d.bound) r = 0u;
          }
          res[l] = r;
        } break;
        default: break;
      }
    }
  }
}
// the static __shared__ arrays of the kernels (function-local statics here): registered by the test from the library's symbol table
// (tests/emu_lib.py: poison_static_lds) so that DART_EMU_POISON_LDS reaches them too
inline unsigned char* static_lds[256];
inline size_t static_lds_bytes[256];
inline int n_static_lds = 0;
inline unsigned char* dyn_lds = nullptr;
inline size_t dyn_lds_cap = 0;
template <class F> inline void launch(dim3 g, dim3 b, size_t lds, F&& f) {
  static thread_local std::remove_reference_t<F>* fp;
  gridDim = g; blockDim = b;
  if (lds > dyn_lds_cap) { free(dyn_lds); dyn_lds = (unsigned char*)aligned_alloc(64, (lds + 63) & ~(size_t)63); dyn_lds_cap = lds; }
  nl = (int)b.x;
  if (nl > MAXL) abort();
  stack_bytes = nl <= 64 ? (size_t)WAVE_EMU_STACK : (size_t)(256u << 10);
  fp = &f;
  body = +[]() { (*fp)(); };
  // DART_EMU_POISON_LDS=<byte>: the dynamic LDS block is filled with that byte before every workgroup (0xff: NaNs / -1) -- a kernel
  // whose results depend on LDS it has not written shows it (tests/test_tree_kernel_emu_parity.py; the device hands a workgroup whatever
  // the previous one left behind)
  static const char* poison = getenv("DART_EMU_POISON_LDS");
  for (unsigned bx = 0; bx < g.x; bx++) {
    blockIdx.x = bx;
    if (poison && lds) memset(dyn_lds, (int)strtol(poison, nullptr, 0), lds);
    if (poison) for (int r = 0; r < n_static_lds; r++) memset(static_lds[r], (int)strtol(poison, nullptr, 0), static_lds_bytes[r]);
    for (int l = 0; l < nl; l++) { make_fiber(l); op_of[l] = NONE; parked[l] = false; }
    bool alive = true;
    while (alive) {
      alive = false;
      for (int l = 0; l < nl; l++) {
        if (fib[l].done) continue;
        alive = true;
        if (parked[l]) continue;
        cur = l; threadIdx.x = (unsigned)l; op_of[l] = NONE;
        wave_emu_switch(&sched_sp, fib[l].sp);
      }
      alive = false;
      for (int l = 0; l < nl; l++) if (!fib[l].done) alive = true;
      if (alive) resolve(b);
    }
  }
  cur = -1;
}
}  // namespace wave_emu

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) wave_emu::launch(grid, block, (size_t)(lds), [&]() { kernel(__VA_ARGS__); })
// the kernels' dynamic LDS block (`extern __shared__ ... name[]` in the device build)
#define DART_DYNAMIC_LDS(name) unsigned char* name = wave_emu::dyn_lds

#define __syncthreads() wave_emu::rendezvous(wave_emu::BARRIER, 0, 0, __FILE__, __LINE__)
#define DART_LOCKSTEP_FENCE() __syncthreads()   // see spatial_dense.hpp
inline unsigned long long __ballot(bool p) { return wave_emu::rendezvous(wave_emu::BALLOT, p ? 1 : 0, 0); }
inline bool __any(bool p) { return wave_emu::rendezvous(wave_emu::ANY, p ? 1 : 0, 0) != 0; }
inline bool __all(bool p) { return wave_emu::rendezvous(wave_emu::ANY, p ? 0 : 1, 0) == 0; }
template <class T> inline uint64_t wave_bits_(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T wave_from_(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
template <class T> inline T __shfl(T v, int src) { return wave_from_<T>(wave_emu::rendezvous(wave_emu::SHFL, wave_bits_(v), (uint64_t)src)); }
template <class T> inline T __shfl_xor(T v, int mask) { return wave_from_<T>(wave_emu::rendezvous(wave_emu::SHFLX, wave_bits_(v), (uint64_t)mask)); }
inline int __builtin_amdgcn_readlane(int v, int l) { return (int)(uint32_t)wave_emu::rendezvous(wave_emu::READLANE, (uint32_t)v, (uint64_t)l); }
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  wave_emu::dpp[wave_emu::cur] = {ctrl, row_mask, bank_mask, bound_ctrl ? 1 : 0, (uint32_t)old};
  return (int)(uint32_t)wave_emu::rendezvous(wave_emu::DPP, (uint32_t)src, 0);
}
