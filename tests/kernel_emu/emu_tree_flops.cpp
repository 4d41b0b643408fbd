// (tree-kernel variant of emu_flops.cpp: the same counting scalar on the fiber runtime; counts are summed over the 64 lanes of a wave)
// TEST / MEASUREMENT INFRASTRUCTURE -- the planar register kernels compiled for the host with a COUNTING scalar type: every
// arithmetic operation the device code performs on a `Real` increments a counter, so one env-step's floating-point work is
// counted exactly (per lane, i.e. per environment), not estimated from operation-count formulas.
//   flops = add + sub + mul + 2 fma + div + sqrt + rcp + rsq      (v_rcp / v_rsq / sqrt / div count 1 each)
// compares, min / max / abs, negations and conversions are counted separately (they issue VALU instructions but are no flops).
// The device code is the same source as the product kernels (dart_env_amd/csrc/planar_kernel.hpp via planar_impl.hpp): the
// counting type behaves like `double` (sizeof 8 -> the fp64 tiers, tolerances and sincos polynomial), so the counts are those
// of the fp64 instantiation; the fp32 one differs only in its shorter sincos polynomial and single Newton steps (noted by
// tools/count_flops.py).  A lane's pivoting loops stop on the lane's OWN convergence here; on the GPU a wave iterates until its
// slowest lane is done, so the device EXECUTES more than this algorithmic count.
// Built by tests/kernel_emu/Makefile into libdart_planar_flops.so; loaded only by tools/count_flops.py and tests/.
#include <hip/hip_runtime.h>   // the stand-in of fake_wave_include/ (fibers)

#undef DART_PIN_VGPR
#define DART_PIN_VGPR(x) ((void)0)

struct FlopCounters { unsigned long long add, mul, fma, div, sqrt, rcp, rsq, cmp, minmax, abs, neg, cvt; };
inline FlopCounters g_fc;

struct CReal {
  double v;
  CReal() = default;
  constexpr CReal(double x) : v(x) {}
  constexpr CReal(float x) : v(x) {}
  constexpr CReal(int x) : v(x) {}
  constexpr CReal(unsigned x) : v(x) {}
  constexpr CReal(long x) : v((double)x) {}
  constexpr CReal(unsigned long x) : v((double)x) {}
  constexpr CReal(long long x) : v((double)x) {}
  constexpr CReal(unsigned long long x) : v((double)x) {}
  explicit operator double() const { g_fc.cvt++; return v; }
  explicit operator float() const { g_fc.cvt++; return (float)v; }
  explicit operator int() const { g_fc.cvt++; return (int)v; }
  explicit operator bool() const { return v != 0; }
};
inline CReal operator+(CReal a, CReal b) { g_fc.add++; return CReal(a.v + b.v); }
inline CReal operator-(CReal a, CReal b) { g_fc.add++; return CReal(a.v - b.v); }
inline CReal operator*(CReal a, CReal b) { g_fc.mul++; return CReal(a.v * b.v); }
inline CReal operator/(CReal a, CReal b) { g_fc.div++; return CReal(a.v / b.v); }
// (constexpr: the baked models' `-Real(inf)` literals and their compile-time `== Real(0)` tests go through these two)
constexpr CReal operator-(CReal a) { if (!__builtin_is_constant_evaluated()) g_fc.neg++; return CReal(-a.v); }
inline CReal operator+(CReal a) { return a; }
inline CReal& operator+=(CReal& a, CReal b) { g_fc.add++; a.v += b.v; return a; }
inline CReal& operator-=(CReal& a, CReal b) { g_fc.add++; a.v -= b.v; return a; }
inline CReal& operator*=(CReal& a, CReal b) { g_fc.mul++; a.v *= b.v; return a; }
inline CReal& operator/=(CReal& a, CReal b) { g_fc.div++; a.v /= b.v; return a; }
#define CREAL_CMP(op) inline bool operator op(CReal a, CReal b) { g_fc.cmp++; return a.v op b.v; }
CREAL_CMP(<) CREAL_CMP(<=) CREAL_CMP(>) CREAL_CMP(>=) CREAL_CMP(!=)
constexpr bool operator==(CReal a, CReal b) { if (!__builtin_is_constant_evaluated()) g_fc.cmp++; return a.v == b.v; }
inline CReal fabs(CReal a) { g_fc.abs++; return CReal(std::fabs(a.v)); }
inline CReal fmax(CReal a, CReal b) { g_fc.minmax++; return CReal(std::fmax(a.v, b.v)); }
inline CReal fmin(CReal a, CReal b) { g_fc.minmax++; return CReal(std::fmin(a.v, b.v)); }
inline CReal sqrt(CReal a) { g_fc.sqrt++; return CReal(std::sqrt(a.v)); }
inline bool isfinite(CReal a) { g_fc.cmp++; return std::isfinite(a.v); }
inline CReal cfma(CReal a, CReal b, CReal c) { g_fc.fma++; return CReal(std::fma(a.v, b.v, c.v)); }
inline CReal fma(CReal a, CReal b, CReal c) { return cfma(a, b, c); }

#include "planar_kernel.hpp"

namespace dartk {
// the fp64 instantiation's sincos / rcp / rsqrt (planar_kernel.hpp), operation for operation
template <> __device__ __forceinline__ void sincos_<CReal>(CReal x, CReal& s, CReal& c) {
  g_fc.mul++; g_fc.cvt += 2;                              // x * 2/pi, rint, (int)
  const double kd = std::rint(x.v * 6.36619772367581382433e-01);
  const int k = (int)kd;
  const CReal kf(kd);
  CReal r = cfma(kf, CReal(-1.57079632673412561417e+00), x);
  r = cfma(kf, CReal(-6.07710050630396597660e-11), r);
  r = cfma(kf, CReal(-2.02226624871116645580e-21), r);
  r = cfma(kf, CReal(-8.47842766036889956997e-32), r);
  const CReal z = r * r;
  CReal ps = cfma(z, CReal(1.58969099521155010221e-10), CReal(-2.50507602534068634195e-08));
  ps = cfma(ps, z, CReal(2.75573137070700676789e-06));
  ps = cfma(ps, z, CReal(-1.98412698298579493134e-04));
  ps = cfma(ps, z, CReal(8.33333333332248946124e-03));
  ps = cfma(ps, z, CReal(-1.66666666666666324348e-01));
  const CReal sn = cfma(ps * z, r, r);
  CReal pc = cfma(z, CReal(-1.13596475577881948265e-11), CReal(2.08757232129817482790e-09));
  pc = cfma(pc, z, CReal(-2.75573143513906633035e-07));
  pc = cfma(pc, z, CReal(2.48015872894767294178e-05));
  pc = cfma(pc, z, CReal(-1.38888888888741095749e-03));
  pc = cfma(pc, z, CReal(4.16666666666666019037e-02));
  const CReal cs = cfma(pc * z, z, cfma(z, CReal(-0.5), CReal(1.0)));
  const bool swap = k & 1;
  const CReal s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
template <> __device__ __forceinline__ CReal rcp_<CReal>(CReal x) {   // v_rcp_f64 + two Newton steps
  g_fc.rcp++;
  CReal r(1.0 / x.v);
  r = cfma(cfma(-x, r, CReal(1.0)), r, r);
  return cfma(cfma(-x, r, CReal(1.0)), r, r);
}
template <> __device__ __forceinline__ CReal rsq_seed_<CReal>(CReal x) { g_fc.rsq++; return CReal(1.0 / std::sqrt(x.v)); }
template <> __device__ __forceinline__ CReal rsqrt_<CReal>(CReal x) {   // v_rsq_f64 + two Newton steps
  g_fc.rsq++;
  CReal r(1.0 / std::sqrt(x.v));
  r = r * cfma(CReal(-0.5) * x * r, r, CReal(1.5));
  return r * cfma(CReal(-0.5) * x * r, r, CReal(1.5));
}
}  // namespace dartk

#include <memory>
#include <string>
#include <vector>

// the kernels' phase stopwatch (SP_TICK: __builtin_readcyclecounter deltas per phase into S.ticks, with DART_CFG_STATS) reads the running
// flop total here, so the per-phase "cycles" of the debug counters are flops per phase
#undef __builtin_readcyclecounter
inline unsigned long long flops_now_() { return g_fc.add + g_fc.mul + 2 * g_fc.fma + g_fc.div + g_fc.sqrt + g_fc.rcp + g_fc.rsq; }
#define __builtin_readcyclecounter() flops_now_()

// libm calls of the task code (angles of the done conditions, the free root's exponential map): one call each, counted as `div`-class work
inline CReal acos(CReal a) { g_fc.div++; return CReal(std::acos(a.v)); }
inline CReal atan2(CReal a, CReal b) { g_fc.div++; return CReal(std::atan2(a.v, b.v)); }
inline CReal asin(CReal a) { g_fc.div++; return CReal(std::asin(a.v)); }
inline CReal sin(CReal a) { g_fc.div++; return CReal(std::sin(a.v)); }
inline CReal cos(CReal a) { g_fc.div++; return CReal(std::cos(a.v)); }
inline CReal exp(CReal a) { g_fc.div++; return CReal(std::exp(a.v)); }
struct CReal2 { CReal x, y; };
#include "spatial_model.hpp"
namespace dartk { template <> struct sp_vec128<CReal> { using type = CReal2; static constexpr int width = 2; }; }

// cross-lane traffic of the counting scalar: the value travels, nothing is counted (v_readlane / DPP moves are no flops)
namespace dartk {
template <> __device__ __forceinline__ CReal readlane_<CReal>(CReal x, int l) { return CReal(readlane_<double>(x.v, l)); }
}
template <int CTRL, int ROW_MASK> inline CReal dpp_add_(CReal v) { g_fc.add++; return CReal(dartk::dpp_add_<CTRL, ROW_MASK>(v.v)); }
template <int CTRL, int ROW_MASK> inline CReal dpp_max_(CReal v) { g_fc.minmax++; return CReal(dartk::dpp_max_<CTRL, ROW_MASK>(v.v)); }

#include "spatial_impl.hpp"

using namespace dartk;
namespace dartk {
std::unique_ptr<Impl> make_planar_impl_f32(const DartModelCard&, std::string&, bool) { return nullptr; }
std::unique_ptr<Impl> make_planar_impl_f64(const DartModelCard&, std::string&, bool) { return nullptr; }
}

struct FlopEmu {
  DartModelCard card;
  int64_t n;
  std::unique_ptr<Impl> impl;
  std::vector<unsigned char> q, dq;
  std::vector<int32_t> elapsed; std::vector<uint32_t> episode;
  std::vector<unsigned long long> stats;
};

extern "C" {
void flops_enable_stats(FlopEmu* h, int on) { h->stats.assign(64, 0); h->impl->set_stats(on ? h->stats.data() : nullptr); }
void flops_get_stats(FlopEmu* h, unsigned long long* out64) { memcpy(out64, h->stats.data(), 64 * sizeof(unsigned long long)); }
FlopEmu* flops_create(const DartModelCard* card, int64_t n, int, char* why_out, int why_len) {
  std::string why;
  auto impl = make_spatial<CReal>(*card, why);
  if (!impl) { if (why_out) snprintf(why_out, why_len, "%s", why.c_str()); return nullptr; }
  auto* h = new FlopEmu();
  h->card = *card; h->n = n; h->impl = std::move(impl);
  h->impl->set_solver(0, 0, 0);
  (void)h->impl->prepare(n);
  const size_t nd = card->ndofs;
  h->q.assign(sizeof(CReal) * nd * n, 0); h->dq.assign(sizeof(CReal) * nd * n, 0);
  h->elapsed.assign(n, 0); h->episode.assign(n, 0);
  return h;
}
void flops_destroy(FlopEmu* h) { if (h) { h->impl->release(); delete h; } }
int flops_is_static(FlopEmu* h) { return h->impl->is_static ? 1 : 0; }
void flops_reset(FlopEmu* h, float* obs, uint64_t seed, uint64_t off) {
  h->impl->reset(nullptr, h->n, h->q.data(), h->dq.data(), h->elapsed.data(), h->episode.data(), nullptr, nullptr, nullptr, obs, seed, off, 0);
}
void flops_step(FlopEmu* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc, uint64_t seed, uint64_t off) {
  h->impl->step(nullptr, h->n, h->q.data(), h->dq.data(), h->elapsed.data(), h->episode.data(), actions, obs, rew, done, trunc, 1, seed, off);
}
void flops_clear() { g_fc = FlopCounters(); }
void flops_read(unsigned long long* out12) { memcpy(out12, &g_fc, sizeof g_fc); }
}
