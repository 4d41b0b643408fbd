// spatial_impl.hpp -- host side of the tree kernel: card -> SpatialModel (multi-dof joints expanded into 1-dof links), launches,
// contact report / external force / task state plumbing, and the dynamics-getter model.
// Included by spatial_f32.hip / spatial_f64.hip only.
#pragma once
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "impl_iface.hpp"
#include "spatial_kernel.hpp"
#include "spatial_build.hpp"

namespace dartk {

template <class Real>
struct SpatialImplT : Impl {
  SpatialImplT() { soa = false; }
  SpatialModel<Real> M;
  SpatialModel<Real>* dM = nullptr;
  Real* init_h = nullptr;
  size_t lds = 0;
  hipError_t prepare(int64_t n) override {
    hipError_t e;
    nenv = n;
    if ((e = hipMalloc((void**)&dM, sizeof(M))) != hipSuccess) return e;
    if ((e = hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&init_h, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    if ((e = hipMemset(init_h, 0, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    if (M.task == DART_TASK_WALKER3D_SPD) {   // per-env constraint forces of the last world step, read by the SPD law
      if ((e = hipMalloc((void**)&d_cf, sizeof(Real) * (size_t)n * M.n)) != hipSuccess) return e;
      if ((e = hipMemset(d_cf, 0, sizeof(Real) * (size_t)n * M.n)) != hipSuccess) return e;
      M.cf_store = d_cf;
      if ((e = hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    pairs = M.npairs > 0;
    big = M.n >= 20 && !M.free_root;
    pattern = matches_pattern<HumanWalkerPattern>() ? 1 : 0;
    is_static = pattern != 0 && uses_big() && !extras && !pairs;   // DART_Q_STATIC_KERNEL: the step kernel is specialised for this model's tree at compile time
    choose_lds();
    (void)hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice);
    if (n > 4096) (void)set_launch_order(1);   // more workgroups than the device holds at once: dispatch the expensive envs first (DART_CFG_LAUNCH_ORDER)
    const size_t lds_max = sp_lds_bytes(M.nl, M.n, sizeof(Real), M.maxm, M.maxcp, 0);
    // the 20+-dof models without a free root run the BIG instantiations (register LCP solver); measured on HumanWalker / Walker3d
    const void* fns[8] = {(const void*)sp_step_kernel<Real, false, false, false, false>, (const void*)sp_step_kernel<Real, false, false, false, true>,
                          (const void*)sp_step_kernel<Real, true, false, false, true>, (const void*)sp_step_kernel<Real, false, true, false, false>,
                          (const void*)sp_step_kernel<Real, true, true, false, true>, (const void*)sp_step_kernel<Real, true, true, true, false>,
                          (const void*)sp_step_kernel<Real, true, false, false, false>,
                          (const void*)sp_step_kernel<Real, false, false, false, true, HumanWalkerPattern>};
    for (const void* fn : fns)
      if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)sp_reset_kernel<Real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max)) != hipSuccess) return e;
    return hipSuccess;
  }
  void release() override {
    if (dM) (void)hipFree(dM); if (init_h) (void)hipFree(init_h); if (d_ext) (void)hipFree(d_ext); if (d_cf) (void)hipFree(d_cf); if (d_cost) (void)hipFree(d_cost); if (d_perm) (void)hipFree(d_perm);
    if (d_creport) (void)hipFree(d_creport); if (d_ccount) (void)hipFree(d_ccount); if (d_cfrep) (void)hipFree(d_cfrep); d_creport = nullptr; d_ccount = nullptr; d_cfrep = nullptr;
    dM = nullptr; init_h = nullptr; d_ext = nullptr; d_cf = nullptr; d_cost = nullptr; d_perm = nullptr;
    M.sched_cost = nullptr; M.sched_perm = nullptr; M.cf_store = nullptr;
  }
  // Sparsity of this model's mass-matrix factor in storage order (row n-1-d = dof d; symbolic elimination of the ancestor
  // relation), compared with a baked pattern: only an exact match may run that pattern's kernel.
  template <class PAT>
  bool matches_pattern() const {
    const int n = M.n;
    if (n != PAT::n || n > 32) return false;
    uint32_t low[32] = {};
    for (int i = 0; i < M.nl; i++) {
      const int d = M.dof[i];
      if (d < 0) continue;
      for (int j = M.parent[i]; j >= 0; j = M.parent[j]) {
        const int a = M.dof[j];
        if (a < 0 || a == d) continue;
        const int r = n - 1 - a, c = n - 1 - d;
        if (r > c) low[r] |= 1u << c; else low[c] |= 1u << r;
      }
    }
    for (int j = 0; j < n; j++)
      for (int a = j + 1; a < n; a++)
        if ((low[a] >> j) & 1u)
          for (int b = j + 1; b < a; b++) if ((low[b] >> j) & 1u) low[a] |= 1u << b;
    for (int i = 0; i < n; i++) if (low[i] != PAT::row(i)) return false;
    // the model constants the pattern kernel takes as compile-time facts (tree_patterns.hpp; sp_step_kernel assumes them)
    if (M.nl != PAT::nl || M.maxm != PAT::maxm || M.maxcp != PAT::maxcp || M.nshapes != PAT::nshapes || M.npairs != PAT::npairs ||
        M.free_root != PAT::free_root || M.task != PAT::task || M.frame_skip != PAT::frame_skip || M.act_dim != PAT::act_dim ||
        M.obs_dim != PAT::obs_dim || M.act_dof0 != PAT::act_dof0 || M.nrounds != PAT::nrounds || M.n_mpairs != PAT::n_mpairs ||
        M.impulse_M != PAT::impulse_M || M.has_joint_friction != PAT::has_joint_friction)
      return false;
    return true;
  }
  // register-LCP models without contact reporting drop the LDS solver's workspace (sp_carve): smaller block, more workgroups per CU
  bool uses_big() const {   // which step-kernel instantiation step() launches
    if (M.creport) return false;
    return pairs ? (extras || big) : (!extras && big);
  }
  bool launches_pattern_kernel() const { return !M.creport && !pairs && !extras && big && pattern == 1; }   // step()'s dispatch, restated
  void choose_lds() {
    M.reg_lcp = (uses_big() && M.npairs == 0 && M.maxm <= 40) ? 1 : 0;
    // layout of H in the block (round 5): the pattern kernel keeps a skyline of the structural entries (HumanWalker: 252 Reals instead of
    // the 576 of the padded rows -- with it the fp64 block drops under 20 KB = eight workgroups per CU), every other kernel the padded rows;
    // the mass-matrix entry list is addressed accordingly
    const bool sky = launches_pattern_kernel();
    M.hreals = sky ? HumanWalkerPattern::hreals : HR(sp_npad(M.n));
    const int n1 = M.n - 1;
    for (int e = 0; e < M.n_mpairs; e++) {
      const int d = (M.mpairs[e] >> 16) & 0xff, dj = M.mpairs[e] >> 24;
      const int a = n1 - d, b = n1 - dj, r = a > b ? a : b, c = a > b ? b : a;
      M.mpair_off[e] = (uint16_t)(sky ? HumanWalkerPattern::hbase(r) + c : HL(r, c));
    }
    lds = sp_lds_bytes(M.nl, M.n, sizeof(Real), M.maxm, M.maxcp, M.reg_lcp, M.hreals);
  }
  void upload() { if (dM) (void)hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice); }
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
#define SP_LAUNCH(P, X, R, B)                                                                                           \
  hipLaunchKernelGGL((sp_step_kernel<Real, P, X, R, B>), dim3((unsigned)n), dim3(64), lds, s, dM, n, (Real*)q, (Real*)dq, init_h, el, ep, \
                     act, obs, rew, done, trunc, autoreset, seed, off)
    if (M.creport) SP_LAUNCH(true, true, true, false);   // contact reporting lives in the most general instantiation only
    else if (pairs) { if (extras) SP_LAUNCH(true, true, false, true); else if (big) SP_LAUNCH(true, false, false, true); else SP_LAUNCH(true, false, false, false); }
    else if (extras) SP_LAUNCH(false, true, false, false);
    else if (launches_pattern_kernel())   // the factor's sparsity is known at compile time (tree_patterns.hpp)
      hipLaunchKernelGGL((sp_step_kernel<Real, false, false, false, true, HumanWalkerPattern>), dim3((unsigned)n), dim3(64), lds, s, dM, n, (Real*)q,
                         (Real*)dq, init_h, el, ep, act, obs, rew, done, trunc, autoreset, seed, off);
    else if (big) SP_LAUNCH(false, false, false, true);
    else SP_LAUNCH(false, false, false, false);
#undef SP_LAUNCH
    if (M.sched_perm) hipLaunchKernelGGL((sp_sched_kernel<Real>), dim3(1), dim3(1024), 0, s, n, d_cost, d_perm);
    return hipGetLastError();
  }
  hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                   const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) override {
    hipLaunchKernelGGL((sp_reset_kernel<Real>), dim3((unsigned)n), dim3(64), lds, s, dM, n, (Real*)q, (Real*)dq, init_h, el, ep,
                       mask, qn, vn, obs, seed, off, obs_masked_only);
    return hipGetLastError();
  }
  hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) override {
    int64_t count = n * M.n;
    hipLaunchKernelGGL((sp_state_io_kernel<Real>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, count, (Real*)q,
                       (Real*)dq, qh, dqh, to_device);
    return hipGetLastError();
  }
  void set_solver(int solver, int it1, int it2) override {   // one wavefront per env: a high cap only costs the hard envs
    // solver 1 = projected Gauss-Seidel only: no pivoting iterations, `it1` wave-level sweeps per stage (row dot products
    // reduced across the wavefront with __shfl_xor); solver 0 = pivoting with the PGS safety net
    (void)it2;
    if (solver == 1) { M.solver_iters = 0; M.pgs_fallback_sweeps = it1 > 0 ? it1 : 30; }
    else { M.solver_iters = it1 > 0 ? it1 : 600; M.pgs_fallback_sweeps = 600; }
    upload();
  }
  double* dbg = nullptr; int64_t nenv = 0;
  Real* d_ext = nullptr;
  Real* d_cf = nullptr;
  bool pairs = false, extras = false, big = false;   // which instantiation of the step kernel this model runs
  int pattern = 0;                                   // 1: HumanWalkerPattern (lean BIG kernel only)
  int body_link_map[DART_MAX_BODIES];
  int set_task_state(hipStream_t s, const uint8_t* d_mask, const double* d_values, int64_t n) override {
    hipLaunchKernelGGL((sp_task_state_kernel<Real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_mask, d_values, init_h);
    return hipGetLastError() == hipSuccess ? DART_OK : DART_E_HIP;
  }
  int set_ext_force(int body, const double* host_force, int64_t n) override {
    if (!extras) return DART_E_UNSUPPORTED;   // the lean kernel has no external-force code: card.generic_kernel = 1
    if (!host_force) { M.ext_force = nullptr; upload(); return DART_OK; }
    if (!d_ext && hipMalloc((void**)&d_ext, sizeof(Real) * 3 * (size_t)n) != hipSuccess) return DART_E_HIP;
    std::vector<Real> tmp(3 * (size_t)n);
    for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (Real)host_force[i];
    if (hipMemcpy(d_ext, tmp.data(), sizeof(Real) * tmp.size(), hipMemcpyHostToDevice) != hipSuccess) return DART_E_HIP;
    // a massless carrier body shares its origin with its group's joint origin: the force moves to the group leader
    const int lk = body_link_map[body];
    M.ext_link = M.group_leader[lk]; M.ext_at_joint_origin = M.group_leader[lk] != lk; M.ext_force = d_ext; upload();
    return DART_OK;
  }
  void set_stats(unsigned long long* p) override {
    M.stats = p;
    if (p && !dbg) { (void)hipMalloc((void**)&dbg, sizeof(double) * 160 * (size_t)nenv); (void)hipMemset(dbg, 0, sizeof(double) * 160 * (size_t)nenv); }
    M.dbg = p ? dbg : nullptr;
    upload();
  }
  // DART_CFG_LAUNCH_ORDER: per-env durations are recorded by the step kernel, sp_sched_kernel turns them into the dispatch order of the
  // next launch on the same stream.  (HumanWalker, 16 384 envs: 7.21 -> 6.72 ms fp32, 17.10 -> 16.27 ms fp64; DartDog fp64 4.20 -> 4.13.)
  int set_launch_order(int on) override {
    if (on && !(d_cost && d_perm)) {   // both or neither: a half-made pair (second allocation failed) is torn down, never enabled
      auto drop = [&]() { if (d_cost) (void)hipFree(d_cost); if (d_perm) (void)hipFree(d_perm); d_cost = nullptr; d_perm = nullptr; M.sched_cost = nullptr; M.sched_perm = nullptr; upload(); return -1; };
      if (d_cost || d_perm) (void)drop();
      if (hipMalloc((void**)&d_cost, 4 * (size_t)nenv) != hipSuccess) { d_cost = nullptr; return drop(); }
      if (hipMalloc((void**)&d_perm, 4 * (size_t)nenv) != hipSuccess) { d_perm = nullptr; return drop(); }
      (void)hipMemset(d_cost, 0, 4 * (size_t)nenv);
      std::vector<int> iota((size_t)nenv);
      for (int64_t i = 0; i < nenv; i++) iota[(size_t)i] = (int)i;
      if (hipMemcpy(d_perm, iota.data(), 4 * (size_t)nenv, hipMemcpyHostToDevice) != hipSuccess) return drop();
    }
    M.sched_cost = on ? d_cost : nullptr;
    M.sched_perm = on ? d_perm : nullptr;
    upload();
    return 0;
  }
  unsigned int* d_cost = nullptr; int* d_perm = nullptr;
  hipError_t debug_dump(double* out) override { return dbg ? hipMemcpy(out, dbg, sizeof(double) * 160 * (size_t)nenv, hipMemcpyDeviceToHost) : hipErrorInvalidValue; }
  int slots() const override { return M.maxm; }
  int max_contacts() const override { return M.maxcp; }
  int64_t lds_bytes() const override { return (int64_t)lds; }
  void persistent(std::vector<std::pair<void*, size_t>>& v, int64_t n) override {
    if (init_h) v.push_back({init_h, sizeof(Real) * 4 * (size_t)n});          // per-env task state (reach targets, initial head height)
    if (d_cf) v.push_back({d_cf, sizeof(Real) * (size_t)M.n * (size_t)n});     // SPD: constraint forces carried to the next step
  }
  Real* d_creport = nullptr; int* d_ccount = nullptr; Real* d_cfrep = nullptr;
  int set_contact_report(bool on, int64_t n) override {
    if (on && !d_creport) {
      if (hipMalloc((void**)&d_creport, sizeof(Real) * 8 * (size_t)M.maxcp * (size_t)n) != hipSuccess) return DART_E_HIP;
      if (hipMalloc((void**)&d_ccount, sizeof(int) * (size_t)n) != hipSuccess) return DART_E_HIP;
      (void)hipMemset(d_ccount, 0, sizeof(int) * (size_t)n);
      if (hipMalloc((void**)&d_cfrep, sizeof(Real) * (size_t)M.n * (size_t)n) != hipSuccess) return DART_E_HIP;
      (void)hipMemset(d_cfrep, 0, sizeof(Real) * (size_t)M.n * (size_t)n);
    }
    M.creport = on ? d_creport : nullptr; M.creport_count = on ? d_ccount : nullptr; M.cf_report = on ? d_cfrep : nullptr;
    choose_lds();
    upload();
    return DART_OK;
  }
  int get_contacts(hipStream_t s, int64_t n, int32_t* count, int32_t* bodies, double* point_force, int maxc) override {
    if (!M.creport) return DART_E_INVALID;
    if (hipStreamSynchronize(s) != hipSuccess) return DART_E_HIP;
    std::vector<Real> rec(8 * (size_t)M.maxcp * (size_t)n);
    std::vector<int> cnt((size_t)n);
    if (hipMemcpy(rec.data(), d_creport, sizeof(Real) * rec.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    if (hipMemcpy(cnt.data(), d_ccount, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    for (int64_t e = 0; e < n; e++) {
      const int k = cnt[(size_t)e] < maxc ? cnt[(size_t)e] : maxc;
      count[e] = cnt[(size_t)e];
      for (int c = 0; c < maxc; c++) {
        const Real* r = rec.data() + ((size_t)e * M.maxcp + c) * 8;
        const bool live = c < k;
        if (bodies) { bodies[((size_t)e * maxc + c) * 2] = live ? (int32_t)r[0] : -1; bodies[((size_t)e * maxc + c) * 2 + 1] = live ? (int32_t)r[1] : -1; }
        if (point_force) for (int a = 0; a < 6; a++) point_force[((size_t)e * maxc + c) * 6 + a] = live ? (double)r[2 + a] : 0.0;
      }
    }
    return DART_OK;
  }
  int get_constraint_forces(hipStream_t s, int64_t n, double* out) override {
    if (!M.cf_report) return DART_E_INVALID;
    if (hipStreamSynchronize(s) != hipSuccess) return DART_E_HIP;
    std::vector<Real> tmp((size_t)M.n * (size_t)n);
    if (hipMemcpy(tmp.data(), d_cfrep, sizeof(Real) * tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    for (size_t i = 0; i < tmp.size(); i++) out[i] = (double)tmp[i];
    return DART_OK;
  }
};

template <class Real>
std::unique_ptr<Impl> make_spatial(const DartModelCard& c, std::string& why) {
  auto p = std::make_unique<SpatialImplT<Real>>();
  std::string w = fill_spatial<Real>(c, p->M, false, p->body_link_map);
  p->extras = c.generic_kernel != 0 || c.task == DART_TASK_SNAKE || c.task == DART_TASK_WALKER3D_SPD || p->M.has_joint_friction != 0 ||
              p->M.free_root != 0;
  if (w.empty()) return p;
  why += w;
  return nullptr;
}

template <class Real>
int dyn_prepare(const DartModelCard& c, DynModel& out, std::string& err) {
  auto M = std::make_unique<SpatialModel<Real>>();
  std::string w = fill_spatial<Real>(c, *M, true);
  if (!w.empty()) { err = "dynamics getters: " + w; return DART_E_UNSUPPORTED; }
  out.free_root = M->free_root != 0;
  if (hipMalloc(&out.dev, sizeof(SpatialModel<Real>)) != hipSuccess) { err = "hipMalloc(dynamics model)"; return DART_E_HIP; }
  if (hipMemcpy(out.dev, M.get(), sizeof(SpatialModel<Real>), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy(dynamics model)"; return DART_E_HIP; }
  out.lds = sp_lds_bytes(M->nl, M->n, sizeof(Real), M->maxm, M->maxcp, 0);
  if (hipFuncSetAttribute((const void*)sp_dynamics_kernel<Real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)out.lds) != hipSuccess) {
    err = "hipFuncSetAttribute(sp_dynamics_kernel)"; return DART_E_HIP;
  }
  return DART_OK;
}
template <class Real>
hipError_t dyn_launch(hipStream_t s, const DynModel& m, int64_t n, const void* q, const void* dq, int soa, double* mass, double* bias,
                      double* pose, int nbodies) {
  hipLaunchKernelGGL((sp_dynamics_kernel<Real>), dim3((unsigned)n), dim3(64), m.lds, s, (const SpatialModel<Real>*)m.dev, n,
                     (const Real*)q, (const Real*)dq, soa, mass, bias, pose, nbodies);
  return hipGetLastError();
}

}  // namespace dartk
