// planar_impl.hpp -- host side of the planar register kernels: card validation -> kernel parameters, launches.
// Included by planar_f32.hip / planar_f64.hip only (each instantiates the kernels for one precision).
#pragma once
#include <cmath>
#ifndef DART_BPP_DEFAULT_ITERS
#define DART_BPP_DEFAULT_ITERS 200   // iteration cap of the lane kernels' pivoting LCP solver (see PlanarImpl::set_solver)
#endif
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "impl_iface.hpp"
#include "planar_kernel.hpp"
#include "static_models.hpp"
#include "cart_kernel.hpp"
#include "arm_kernel.hpp"
#include "chain3d_kernel.hpp"
#include "spatial_build.hpp"

namespace dartk {

template <class Real, class T, class PT = Params<Real, T>>
struct ImplT : Impl {
  PT P;
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
    dim3 grid((unsigned)((n + block_threads - 1) / block_threads)), block(block_threads);
    if (P.ex.ext_force != nullptr || P.ex.creport != nullptr)
      hipLaunchKernelGGL((step_kernel<Real, T, PT, true>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew,
                         done, trunc, autoreset, seed, off);
    else
      hipLaunchKernelGGL((step_kernel<Real, T, PT, false>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew,
                         done, trunc, autoreset, seed, off);
    return hipGetLastError();
  }
  hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                   const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((reset_kernel<Real, T, PT>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, mask, qn, vn, obs,
                       seed, off, obs_masked_only);
    return hipGetLastError();
  }
  hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((state_io_kernel<Real, T::NDOF>), grid, block, 0, s, n, (Real*)q, (Real*)dq, qh, dqh, to_device);
    return hipGetLastError();
  }
  void set_solver(int solver, int it1, int it2) override {   // 0 = default cap
    // default caps: pivoting 200 (the loop ends on the wave's vote, so the cap costs nothing until it is needed -- and it is: with the
    // impulse pass on M (A3) 2 of 4 096 half-cheetah envs hit the old cap of 24 within 10 env-steps and left the oracle's
    // trajectory by O(1); every one of them converges within 60), PGS 30 sweeps
    const int dflt = solver == 0 ? DART_BPP_DEFAULT_ITERS : 30;
    P.solver = solver; P.iters1 = it1 > 0 ? it1 : dflt; P.iters2 = it2 > 0 ? it2 : dflt;
  }
  void set_stats(unsigned long long* p) override { P.stats = p; }
  void set_force_slow(int on) override { P.force_slow = on ? 1 : 0; }   // DART_CFG_DEBUG_FORCE_FALLBACK (a test mode)
  bool set_mt_bank(const void* d_view) override { P.ex.mt = (const MtBankView*)d_view; return true; }
  // ---- optional extras: external body force, contact report (see Extras in planar_kernel.hpp)
  int link_body[T::NL] = {};                 // card body of each link (welded bodies have no link of their own)
  Real* d_ext = nullptr; Real* d_rec = nullptr; int* d_cnt = nullptr; Real* d_cf = nullptr;
  void release() override {
    if (d_ext) (void)hipFree(d_ext); if (d_rec) (void)hipFree(d_rec); if (d_cnt) (void)hipFree(d_cnt); if (d_cf) (void)hipFree(d_cf);
    d_ext = nullptr; d_rec = nullptr; d_cnt = nullptr; d_cf = nullptr;
  }
  int set_ext_force(int body, const double* host_force, int64_t n) override {
    if (!host_force) { P.ex.ext_force = nullptr; return DART_OK; }
    int link = -1;
    for (int k = 0; k < T::NL; k++) if (link_body[k] == body) link = k;
    if (link < 0) return DART_E_UNSUPPORTED;   // a root carrier or a welded body: the tree kernel serves those (generic_kernel)
    if (!d_ext && hipMalloc((void**)&d_ext, sizeof(Real) * 3 * (size_t)n) != hipSuccess) return DART_E_HIP;
    std::vector<Real> tmp(3 * (size_t)n);
    for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (Real)host_force[i];
    if (hipMemcpy(d_ext, tmp.data(), sizeof(Real) * tmp.size(), hipMemcpyHostToDevice) != hipSuccess) return DART_E_HIP;
    P.ex.ext_force = d_ext; P.ex.ext_link = link;
    return DART_OK;
  }
  int max_contacts() const override { return T::NC; }
  int set_contact_report(bool on, int64_t n) override {
    if (on && !d_rec) {
      if (hipMalloc((void**)&d_rec, sizeof(Real) * 8 * T::NC * (size_t)n) != hipSuccess) return DART_E_HIP;
      if (hipMalloc((void**)&d_cnt, sizeof(int) * (size_t)n) != hipSuccess) return DART_E_HIP;
      if (hipMalloc((void**)&d_cf, sizeof(Real) * T::NDOF * (size_t)n) != hipSuccess) return DART_E_HIP;
      (void)hipMemset(d_cnt, 0, sizeof(int) * (size_t)n); (void)hipMemset(d_cf, 0, sizeof(Real) * T::NDOF * (size_t)n);
    }
    P.ex.creport = on ? d_rec : nullptr; P.ex.creport_count = on ? d_cnt : nullptr; P.ex.cf_report = on ? d_cf : nullptr;
    return DART_OK;
  }
  int get_contacts(hipStream_t s, int64_t n, int32_t* count, int32_t* bodies, double* point_force, int maxc) override {
    if (!P.ex.creport) return DART_E_INVALID;
    if (hipStreamSynchronize(s) != hipSuccess) return DART_E_HIP;
    std::vector<Real> rec(8 * (size_t)T::NC * (size_t)n);
    std::vector<int> cnt((size_t)n);
    if (hipMemcpy(rec.data(), d_rec, sizeof(Real) * rec.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    if (hipMemcpy(cnt.data(), d_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    for (int64_t e = 0; e < n; e++) {
      const int k = cnt[(size_t)e] < maxc ? cnt[(size_t)e] : maxc;
      count[e] = cnt[(size_t)e];
      for (int c = 0; c < maxc; c++) {
        const bool live = c < k && c < T::NC;
        const Real* r = rec.data() + ((size_t)e * T::NC + (live ? c : 0)) * 8;
        if (bodies) { bodies[((size_t)e * maxc + c) * 2] = live ? (int32_t)r[0] : -1; bodies[((size_t)e * maxc + c) * 2 + 1] = -1; }
        if (point_force) for (int a = 0; a < 6; a++) point_force[((size_t)e * maxc + c) * 6 + a] = live ? (double)r[2 + a] : 0.0;
      }
    }
    return DART_OK;
  }
  int get_constraint_forces(hipStream_t s, int64_t n, double* out) override {
    if (!P.ex.cf_report) return DART_E_INVALID;
    if (hipStreamSynchronize(s) != hipSuccess) return DART_E_HIP;
    std::vector<Real> tmp((size_t)T::NDOF * (size_t)n);
    if (hipMemcpy(tmp.data(), d_cf, sizeof(Real) * tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) return DART_E_HIP;
    for (size_t i = 0; i < tmp.size(); i++) out[i] = (double)tmp[i];
    return DART_OK;
  }
  int slots() const override { return 2 * T::NC + n_limited<T>(); }
  int64_t lds_bytes() const override {
    const int w = constraint_lds_words<T, Real>() + (topo_wave_fallback<T>::value ? coop_words<16>() : 0);
    return w > 1 ? (int64_t)(w * sizeof(Real)) : 0;
  }
};

static inline bool is_identity3(const double* T16, double tol = 1e-12) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (std::fabs(T16[4 * i + j] - (i == j ? 1.0 : 0.0)) > tol) return false;
  return true;
}

// Validate the card against topology T and fill the kernel parameters.  Returns "" or the reason it does not fit.
// Bodies on weld joints (half_cheetah.skel's head) are folded into the link they are welded to: mass, COM, inertia and
// collision shapes; a moving joint on a welded body is declined.
template <class Real, class T>
std::string fill_params(const DartModelCard& c, Params<Real, T>& P) {
  // No fp contraction in here: tools/gen_static_models.py restates this function in Python and prints the doubles it computes (welded
  // bodies folded into their link, capsule end points) as the baked models' constants; a baked kernel is selected only when the block built
  // here is bit-identical to them (Static::matches), and an fma in one of these expressions would round differently from Python's * and +.
#pragma clang fp contract(off)
  constexpr int NL = T::NL;
  // plane of motion: x-y (rotations about +-z, gravity along -y) or, for topologies with PLANE_XZ, the horizontal x-z plane
  // (rotations about +-y -- a turn about +y is clockwise seen in (x, z) coordinates, hence the opposite sigma; gravity is normal
  // to the plane and drops out).  V = index of the second in-plane coordinate, W = the normal one.
  constexpr bool XZ = topo_plane_xz<T>::value;
  constexpr int V = XZ ? 2 : 1, W = XZ ? 1 : 2;
  if (c.ndofs != T::NDOF || c.nbodies < NL + 2) return "body/dof count";
  if (c.act_dim != T::NA || c.act_dof0 != T::NDOF - T::NA) return "action layout";
  if (topo_physics<T>::value != (c.task == DART_TASK_NONE)) return "task";   // PhysTopo variants serve DART_TASK_NONE, the others never
  if (c.obs_dim != obs_dim_of<T>()) return "obs_dim";
  if (!topo_physics<T>::value && XZ != (c.task == DART_TASK_SNAKE)) return "task";   // (a physics-only card's plane shows in its root axes, below)
  if (c.task != DART_TASK_NONE && c.task != DART_TASK_HOPPER && c.task != DART_TASK_WALKER2D && c.task != DART_TASK_HALFCHEETAH &&
      c.task != DART_TASK_SNAKE) return "task";
  for (int d = 0; d < c.ndofs; d++) if (c.joint_friction[d] != 0.0) return "joint Coulomb friction";
  if (c.gravity[0] != 0 || c.gravity[2] != 0) return "gravity must be along y";
  // floating base: prismatic x, prismatic y, revolute +-z
  if (c.jtype[0] != DART_JT_PRISMATIC || c.jtype[1] != DART_JT_PRISMATIC || c.parent[0] != -1 || c.parent[1] != 0)
    return "root carriers";
  if (std::fabs(c.axes[0][0] - 1) > 1e-12 || std::fabs(c.axes[1][V] - 1) > 1e-12) return "root prismatic axes";
  if (c.mass[0] != 0 || c.mass[1] != 0) return "root carriers must be massless";
  double x0 = 0, y0 = 0, y_plane = 0;   // y_plane: height of an x-z plane of motion
  for (int b = 0; b < 3; b++) {
    if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated root frames";
    y_plane += c.T_pj[b][7] - c.T_cj[b][7];
    x0 += c.T_pj[b][3] - c.T_cj[b][3];
    y0 += c.T_pj[b][3 + 4 * V] - c.T_cj[b][3 + 4 * V];
    if (!XZ && (c.T_pj[b][3 + 4 * W] != 0 || c.T_cj[b][3 + 4 * W] != 0)) return "root offset out of the plane";
  }
  P.root_x0 = (Real)x0; P.root_y0 = (Real)y0;
  // bodies -> links (welded bodies share their parent's link, shifted by the weld offset)
  int link_of_body[DART_MAX_BODIES], body_of_link[NL], nl = 0;
  double wx[DART_MAX_BODIES], wy[DART_MAX_BODIES];   // origin of the body frame in its link frame
  for (int b = 0; b < c.nbodies; b++) { link_of_body[b] = -1; wx[b] = 0; wy[b] = 0; }
  for (int b = 2; b < c.nbodies; b++) {
    if (c.jtype[b] == DART_JT_WELD) {
      const int pb = c.parent[b];
      if (b == 2 || pb < 2 || link_of_body[pb] < 0) return "weld without a link to hold it";
      if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated weld";
      if (c.T_pj[b][3 + 4 * W] != 0 || c.T_cj[b][3 + 4 * W] != 0) return "weld off plane";
      link_of_body[b] = link_of_body[pb];
      wx[b] = wx[pb] + c.T_pj[b][3] - c.T_cj[b][3]; wy[b] = wy[pb] + c.T_pj[b][3 + 4 * V] - c.T_cj[b][3 + 4 * V];
      continue;
    }
    if (nl >= NL) return "body/dof count";
    body_of_link[nl] = b; link_of_body[b] = nl++;
  }
  if (nl != NL) return "body/dof count";
  double lm[NL], lcx[NL], lcy[NL], lizz[NL];
  for (int k = 0; k < NL; k++) {
    const int b = body_of_link[k];
    if (c.jtype[b] != DART_JT_REVOLUTE) return "non-revolute link joint";
    if (std::fabs(std::fabs(c.axes[b][W]) - 1) > 1e-12) return "link axis must be normal to the plane";
    if (k > 0) {
      const int pb = c.parent[b];
      if (pb < 2 || link_of_body[pb] != T::parent(k)) return "tree shape";
      if (c.jtype[pb] == DART_JT_WELD) return "joint on a welded body";
      if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated joint frames";
      if (c.T_pj[b][3 + 4 * W] != 0 || c.T_cj[b][3] != 0 || c.T_cj[b][7] != 0 || c.T_cj[b][11] != 0) return "joint offsets";
      P.jx[k] = (Real)c.T_pj[b][3]; P.jy[k] = (Real)c.T_pj[b][3 + 4 * V];
    } else {
      if (c.parent[b] != 1) return "root link parent";
      P.jx[0] = 0; P.jy[0] = 0;
    }
    if (c.com[b][W] != 0) return "com off plane";
    P.sigma[k] = (Real)((c.axes[b][W] > 0) != XZ ? 1.0 : -1.0);
    lm[k] = c.mass[b]; lcx[k] = c.com[b][0]; lcy[k] = c.com[b][V]; lizz[k] = c.inertia[b][4 * W];
    const int d = c.dof_offset[b];
    if (d != 2 + k) return "dof order";
    bool lim = c.limited[d] != 0;
    if (lim && !T::limited(k)) return "limit on unlimited link";
    P.lo[k] = (Real)(lim ? c.lower[d] : -INFINITY);
    P.hi[k] = (Real)(lim ? c.upper[d] : INFINITY);
  }
  for (int b = 2; b < c.nbodies; b++) {   // fold the welded bodies in: composite mass, COM, inertia about the new COM
    if (c.jtype[b] != DART_JT_WELD || c.mass[b] == 0) continue;
    if (c.com[b][W] != 0) return "com off plane";
    const int k = link_of_body[b];
    const double mb = c.mass[b], bx = wx[b] + c.com[b][0], by = wy[b] + c.com[b][V];
    const double m = lm[k] + mb, nx = (lm[k] * lcx[k] + mb * bx) / m, ny = (lm[k] * lcy[k] + mb * by) / m;
    lizz[k] = lizz[k] + lm[k] * ((lcx[k] - nx) * (lcx[k] - nx) + (lcy[k] - ny) * (lcy[k] - ny)) + c.inertia[b][4 * W] +
              mb * ((bx - nx) * (bx - nx) + (by - ny) * (by - ny));
    lm[k] = m; lcx[k] = nx; lcy[k] = ny;
  }
  for (int k = 0; k < NL; k++) { P.mass[k] = (Real)lm[k]; P.cx[k] = (Real)lcx[k]; P.cy[k] = (Real)lcy[k]; P.izz[k] = (Real)lizz[k]; }
  for (int d = 0; d < T::NDOF; d++) {
    if (d < 2 && (c.limited[d] || c.stiffness[d] != 0)) return "limits/springs on root translation";
    if (d < 3 && (c.damping[d] != 0 || c.stiffness[d] != 0)) return "damping / springs on the floating root";   // (ImplicitDofs)
    P.damp[d] = (Real)c.damping[d]; P.q0[d] = (Real)c.init_pos[d]; P.dq0[d] = (Real)c.init_vel[d];
    P.stiff[d] = (Real)c.stiffness[d]; P.rest[d] = (Real)c.rest[d];
    P.sqe[d] = (Real)std::sqrt(c.dt * c.damping[d] + c.dt * c.dt * c.stiffness[d]);
  }
  P.impulse_M = c.impulse_inertia == DART_IMPULSE_MASS ? 1 : 0;
  int nc = 0;
  if constexpr (!topo_contacts<T>::value) {
    // the robot slides in a horizontal plane: its joints cannot bring a shape to the floor, so a shape that clears it now always will
    for (int s = 0; s < c.nshapes; s++) {
      if (!c.shape_collidable[s]) continue;
      const double reach = c.shape_type[s] == DART_SH_CAPSULE ? c.shape_size[s][0] + 0.5 * c.shape_size[s][1]
                                                               : 0.5 * std::sqrt(c.shape_size[s][0] * c.shape_size[s][0] + c.shape_size[s][1] * c.shape_size[s][1] + c.shape_size[s][2] * c.shape_size[s][2]);
      const double clear = c.shape_type[s] == DART_SH_CAPSULE && std::fabs(c.shape_pose[s][4 + 2]) < 1e-9 ? c.shape_size[s][0] : reach;
      if (y_plane + c.shape_pose[s][7] - clear <= c.ground_y) return "a shape reaches the floor";
    }
    for (int k = 0; k < T::NC; k++) { P.e1x[k] = P.e1y[k] = P.e2x[k] = P.e2y[k] = 0; P.rad[k] = (Real)-INFINITY; P.cbody[k] = 0; }
    nc = T::NC;
  } else
  for (int s = 0; s < c.nshapes; s++) {
    if (!c.shape_collidable[s]) continue;
    if (c.shape_type[s] != DART_SH_CAPSULE) return "collidable non-capsule shape";
    if (nc >= T::NC) return "too many collidable shapes";
    const int sb = c.shape_body[s];
    if (sb < 2 || link_of_body[sb] != T::clink(nc)) return "collidable shape on unexpected link";
    const double* S = c.shape_pose[s];
    double hl = 0.5 * c.shape_size[s][1];
    if (std::fabs(S[10]) > 1e-9 || S[11] != 0) return "capsule axis off plane";
    P.e1x[nc] = (Real)(wx[sb] + S[3] + hl * S[2]); P.e1y[nc] = (Real)(wy[sb] + S[7] + hl * S[6]);
    P.e2x[nc] = (Real)(wx[sb] + S[3] - hl * S[2]); P.e2y[nc] = (Real)(wy[sb] + S[7] - hl * S[6]);
    P.rad[nc] = (Real)c.shape_size[s][0];
    P.cbody[nc] = sb;
    nc++;
  }
  if (nc != T::NC) return "collidable shape count";
  (void)y_plane;
  P.dt = (Real)c.dt; P.ground_y = (Real)c.ground_y; P.g = (Real)(XZ ? 0.0 : -c.gravity[1]); P.mu = (Real)c.friction;
  P.erp_dt = (Real)(c.erp / c.dt); P.max_erv = (Real)c.max_erv; P.limit_erp_dt = (Real)(c.limit_erp / c.dt);
  P.cfm1 = (Real)(1.0 + c.cfm); P.ccfm1 = (Real)(1.0 + c.contact_cfm);
  for (int k = 0; k < T::NA; k++) {
    P.act_scale[k] = (Real)c.act_scale[k]; P.act_lo[k] = (Real)c.act_low[k]; P.act_hi[k] = (Real)c.act_high[k];
  }
  P.alive = (Real)c.alive_bonus; P.ctrl_cost = (Real)c.ctrl_cost; P.pen_each = (Real)(c.limit_penalty * 1.5);
  P.pen_margin = (Real)c.penalty_margin; P.h_lo = (Real)c.height_lo; P.h_hi = (Real)c.height_hi;
  P.ang_max = (Real)c.angle_max; P.s_max = (Real)c.state_abs_max; P.v_clip = (Real)c.obs_vel_clip;
  P.inv_envdt = (Real)(1.0 / (c.dt * c.frame_skip)); P.noise = (Real)c.reset_noise; P.noise_v = (Real)c.reset_noise_vel;
  P.frame_skip = c.frame_skip; P.max_steps = c.max_episode_steps; P.task = c.task;
  P.penalty_link = c.penalty_dof >= 2 ? c.penalty_dof - 2 : -1;
  if (c.task != DART_TASK_NONE && c.height_body != 2) return "height body must be the root link";
  P.solver = 0; P.iters1 = DART_BPP_DEFAULT_ITERS; P.iters2 = DART_BPP_DEFAULT_ITERS; P.stats = nullptr; P.force_slow = 0; P.ex = Extras<Real>();
  P.fluid_k = 0; P.dev_cost = 0;
  if (c.task == DART_TASK_SNAKE) {   // aux_real = {alive bonus, control cost, deviation cost, fluid coefficient} (model_card.py SNAKE)
    P.alive = (Real)c.aux_real[0]; P.ctrl_cost = (Real)c.aux_real[1]; P.dev_cost = (Real)c.aux_real[2]; P.fluid_k = (Real)c.aux_real[3];
    P.pen_each = 0; P.penalty_link = -1;
  }
  return "";
}

// generic (runtime-parameter) kernel, or the compile-time specialisation when the card is bit-identical to a baked one
template <class Real, class T, class Static>
std::unique_ptr<Impl> make_for_topology(const DartModelCard& c, std::string& why, bool allow_static) {
  Params<Real, T> R;
  std::string w = fill_params<Real, T>(c, R);
  if (!w.empty()) { why += w; return nullptr; }
  if constexpr (!std::is_void<Static>::value) {
    if (allow_static && Static::matches(R)) {
      auto p = std::make_unique<ImplT<Real, T, Static>>();
      p->P.max_steps = R.max_steps; p->P.solver = R.solver; p->P.iters1 = R.iters1; p->P.iters2 = R.iters2;
      p->P.stats = nullptr; p->P.force_slow = 0; p->P.ex = Extras<Real>();
      for (int b = 2, k = 0; b < c.nbodies && k < T::NL; b++) if (c.jtype[b] != DART_JT_WELD) p->link_body[k++] = b;
      p->is_static = true;
      return p;
    }
  }
  auto p = std::make_unique<ImplT<Real, T>>();
  p->P = R;
  p->P.force_slow = 0;
  for (int b = 2, k = 0; b < c.nbodies && k < T::NL; b++) if (c.jtype[b] != DART_JT_WELD) p->link_body[k++] = b;
  return p;
}

// ------------------------------------------------------------------ cart + pendulum chains (cart_kernel.hpp)
template <class Real, int NP>
struct CartImplT : Impl {
  CartParams<Real, NP> P;
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
    dim3 grid((unsigned)((n + block_threads - 1) / block_threads)), block(block_threads);
    hipLaunchKernelGGL((cart_step_kernel<Real, NP>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew, done, trunc,
                       autoreset, seed, off);
    return hipGetLastError();
  }
  hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                   const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((cart_reset_kernel<Real, NP>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, mask, qn, vn, obs, seed, off,
                       obs_masked_only);
    return hipGetLastError();
  }
  hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((state_io_kernel<Real, 1 + NP>), grid, block, 0, s, n, (Real*)q, (Real*)dq, qh, dqh, to_device);
    return hipGetLastError();
  }
  void set_solver(int, int it1, int) override { P.iters = it1 > 0 ? it1 : DART_BPP_DEFAULT_ITERS; }   // limits only: one exact (pivoting) stage
  void set_stats(unsigned long long*) override {}
  int slots() const override { return 1 + NP; }
};

// cart (prismatic along x, body 0) + NP links on revolute +-z joints in a chain; welded bodies (the weights) fold into their link
template <class Real, int NP>
std::string fill_cart(const DartModelCard& c, CartParams<Real, NP>& P) {
  constexpr int N = 1 + NP;
  if (c.ndofs != N || c.nbodies < N) return "body/dof count";
  if (c.task != DART_TASK_NONE && c.task != DART_TASK_CARTPOLE && c.task != DART_TASK_CARTPOLE_SWINGUP && c.task != DART_TASK_DOUBLE_PENDULUM) return "task";
  if (c.act_dim != (c.task == DART_TASK_NONE ? N : 1) || c.act_dof0 != 0) return "action layout";   // physics-only: a generalized force per dof
  if (c.obs_dim != cart_obs_dim<NP>(c.task)) return "obs_dim";
  if (c.gravity[0] != 0 || c.gravity[2] != 0) return "gravity must be along y";
  for (int d = 0; d < c.ndofs; d++) if (c.joint_friction[d] != 0.0) return "joint Coulomb friction";
  for (int s = 0; s < c.nshapes; s++) if (c.shape_collidable[s]) return "collidable shapes";
  if (c.jtype[0] != DART_JT_PRISMATIC || c.parent[0] != -1 || std::fabs(c.axes[0][0] - 1) > 1e-12) return "slider root";
  if (!is_identity3(c.T_pj[0]) || !is_identity3(c.T_cj[0]) || c.T_cj[0][3] != 0 || c.T_cj[0][7] != 0) return "rotated slider frames";
  int link_of_body[DART_MAX_BODIES], body_of_link[N], nl = 1;
  double wx[DART_MAX_BODIES], wy[DART_MAX_BODIES];
  for (int b = 0; b < c.nbodies; b++) { link_of_body[b] = -1; wx[b] = 0; wy[b] = 0; }
  link_of_body[0] = 0; body_of_link[0] = 0;
  for (int b = 1; b < c.nbodies; b++) {
    const int pb = c.parent[b];
    if (pb < 0 || link_of_body[pb] < 0) return "tree shape";
    if (c.jtype[b] == DART_JT_WELD) {
      if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b]) || c.T_pj[b][11] != 0 || c.T_cj[b][11] != 0) return "rotated weld";
      link_of_body[b] = link_of_body[pb];
      wx[b] = wx[pb] + c.T_pj[b][3] - c.T_cj[b][3]; wy[b] = wy[pb] + c.T_pj[b][7] - c.T_cj[b][7];
      continue;
    }
    if (nl >= N) return "body/dof count";
    if (c.jtype[b] != DART_JT_REVOLUTE || std::fabs(std::fabs(c.axes[b][2]) - 1) > 1e-12) return "link joint must be revolute about +-z";
    if (link_of_body[pb] != nl - 1 || c.jtype[pb] == DART_JT_WELD) return "not a chain";
    if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b]) || c.T_cj[b][3] != 0 || c.T_cj[b][7] != 0 || c.T_pj[b][11] != 0) return "joint frames";
    if (c.dof_offset[b] != nl) return "dof order";
    body_of_link[nl] = b; link_of_body[b] = nl++;
  }
  if (nl != N) return "body/dof count";
  double lm[N], lcx[N], lcy[N], lizz[N];
  for (int k = 0; k < N; k++) {
    const int b = body_of_link[k];
    if (c.com[b][2] != 0) return "com off plane";
    lm[k] = c.mass[b]; lcx[k] = c.com[b][0]; lcy[k] = c.com[b][1]; lizz[k] = c.inertia[b][8];
  }
  P.tipx = 0; P.tipy = 0;
  for (int b = 1; b < c.nbodies; b++) {
    if (c.jtype[b] != DART_JT_WELD) continue;
    const int k = link_of_body[b];
    if (c.task == DART_TASK_DOUBLE_PENDULUM && b == c.aux_body[1]) {
      if (k != N - 1) return "weight must hang on the last link";
      P.tipx = (Real)wx[b]; P.tipy = (Real)wy[b];
    }
    if (c.mass[b] == 0) continue;
    const double mb = c.mass[b], bx = wx[b] + c.com[b][0], by = wy[b] + c.com[b][1];
    const double m = lm[k] + mb, nx = (lm[k] * lcx[k] + mb * bx) / m, ny = (lm[k] * lcy[k] + mb * by) / m;
    lizz[k] = lizz[k] + lm[k] * ((lcx[k] - nx) * (lcx[k] - nx) + (lcy[k] - ny) * (lcy[k] - ny)) + c.inertia[b][8] +
              mb * ((bx - nx) * (bx - nx) + (by - ny) * (by - ny));
    lm[k] = m; lcx[k] = nx; lcy[k] = ny;
  }
  if (c.task == DART_TASK_DOUBLE_PENDULUM && (c.aux_body[0] != 0 || c.aux_body[1] < 0 || c.aux_body[1] >= c.nbodies ||
                                              link_of_body[c.aux_body[1]] != N - 1)) return "double pendulum bodies";
  P.cart_mass = (Real)lm[0];
  for (int k = 1; k < N; k++) {
    const int b = body_of_link[k], i = k - 1;
    P.sigma[i] = (Real)(c.axes[b][2] > 0 ? 1.0 : -1.0);
    P.mass[i] = (Real)lm[k]; P.cx[i] = (Real)lcx[k]; P.cy[i] = (Real)lcy[k]; P.izz[i] = (Real)lizz[k];
    P.jx[i] = (Real)c.T_pj[b][3]; P.jy[i] = (Real)c.T_pj[b][7];
  }
  for (int d = 0; d < N; d++) {
    P.lo[d] = (Real)(c.limited[d] ? c.lower[d] : -INFINITY); P.hi[d] = (Real)(c.limited[d] ? c.upper[d] : INFINITY);
    P.damp[d] = (Real)c.damping[d]; P.stiff[d] = (Real)c.stiffness[d]; P.rest[d] = (Real)c.rest[d];
    P.sqe[d] = (Real)std::sqrt(c.dt * c.damping[d] + c.dt * c.dt * c.stiffness[d]);
    P.q0[d] = (Real)c.init_pos[d]; P.dq0[d] = (Real)c.init_vel[d];
  }
  P.impulse_M = c.impulse_inertia == DART_IMPULSE_MASS ? 1 : 0;
  P.dt = (Real)c.dt; P.g = (Real)(-c.gravity[1]); P.limit_erp_dt = (Real)(c.limit_erp / c.dt); P.max_erv = (Real)c.max_erv;
  P.cfm1 = (Real)(1.0 + c.cfm);
  P.act_scale = (Real)c.act_scale[0]; P.act_lo = (Real)c.act_low[0]; P.act_hi = (Real)c.act_high[0];
  for (int k = 0; k < 8; k++) P.aux[k] = (Real)c.aux_real[k];
  if (c.task == DART_TASK_CARTPOLE) { P.aux[0] = (Real)c.alive_bonus; P.aux[1] = (Real)c.ctrl_cost; }
  P.angle_max = (Real)c.angle_max; P.s_max = (Real)c.state_abs_max; P.noise = (Real)c.reset_noise; P.noise_v = (Real)c.reset_noise_vel;
  P.frame_skip = c.frame_skip; P.max_steps = c.max_episode_steps; P.task = c.task; P.iters = DART_BPP_DEFAULT_ITERS;
  return "";
}

template <class Real, int NP>
std::unique_ptr<Impl> make_cart(const DartModelCard& c, std::string& why) {
  auto p = std::make_unique<CartImplT<Real, NP>>();
  std::string w = fill_cart<Real, NP>(c, p->P);
  if (!w.empty()) { why += w; return nullptr; }
  return p;
}

// ------------------------------------------------------------------ fixed-base arm in the horizontal plane (arm_kernel.hpp)
template <class Real, int NP>
struct ArmImplT : Impl {
  ArmParams<Real, NP> P;
  Real* d_tstate = nullptr;
  hipError_t prepare(int64_t n) override {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_tstate, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    if ((e = hipMemset(d_tstate, 0, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    P.tstate = d_tstate;
    return hipSuccess;
  }
  void release() override { if (d_tstate) (void)hipFree(d_tstate); d_tstate = nullptr; P.tstate = nullptr; }
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
    dim3 grid((unsigned)((n + block_threads - 1) / block_threads)), block(block_threads);
    hipLaunchKernelGGL((arm_step_kernel<Real, NP>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew, done, trunc,
                       autoreset, seed, off);
    return hipGetLastError();
  }
  hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                   const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((arm_reset_kernel<Real, NP>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, mask, qn, vn, obs, seed, off,
                       obs_masked_only);
    return hipGetLastError();
  }
  hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((state_io_kernel<Real, NP>), grid, block, 0, s, n, (Real*)q, (Real*)dq, qh, dqh, to_device);
    return hipGetLastError();
  }
  int set_task_state(hipStream_t s, const uint8_t* d_mask, const double* d_values, int64_t n) override {
    hipLaunchKernelGGL((arm_task_state_kernel<Real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_mask, d_values, d_tstate);
    return hipGetLastError() == hipSuccess ? DART_OK : DART_E_HIP;
  }
  void persistent(std::vector<std::pair<void*, size_t>>& v, int64_t n) override {
    if (d_tstate) v.push_back({d_tstate, sizeof(Real) * 4 * (size_t)n});      // the reach targets
  }
  void set_solver(int, int it1, int) override { P.iters = it1 > 0 ? it1 : DART_BPP_DEFAULT_ITERS; }   // no contacts: one exact (pivoting) stage
  void set_stats(unsigned long long*) override {}
  int slots() const override { return 2 * NP; }
};

// NP links on revolute +-y joints in a chain from the world, moving in the x-z plane; welded bodies (the finger tip) fold into
// their link.  Task: the 2-D reacher (reacher2d.py).
template <class Real, int NP>
std::string fill_arm(const DartModelCard& c, ArmParams<Real, NP>& P) {
  if (c.task != DART_TASK_REACHER2D && c.task != DART_TASK_NONE) return "task";
  if (c.ndofs != NP || c.nbodies < NP) return "body/dof count";
  if (c.act_dim != NP || c.act_dof0 != 0) return "action layout";
  if (c.obs_dim != arm_obs_dim_rt<NP>(c.task)) return "obs_dim";
  if (c.gravity[0] != 0 || c.gravity[2] != 0) return "gravity must be normal to the plane of motion";
  for (int s = 0; s < c.nshapes; s++) if (c.shape_collidable[s]) return "collidable shapes";
  int link_of_body[DART_MAX_BODIES], body_of_link[NP], nl = 0;
  double wx[DART_MAX_BODIES], wy[DART_MAX_BODIES];   // origin of a (welded) body's frame in its link frame, plane coordinates (x, z)
  for (int b = 0; b < c.nbodies; b++) { link_of_body[b] = -1; wx[b] = 0; wy[b] = 0; }
  for (int b = 0; b < c.nbodies; b++) {
    const int pb = c.parent[b];
    if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated joint frames";
    if (c.jtype[b] == DART_JT_WELD) {
      if (pb < 0 || link_of_body[pb] < 0) return "weld without a link to hold it";
      if (c.T_pj[b][7] != 0 || c.T_cj[b][7] != 0) return "weld off plane";
      link_of_body[b] = link_of_body[pb];
      wx[b] = wx[pb] + c.T_pj[b][3] - c.T_cj[b][3]; wy[b] = wy[pb] + c.T_pj[b][11] - c.T_cj[b][11];
      continue;
    }
    if (nl >= NP) return "body/dof count";
    if (c.jtype[b] != DART_JT_REVOLUTE || std::fabs(std::fabs(c.axes[b][1]) - 1) > 1e-12) return "link joint must be revolute about +-y";
    if (nl == 0 ? pb != -1 : (pb < 0 || link_of_body[pb] != nl - 1 || c.jtype[pb] == DART_JT_WELD)) return "not a chain from the world";
    if (c.T_cj[b][3] != 0 || c.T_cj[b][7] != 0 || c.T_cj[b][11] != 0 || (nl > 0 && c.T_pj[b][7] != 0)) return "joint offsets";
    if (c.dof_offset[b] != nl) return "dof order";
    body_of_link[nl] = b; link_of_body[b] = nl++;
  }
  if (nl != NP) return "body/dof count";
  double lm[NP], lcx[NP], lcy[NP], lizz[NP];
  for (int k = 0; k < NP; k++) {
    const int b = body_of_link[k];
    if (c.com[b][1] != 0) return "com off plane";
    lm[k] = c.mass[b]; lcx[k] = c.com[b][0]; lcy[k] = c.com[b][2]; lizz[k] = c.inertia[b][4];
  }
  for (int b = 0; b < c.nbodies; b++) {   // fold the welded bodies in: composite mass, COM, inertia about the new COM
    if (c.jtype[b] != DART_JT_WELD || c.mass[b] == 0) continue;
    if (c.com[b][1] != 0) return "com off plane";
    const int k = link_of_body[b];
    const double mb = c.mass[b], bx = wx[b] + c.com[b][0], by = wy[b] + c.com[b][2];
    const double m = lm[k] + mb, nx = (lm[k] * lcx[k] + mb * bx) / m, ny = (lm[k] * lcy[k] + mb * by) / m;
    lizz[k] = lizz[k] + lm[k] * ((lcx[k] - nx) * (lcx[k] - nx) + (lcy[k] - ny) * (lcy[k] - ny)) + c.inertia[b][4] +
              mb * ((bx - nx) * (bx - nx) + (by - ny) * (by - ny));
    lm[k] = m; lcx[k] = nx; lcy[k] = ny;
  }
  const int tb = c.task == DART_TASK_NONE ? body_of_link[NP - 1] : c.aux_body[0];   // reacher2d.py:31: bodynodes[-1].com() (a physics-only card names no tip)
  if (tb < 0 || tb >= c.nbodies || link_of_body[tb] != NP - 1) return "tip body must ride on the last link";
  if (c.aux_real[1] != 0) return "tip offset off plane";
  P.tipx = (Real)(wx[tb] + c.aux_real[0]); P.tipy = (Real)(wy[tb] + c.aux_real[2]);
  P.height = (Real)(c.T_pj[body_of_link[0]][7]);
  for (int k = 0; k < NP; k++) {
    const int b = body_of_link[k], d = c.dof_offset[b];
    P.sigma[k] = (Real)(c.axes[b][1] > 0 ? -1.0 : 1.0);   // a turn about +y is clockwise in (x, z)
    P.mass[k] = (Real)lm[k]; P.cx[k] = (Real)lcx[k]; P.cy[k] = (Real)lcy[k]; P.izz[k] = (Real)lizz[k];
    P.jx[k] = (Real)c.T_pj[b][3]; P.jy[k] = (Real)c.T_pj[b][11];
    P.lo[k] = (Real)(c.limited[d] ? c.lower[d] : -INFINITY); P.hi[k] = (Real)(c.limited[d] ? c.upper[d] : INFINITY);
    P.damp[k] = (Real)c.damping[d]; P.stiff[k] = (Real)c.stiffness[d]; P.rest[k] = (Real)c.rest[d];
    P.sqe[k] = (Real)std::sqrt(c.dt * c.damping[d] + c.dt * c.dt * c.stiffness[d]);
    P.q0[k] = (Real)c.init_pos[d]; P.dq0[k] = (Real)c.init_vel[d];
    P.fric_dt[k] = (Real)(c.joint_friction[d] * c.dt);
    P.act_scale[k] = (Real)c.act_scale[k]; P.act_lo[k] = (Real)c.act_low[k]; P.act_hi[k] = (Real)c.act_high[k];
  }
  P.dt = (Real)c.dt; P.limit_erp_dt = (Real)(c.limit_erp / c.dt); P.max_erv = (Real)c.max_erv; P.cfm1 = (Real)(1.0 + c.cfm);
  P.noise = (Real)c.reset_noise; P.noise_v = (Real)c.reset_noise_vel;
  P.frame_skip = c.frame_skip; P.max_steps = c.max_episode_steps; P.task = c.task; P.iters = DART_BPP_DEFAULT_ITERS; P.tstate = nullptr;
  P.impulse_M = c.impulse_inertia == DART_IMPULSE_MASS ? 1 : 0;
  return "";
}

template <class Real, int NP>
std::unique_ptr<Impl> make_arm(const DartModelCard& c, std::string& why) {
  auto p = std::make_unique<ArmImplT<Real, NP>>();
  std::string w = fill_arm<Real, NP>(c, p->P);
  if (!w.empty()) { why += w; return nullptr; }
  return p;
}

// ------------------------------------------------------------------ fixed-base 3-D chain of revolute links (chain3d_kernel.hpp)
template <class Real, int NL, bool FRIC>
struct Chain3dImplT : Impl {
  Chain3Params<Real, NL> P;
  Real* d_tstate = nullptr;
  hipError_t prepare(int64_t n) override {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_tstate, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    if ((e = hipMemset(d_tstate, 0, sizeof(Real) * 4 * (size_t)n)) != hipSuccess) return e;
    P.tstate = d_tstate;
    return hipSuccess;
  }
  void release() override { if (d_tstate) (void)hipFree(d_tstate); d_tstate = nullptr; P.tstate = nullptr; }
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
    dim3 grid((unsigned)((n + block_threads - 1) / block_threads)), block(block_threads);
    hipLaunchKernelGGL((chain3d_step_kernel<Real, NL, FRIC>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew, done, trunc,
                       autoreset, seed, off);
    return hipGetLastError();
  }
  hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                   const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((chain3d_reset_kernel<Real, NL>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, mask, qn, vn, obs, seed, off,
                       obs_masked_only);
    return hipGetLastError();
  }
  hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) override {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL((state_io_kernel<Real, NL>), grid, block, 0, s, n, (Real*)q, (Real*)dq, qh, dqh, to_device);
    return hipGetLastError();
  }
  int set_task_state(hipStream_t s, const uint8_t* d_mask, const double* d_values, int64_t n) override {
    hipLaunchKernelGGL((arm_task_state_kernel<Real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_mask, d_values, d_tstate);
    return hipGetLastError() == hipSuccess ? DART_OK : DART_E_HIP;
  }
  void persistent(std::vector<std::pair<void*, size_t>>& v, int64_t n) override {
    if (d_tstate) v.push_back({d_tstate, sizeof(Real) * 4 * (size_t)n});      // the reach targets
  }
  void set_solver(int, int it1, int) override { P.iters = it1 > 0 ? it1 : DART_BPP_DEFAULT_ITERS; }   // no contacts: one exact (pivoting) stage
  void set_stats(unsigned long long*) override {}
  int slots() const override { return FRIC ? 2 * NL : NL; }
};

// The card through the tree kernel's model builder (multi-dof joints expanded into 1-dof links), then the checks that make it this
// kernel's case: NL revolute links in one chain from the world, one dof each in chain order, nothing that collides.  Task: the
// 3-D reacher (reacher.py).
template <class Real, int NL, bool FRIC>
std::unique_ptr<Impl> make_chain3d(const DartModelCard& c, std::string& why) {
  if (c.task != DART_TASK_REACHER3D && c.task != DART_TASK_NONE) { why += "task"; return nullptr; }
  if (c.ndofs != NL || c.act_dim != NL || c.act_dof0 != 0 || c.obs_dim != chain3d_obs_dim_rt<NL>(c.task)) { why += "dof / action / observation layout"; return nullptr; }
  auto Mp = std::make_unique<SpatialModel<Real>>();
  SpatialModel<Real>& M = *Mp;
  const std::string w = fill_spatial<Real>(c, M);
  if (!w.empty()) { why += w; return nullptr; }
  if (M.nl != NL || M.n != NL) { why += "expanded link count"; return nullptr; }
  if (M.nshapes != 0 || M.npairs != 0 || M.free_root) { why += "contacts or a free root"; return nullptr; }
  if ((M.has_joint_friction != 0) != FRIC) { why += "joint friction"; return nullptr; }
  for (int i = 0; i < NL; i++)
    if (M.parent[i] != i - 1 || M.jtype[i] != 2 || M.dof[i] != i) { why += "not a chain of revolute links in dof order"; return nullptr; }
  if (c.task != DART_TASK_NONE && M.aux_link[0] != NL - 1) { why += "tip body must be the last link"; return nullptr; }
  auto p = std::make_unique<Chain3dImplT<Real, NL, FRIC>>();
  Chain3Params<Real, NL>& P = p->P;
  for (int i = 0; i < NL; i++) {
    const Real* g = M.lconst[i];
    for (int k = 0; k < 9; k++) { P.Rpre[i][k] = g[LC_RPRE + k]; P.Rpost[i][k] = g[LC_RPOST + k]; P.inertia[i][k] = g[LC_INERTIA + k]; }
    for (int k = 0; k < 3; k++) {
      P.ppre[i][k] = g[LC_PPRE + k]; P.ppost[i][k] = g[LC_PPOST + k]; P.axis[i][k] = g[LC_AXIS + k]; P.axr[i][k] = g[LC_AXR + k];
      P.cpost[i][k] = g[LC_CPOST + k]; P.com[i][k] = g[LC_COM + k];
    }
    P.mass[i] = M.mass[i];
    P.lo[i] = M.limited[i] ? M.lower[i] : (Real)-INFINITY; P.hi[i] = M.limited[i] ? M.upper[i] : (Real)INFINITY;
    P.damp[i] = M.damp[i]; P.stiff[i] = M.stiff[i]; P.rest[i] = M.rest[i]; P.q0[i] = M.q0[i]; P.dq0[i] = M.dq0[i];
    P.sqe[i] = (Real)std::sqrt((double)M.dt * (double)M.damp[i] + (double)M.dt * (double)M.dt * (double)M.stiff[i]);
    P.fric_dt[i] = M.jfric_dt[i];
    P.act_scale[i] = M.act_scale[i]; P.act_lo[i] = M.act_lo[i]; P.act_hi[i] = M.act_hi[i];
  }
  for (int k = 0; k < 3; k++) { P.g[k] = M.g[k]; P.tip[k] = M.aux_real[k]; }
  P.dt = M.dt; P.limit_erp_dt = M.limit_erp_dt; P.max_erv = M.max_erv; P.cfm1 = M.cfm1;
  P.ctrl_w = M.aux_real[3]; P.done_dist = M.aux_real[4];
  P.noise = M.noise; P.noise_v = M.noise_v;
  P.frame_skip = M.frame_skip; P.max_steps = M.max_steps; P.task = M.task; P.iters = DART_BPP_DEFAULT_ITERS; P.tstate = nullptr;
  P.impulse_M = M.impulse_M;
  return p;
}

template <class Real>
std::unique_ptr<Impl> make_planar(const DartModelCard& c, std::string& why, bool allow_static) {
  why += "hopper-chain, feet only: ";
  if (auto p = make_for_topology<Real, HopperTopo, HopperStatic<Real>>(c, why, allow_static)) return p;
  why += "; hopper-chain, all capsules: ";
  if (auto p = make_for_topology<Real, HopperAllTopo, HopperAllStatic<Real>>(c, why, allow_static)) return p;
  why += "; walker2d-tree, feet only: ";
  if (auto p = make_for_topology<Real, Walker2dTopo, Walker2dStatic<Real>>(c, why, allow_static)) return p;
  why += "; walker2d-tree, all capsules: ";
  if (auto p = make_for_topology<Real, Walker2dAllTopo, Walker2dAllStatic<Real>>(c, why, allow_static)) return p;
  why += "; hopper-chain, physics only: ";
  if (auto p = make_for_topology<Real, PhysTopo<HopperAllTopo>, void>(c, why, allow_static)) return p;
  why += "; walker2d-tree, physics only: ";
  if (auto p = make_for_topology<Real, PhysTopo<Walker2dAllTopo>, void>(c, why, allow_static)) return p;
  why += "; half-cheetah: ";
  if (auto p = make_for_topology<Real, CheetahTopo, CheetahStatic<Real>>(c, why, allow_static)) return p;
  why += "; snake chain in the x-z plane: ";
  if (auto p = make_for_topology<Real, SnakeTopo, void>(c, why, allow_static)) return p;
  // physics-only cards (a user's .skel through envs.DartEnv) of the two remaining planar shapes (round 5)
  why += "; half-cheetah tree, physics only: ";
  if (auto p = make_for_topology<Real, PhysTopo<CheetahTopo>, void>(c, why, allow_static)) return p;
  why += "; snake chain, physics only: ";
  if (auto p = make_for_topology<Real, PhysTopo<SnakeTopo>, void>(c, why, allow_static)) return p;
  why += "; cart + 1 link: ";
  if (auto p = make_cart<Real, 1>(c, why)) return p;
  why += "; cart + 2 links: ";
  if (auto p = make_cart<Real, 2>(c, why)) return p;
  why += "; two-link arm in the x-z plane: ";
  if (auto p = make_arm<Real, 2>(c, why)) return p;
  why += "; five-link revolute chain in 3-D: ";
  if (auto p = make_chain3d<Real, 5, false>(c, why)) return p;
  return nullptr;
}

}  // namespace dartk
