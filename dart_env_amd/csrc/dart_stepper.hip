// dart_stepper.hip -- host side of the C ABI declared in include/dart_stepper.h.
//
// Owns the SoA world state in HBM, validates a DartModelCard against the compiled planar topologies,
// packs the runtime parameters into a kernel argument, and launches the fused step / reset kernels of
// planar_kernel.hpp on one HIP stream per handle.  No CPU compute path exists here by design.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "../../include/dart_stepper.h"
#include "planar_kernel.hpp"
#include "static_models.hpp"
#include "spatial_kernel.hpp"
#include <type_traits>
#include <vector>
#include "mt19937_kernels.hpp"
#include "episode_kernels.hpp"

using namespace dartk;

namespace {

thread_local std::string g_err;

struct Impl {
  virtual ~Impl() {}
  virtual hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act,
                          float* obs, float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed,
                          uint64_t off) = 0;
  virtual hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                           const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off,
                           int obs_masked_only = 0) = 0;
  virtual hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) = 0;
  virtual void set_solver(int solver, int it1, int it2) = 0;
  virtual void set_stats(unsigned long long* p) = 0;
  virtual hipError_t prepare(int64_t) { return hipSuccess; }   // per-handle device allocations of the implementation
  virtual void release() {}
  virtual hipError_t debug_dump(double*) { return hipErrorInvalidValue; }
  virtual int set_ext_force(int /*body*/, const double* /*host_force*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int set_task_state(hipStream_t, const uint8_t* /*d_mask*/, const double* /*d_values*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int slots() const = 0;
  virtual int max_contacts() const { return 0; }
  // device buffers of the implementation that persist between steps (dart_snapshot / dart_restore)
  virtual void persistent(std::vector<std::pair<void*, size_t>>&, int64_t /*n*/) {}
  virtual int set_contact_report(bool /*on*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int get_contacts(hipStream_t, int64_t /*n*/, int32_t* /*count*/, int32_t* /*bodies*/, double* /*point_force*/, int /*max*/) { return DART_E_UNSUPPORTED; }
  virtual int get_constraint_forces(hipStream_t, int64_t /*n*/, double* /*out*/) { return DART_E_UNSUPPORTED; }
  bool soa = true;         // state layout: q[n][N] (planar kernels) or q[N][n] (spatial kernel)
  int block_threads = 64;  // active lanes per wave64 workgroup (32 -> twice the waves)
  bool is_static = false;  // true: model constants are compile-time immediates (static_models.hpp)
};

template <class Real, class T, class PT = Params<Real, T>>
struct ImplT : Impl {
  PT P;
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
    dim3 grid((unsigned)((n + block_threads - 1) / block_threads)), block(block_threads);
    hipLaunchKernelGGL((step_kernel<Real, T, PT>), grid, block, 0, s, P, n, (Real*)q, (Real*)dq, el, ep, act, obs, rew,
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
    P.solver = solver; P.iters1 = it1 > 0 ? it1 : 24; P.iters2 = it2 > 0 ? it2 : 24;
  }
  void set_stats(unsigned long long* p) override { P.stats = p; }
  int slots() const override { return 2 * T::NC + n_limited<T>(); }
};

bool is_identity3(const double* T16, double tol = 1e-12) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (std::fabs(T16[4 * i + j] - (i == j ? 1.0 : 0.0)) > tol) return false;
  return true;
}

// Validate the card against topology T and fill the kernel parameters.  Returns "" or the reason it does not fit.
template <class Real, class T>
std::string fill_params(const DartModelCard& c, Params<Real, T>& P) {
  constexpr int NL = T::NL;
  if (c.nbodies != NL + 2 || c.ndofs != T::NDOF) return "body/dof count";
  if (c.act_dim != T::NA || c.act_dof0 != T::NDOF - T::NA) return "action layout";
  if (c.obs_dim != 2 * T::NDOF - 1 && c.task != DART_TASK_NONE) return "obs_dim";
  for (int d = 0; d < c.ndofs; d++) if (c.joint_friction[d] != 0.0) return "joint Coulomb friction";
  if (c.gravity[0] != 0 || c.gravity[2] != 0) return "gravity must be along y";
  // floating base: prismatic x, prismatic y, revolute +-z
  if (c.jtype[0] != DART_JT_PRISMATIC || c.jtype[1] != DART_JT_PRISMATIC || c.parent[0] != -1 || c.parent[1] != 0)
    return "root carriers";
  if (std::fabs(c.axes[0][0] - 1) > 1e-12 || std::fabs(c.axes[1][1] - 1) > 1e-12) return "root prismatic axes";
  if (c.mass[0] != 0 || c.mass[1] != 0) return "root carriers must be massless";
  double x0 = 0, y0 = 0;
  for (int b = 0; b < 3; b++) {
    if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated root frames";
    x0 += c.T_pj[b][3] - c.T_cj[b][3];
    y0 += c.T_pj[b][7] - c.T_cj[b][7];
    if (c.T_pj[b][11] != 0 || c.T_cj[b][11] != 0) return "root z offset";
  }
  P.root_x0 = (Real)x0; P.root_y0 = (Real)y0;
  for (int k = 0; k < NL; k++) {
    int b = k + 2;
    if (c.jtype[b] != DART_JT_REVOLUTE) return "non-revolute link joint";
    if (std::fabs(std::fabs(c.axes[b][2]) - 1) > 1e-12) return "link axis must be +-z";
    if (k > 0) {
      if (c.parent[b] - 2 != T::parent(k)) return "tree shape";
      if (!is_identity3(c.T_pj[b]) || !is_identity3(c.T_cj[b])) return "rotated joint frames";
      if (c.T_pj[b][11] != 0 || c.T_cj[b][3] != 0 || c.T_cj[b][7] != 0 || c.T_cj[b][11] != 0) return "joint offsets";
      P.jx[k] = (Real)c.T_pj[b][3]; P.jy[k] = (Real)c.T_pj[b][7];
    } else {
      if (c.parent[b] != 1) return "root link parent";
      P.jx[0] = 0; P.jy[0] = 0;
    }
    if (c.com[b][2] != 0) return "com off plane";
    P.sigma[k] = (Real)(c.axes[b][2] > 0 ? 1.0 : -1.0);
    P.mass[k] = (Real)c.mass[b]; P.cx[k] = (Real)c.com[b][0]; P.cy[k] = (Real)c.com[b][1];
    P.izz[k] = (Real)c.inertia[b][8];
    int d = 2 + k;
    if (c.stiffness[d] != 0) return "joint springs";
    bool lim = c.limited[d] != 0;
    if (lim && !T::limited(k)) return "limit on unlimited link";
    P.lo[k] = (Real)(lim ? c.lower[d] : -INFINITY);
    P.hi[k] = (Real)(lim ? c.upper[d] : INFINITY);
  }
  for (int d = 0; d < T::NDOF; d++) {
    if (d < 2 && (c.limited[d] || c.stiffness[d] != 0)) return "limits/springs on root translation";
    P.damp[d] = (Real)c.damping[d]; P.q0[d] = (Real)c.init_pos[d]; P.dq0[d] = (Real)c.init_vel[d];
  }
  int nc = 0;
  for (int s = 0; s < c.nshapes; s++) {
    if (!c.shape_collidable[s]) continue;
    if (c.shape_type[s] != DART_SH_CAPSULE) return "collidable non-capsule shape";
    if (nc >= T::NC) return "too many collidable shapes";
    if (c.shape_body[s] - 2 != T::clink(nc)) return "collidable shape on unexpected link";
    const double* S = c.shape_pose[s];
    double hl = 0.5 * c.shape_size[s][1];
    if (std::fabs(S[10]) > 1e-9 || S[11] != 0) return "capsule axis off plane";
    P.e1x[nc] = (Real)(S[3] + hl * S[2]); P.e1y[nc] = (Real)(S[7] + hl * S[6]);
    P.e2x[nc] = (Real)(S[3] - hl * S[2]); P.e2y[nc] = (Real)(S[7] - hl * S[6]);
    P.rad[nc] = (Real)c.shape_size[s][0];
    nc++;
  }
  if (nc != T::NC) return "collidable shape count";
  P.dt = (Real)c.dt; P.ground_y = (Real)c.ground_y; P.g = (Real)(-c.gravity[1]); P.mu = (Real)c.friction;
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
  P.solver = 0; P.iters1 = 24; P.iters2 = 24; P.stats = nullptr;
  return "";
}

// ------------------------------------------------------------------ general 3-D skeletons (spatial_kernel.hpp)
void mat4_to_Rp(const double* T, double* R, double* p) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j]; p[i] = T[4 * i + 3]; }
}
void inv_Rp(const double* R, const double* p, double* Ri, double* pi) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ri[3 * i + j] = R[3 * j + i];
  for (int i = 0; i < 3; i++) pi[i] = -(Ri[3 * i] * p[0] + Ri[3 * i + 1] * p[1] + Ri[3 * i + 2] * p[2]);
}

// Expand every multi-dof joint of the card into a chain of 1-dof links (massless carriers in between).
template <class Real>
std::string fill_spatial(const DartModelCard& c, SpatialModel<Real>& M, bool physics_only = false, int* body_link_out = nullptr) {
  memset(&M, 0, sizeof(M));
  for (int i = 0; i < SP_MAXL; i++) M.link_body[i] = -1;
  if (c.ndofs > SP_MAXN) return "too many dofs";
  int body_link[DART_MAX_BODIES];
  int nl = 0;
  static const double EX[3] = {1, 0, 0}, EY[3] = {0, 1, 0}, EZ[3] = {0, 0, 1}, I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
  auto add = [&](int parent, int jtype, int dof, const double* axis, const double* Rpre, const double* ppre,
                 const double* Rpost, const double* ppost) -> int {
    if (nl >= SP_MAXL) return -1;
    int i = nl++;
    M.parent[i] = parent; M.jtype[i] = jtype; M.dof[i] = dof;
    for (int k = 0; k < 3; k++) { M.axis[i][k] = (Real)(axis ? axis[k] : 0.0); M.ppre[i][k] = (Real)ppre[k]; M.ppost[i][k] = (Real)ppost[k]; }
    for (int k = 0; k < 9; k++) { M.Rpre[i][k] = (Real)Rpre[k]; M.Rpost[i][k] = (Real)Rpost[k]; }
    if (dof >= 0) M.dof_link[dof] = i;
    auto ident = [](const double* R, const double* p) {
      for (int k = 0; k < 9; k++) if (R[k] != ((k % 4 == 0) ? 1.0 : 0.0)) return 0;
      return (p[0] == 0 && p[1] == 0 && p[2] == 0) ? 1 : 0;
    };
    M.pre_ident[i] = ident(Rpre, ppre); M.post_ident[i] = ident(Rpost, ppost);
    bool anc_root = parent < 0 || M.root_trans[parent];
    M.root_trans[i] = (jtype == 1 && anc_root) ? 1 : 0;
    return i;
  };
  for (int b = 0; b < c.nbodies; b++) {
    int pl = c.parent[b] < 0 ? -1 : body_link[c.parent[b]];
    int d0 = c.dof_offset[b];
    double Rpj[9], ppj[3], Rcj[9], pcj[3], Rpo[9], ppo[3];
    mat4_to_Rp(c.T_pj[b], Rpj, ppj);
    mat4_to_Rp(c.T_cj[b], Rcj, pcj);
    inv_Rp(Rcj, pcj, Rpo, ppo);   // child link frame expressed in the joint frame
    const double* ax = c.axes[b];
    int last = -1;
    switch (c.jtype[b]) {
      case DART_JT_WELD: last = add(pl, 0, -1, nullptr, Rpj, ppj, Rpo, ppo); break;
      case DART_JT_PRISMATIC: last = add(pl, 1, d0, ax, Rpj, ppj, Rpo, ppo); break;
      case DART_JT_REVOLUTE: last = add(pl, 2, d0, ax, Rpj, ppj, Rpo, ppo); break;
      case DART_JT_TRANSLATIONAL: {
        int a = add(pl, 1, d0, EX, Rpj, ppj, I3, Z3); int bb = add(a, 1, d0 + 1, EY, I3, Z3, I3, Z3);
        last = add(bb, 1, d0 + 2, EZ, I3, Z3, Rpo, ppo);
      } break;
      case DART_JT_EULER_XYZ: {
        int a = add(pl, 2, d0, EX, Rpj, ppj, I3, Z3); int bb = add(a, 2, d0 + 1, EY, I3, Z3, I3, Z3);
        last = add(bb, 2, d0 + 2, EZ, I3, Z3, Rpo, ppo);
      } break;
      case DART_JT_EULER_ZYX: {
        int a = add(pl, 2, d0, EZ, Rpj, ppj, I3, Z3); int bb = add(a, 2, d0 + 1, EY, I3, Z3, I3, Z3);
        last = add(bb, 2, d0 + 2, EX, I3, Z3, Rpo, ppo);
      } break;
      case DART_JT_UNIVERSAL: {
        int a = add(pl, 2, d0, ax, Rpj, ppj, I3, Z3);
        last = add(a, 2, d0 + 1, ax + 3, I3, Z3, Rpo, ppo);
      } break;
      case DART_JT_FREE: {   // translation x y z (dofs d0+3..5), then rotations x y z (dofs d0..d0+2) re-centred on the pose: see spatial_kernel.hpp
        if (b != 0 || c.parent[b] >= 0 || d0 != 0) return "only the root body may hang on a free joint";
        int t1 = add(pl, 1, d0 + 3, EX, Rpj, ppj, I3, Z3); int t2 = add(t1, 1, d0 + 4, EY, I3, Z3, I3, Z3);
        int t3 = add(t2, 1, d0 + 5, EZ, I3, Z3, I3, Z3);
        int r1 = add(t3, 2, d0, EX, I3, Z3, I3, Z3); int r2 = add(r1, 2, d0 + 1, EY, I3, Z3, I3, Z3);
        last = add(r2, 2, d0 + 2, EZ, I3, Z3, Rpo, ppo);
        M.free_root = 1; M.free_link = last;
      } break;
      default: return "unsupported joint type";
    }
    if (last < 0) return "too many links";
    body_link[b] = last;
    M.link_is_body[last] = 1;
    M.link_body[last] = b;
    M.mass[last] = (Real)c.mass[b];
    for (int k = 0; k < 3; k++) M.com[last][k] = (Real)c.com[b][k];
    for (int k = 0; k < 9; k++) M.inertia[last][k] = (Real)(c.mass[b] > 0 ? c.inertia[b][k] : 0.0);
  }
  M.nl = nl; M.n = c.ndofs;
  M.has_joint_friction = 0;
  for (int d = 0; d < c.ndofs; d++) {
    if (c.joint_friction[d] < 0) return "negative joint friction";
    M.jfric_dt[d] = (Real)(c.joint_friction[d] * c.dt);
    if (c.joint_friction[d] != 0.0) M.has_joint_friction = 1;
  }
  if (body_link_out) for (int b = 0; b < c.nbodies; b++) body_link_out[b] = body_link[b];
  {  // depth levels, children lists, constant world axes of the root-chain prismatic links
    int depth[SP_MAXL], maxd = 0;
    double Rw[SP_MAXL][9];   // world rotation of each link's JOINT frame at q = 0 (valid for root-chain links)
    for (int i = 0; i < nl; i++) {
      depth[i] = M.parent[i] < 0 ? 0 : depth[M.parent[i]] + 1;
      if (depth[i] > maxd) maxd = depth[i];
    }
    M.nrounds = 0;
    while ((1 << M.nrounds) < maxd + 1) M.nrounds++;
    if (M.nrounds > SP_ROUNDS) return "tree deeper than 64 links";
    for (int i = 0; i < nl; i++) {
      M.anc[i][0] = M.parent[i];
      for (int r = 1; r < SP_ROUNDS; r++) M.anc[i][r] = M.anc[i][r - 1] < 0 ? -1 : M.anc[M.anc[i][r - 1]][r - 1];
    }
    int k = 0;
    for (int i = 0; i < nl; i++) { M.child_start[i] = k; for (int j = 0; j < nl; j++) if (M.parent[j] == i) M.child_list[k++] = j; }
    M.child_start[nl] = k;
    for (int i = 0; i < nl; i++) if (M.child_start[i + 1] - M.child_start[i] > 8) return "more than 8 child links on one link";
    // groups: parent p and its only child i share their joint origin for every q when p is a massless carrier whose
    // own motion does not move the child's joint frame origin (revolute, weld, or a root translation folded into roff)
    for (int i = 0; i < nl; i++) M.group_leader[i] = i;
    for (int i = nl - 1; i > 0; i--) {
      const int p = M.parent[i];
      if (p < 0) continue;
      const bool still = M.jtype[p] == 2 || M.jtype[p] == 0 || (M.jtype[p] == 1 && M.root_trans[p]);
      // (the snake's fluid model pushes massless carrier bodies too: every link keeps its own wrench there)
      if (c.task != DART_TASK_SNAKE && M.mass[p] == (Real)0 && M.child_start[p + 1] - M.child_start[p] == 1 && M.pre_ident[i] &&
          M.post_ident[p] && still)
        M.group_leader[p] = M.group_leader[i];
    }
    int gd[SP_MAXL], maxg = 0;
    for (int i = 0; i < nl; i++) {
      const int p = M.parent[i];
      gd[i] = p < 0 ? 0 : (M.group_leader[p] == M.group_leader[i] ? gd[p] : gd[p] + 1);
      if (gd[i] > maxg) maxg = gd[i];
    }
    for (int i = 0; i < nl; i++) M.group_level[i] = M.group_leader[i] == i ? gd[i] : -1;
    M.n_group_levels = maxg + 1;
    M.n_root_trans = 0;
    for (int i = 0; i < nl; i++) if (M.root_trans[i]) { if (M.n_root_trans >= 8) return "more than 8 root translation links"; M.root_trans_link[M.n_root_trans++] = i; }
    for (int i = 0; i < nl; i++) {
      Real* g = M.lconst[i];
      for (int t = 0; t < 9; t++) { g[LC_RPRE + t] = M.Rpre[i][t]; g[LC_RPOST + t] = M.Rpost[i][t]; g[LC_INERTIA + t] = M.inertia[i][t]; }
      for (int t = 0; t < 3; t++) {
        g[LC_PPRE + t] = M.ppre[i][t]; g[LC_PPOST + t] = M.ppost[i][t]; g[LC_AXIS + t] = M.axis[i][t]; g[LC_COM + t] = M.com[i][t];
        // Rpost^T axis, Rpost^T ppost: world axis and joint origin follow from the link frame alone
        g[LC_AXR + t] = M.Rpost[i][t] * M.axis[i][0] + M.Rpost[i][3 + t] * M.axis[i][1] + M.Rpost[i][6 + t] * M.axis[i][2];
        g[LC_CPOST + t] = M.Rpost[i][t] * M.ppost[i][0] + M.Rpost[i][3 + t] * M.ppost[i][1] + M.Rpost[i][6 + t] * M.ppost[i][2];
      }
    }
    for (int i = 0; i < nl; i++) {
      double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (M.parent[i] >= 0) {   // parent link frame = parent joint frame * Rpost(parent) (prismatic parents do not rotate)
        int p = M.parent[i];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
          double s2 = 0; for (int t = 0; t < 3; t++) s2 += Rw[p][3 * a + t] * (double)M.Rpost[p][3 * t + b];
          Rp[3 * a + b] = s2;
        }
      }
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        double s2 = 0; for (int t = 0; t < 3; t++) s2 += Rp[3 * a + t] * (double)M.Rpre[i][3 * t + b];
        Rw[i][3 * a + b] = s2;
      }
      for (int a = 0; a < 3; a++)
        M.root_axis_world[i][a] = (Real)(Rw[i][3 * a] * (double)M.axis[i][0] + Rw[i][3 * a + 1] * (double)M.axis[i][1] + Rw[i][3 * a + 2] * (double)M.axis[i][2]);
    }
  }
  for (int d = 0; d < c.ndofs; d++) {
    M.limited[d] = c.limited[d]; M.lower[d] = (Real)c.lower[d]; M.upper[d] = (Real)c.upper[d];
    M.damp[d] = (Real)c.damping[d]; M.stiff[d] = (Real)c.stiffness[d]; M.rest[d] = (Real)c.rest[d];
    M.q0[d] = (Real)c.init_pos[d]; M.dq0[d] = (Real)c.init_vel[d];
  }
  int ns = 0;
  for (int s = 0; s < c.nshapes; s++) {
    if (!c.shape_collidable[s]) continue;
    if (c.shape_type[s] != DART_SH_CAPSULE && c.shape_type[s] != DART_SH_BOX) return "collidable shape must be a capsule or a box";
    if (ns >= SP_MAXS) return "too many collidable shapes";
    M.sh_link[ns] = body_link[c.shape_body[s]];
    M.sh_type[ns] = c.shape_type[s] == DART_SH_CAPSULE ? 0 : 1;
    double R[9], pp[3];
    mat4_to_Rp(c.shape_pose[s], R, pp);
    for (int k = 0; k < 9; k++) M.sh_R[ns][k] = (Real)R[k];
    for (int k = 0; k < 3; k++) { M.sh_p[ns][k] = (Real)pp[k]; M.sh_size[ns][k] = (Real)c.shape_size[s][k]; }
    ns++;
  }
  M.nshapes = ns;
  // link-link contact candidates (card.self_collision): box pairs whose bodies are not parent and child, in the
  // oracle's order (first shape ascending, then the second)
  M.npairs = 0;
  M.maxm = 36; M.maxcp = 12;
  if (c.self_collision) {
    int slot_of[DART_MAX_SHAPES];
    { int k = 0; for (int s2 = 0; s2 < c.nshapes; s2++) slot_of[s2] = c.shape_collidable[s2] ? k++ : -1; }
    for (int sa = 0; sa < c.nshapes; sa++)
      for (int sb = sa + 1; sb < c.nshapes; sb++) {
        const int ba = c.shape_body[sa], bb = c.shape_body[sb];
        if (c.shape_type[sa] != DART_SH_BOX || c.shape_type[sb] != DART_SH_BOX) continue;
        if (!c.shape_collidable[sa] || !c.shape_collidable[sb]) continue;
        if (ba == bb || c.parent[ba] == bb || c.parent[bb] == ba) continue;
        if (M.npairs >= SP_MAXPAIRS) return "too many self-collision pairs";
        M.pair_a[M.npairs] = slot_of[sa]; M.pair_b[M.npairs] = slot_of[sb]; M.npairs++;
      }
    if (M.npairs > 0) { M.maxm = 64; M.maxcp = 20; }
    if (M.npairs * 40 > sp_tri(M.maxm)) return "self-collision clipping workspace";
  }
  M.dt = (Real)c.dt; for (int k = 0; k < 3; k++) M.g[k] = (Real)c.gravity[k];
  M.ground_y = (Real)c.ground_y; M.mu = (Real)c.friction; M.erp_dt = (Real)(c.erp / c.dt); M.max_erv = (Real)c.max_erv;
  M.limit_erp_dt = (Real)(c.limit_erp / c.dt); M.cfm1 = (Real)(1.0 + c.cfm); M.ccfm1 = (Real)(1.0 + c.contact_cfm);
  if (physics_only) { M.task = 0; return ""; }   // dynamics getters: geometry, inertia and topology only
  if (c.task < DART_TASK_NONE || c.task > DART_TASK_DOG) return "task not served by the spatial kernel";
  M.task = c.task; M.frame_skip = c.frame_skip; M.act_dim = c.act_dim; M.obs_dim = c.obs_dim; M.act_dof0 = c.act_dof0;
  M.max_steps = c.max_episode_steps;
  if (c.act_dim > 32 || c.act_dof0 + c.act_dim > c.ndofs) return "action layout";
  for (int k = 0; k < c.act_dim; k++) { M.act_scale[k] = (Real)c.act_scale[k]; M.act_lo[k] = (Real)c.act_low[k]; M.act_hi[k] = (Real)c.act_high[k]; }
  for (int k = 0; k < 4; k++) M.aux_link[k] = (c.task == DART_TASK_HUMANWALKER) ? body_link[c.aux_body[k]] : 0;
  for (int k = 0; k < 8; k++) M.aux_real[k] = (Real)c.aux_real[k];
  for (int k = 0; k < 4; k++) M.aux_real2[k] = (Real)c.aux_real2[k];
  M.aux_real2[1] = (Real)c.angle_max;   // up / forward angle threshold (human_walker.py:124, walker3d.py:89)
  if (c.task == DART_TASK_HOPPER || c.task == DART_TASK_WALKER2D) {   // cards the planar kernels decline (all-capsule contacts)
    if (c.height_body < 0 || c.height_body >= c.nbodies || c.penalty_dof >= c.ndofs) return "task indices";
    M.aux_link[0] = body_link[c.height_body]; M.aux_link[1] = c.penalty_dof;
    const double ar[7] = {c.alive_bonus, c.ctrl_cost, c.limit_penalty, 0.0, c.height_lo, c.height_hi, c.penalty_margin};
    for (int k = 0; k < 7; k++) M.aux_real[k] = (Real)ar[k];
  }
  if (c.task == DART_TASK_REACHER2D || c.task == DART_TASK_REACHER3D) {
    if (c.aux_body[0] < 0 || c.aux_body[0] >= c.nbodies || c.obs_dim != 3 * c.ndofs + (c.task == DART_TASK_REACHER2D ? 5 : 6)) return "reacher card";
    M.aux_link[0] = body_link[c.aux_body[0]];
  }
  if (c.task == DART_TASK_DOUBLE_PENDULUM) {
    if (c.ndofs != 3 || c.aux_body[0] < 0 || c.aux_body[0] >= c.nbodies || c.aux_body[1] < 0 || c.aux_body[1] >= c.nbodies) return "double pendulum card";
    M.aux_link[0] = body_link[c.aux_body[0]]; M.aux_link[1] = body_link[c.aux_body[1]];
  }
  if (c.task == DART_TASK_CARTPOLE_SWINGUP && c.ndofs != 2) return "swing-up card";
  if (c.task == DART_TASK_CARTPOLE || c.task == DART_TASK_HALFCHEETAH) { M.aux_real[0] = (Real)c.alive_bonus; M.aux_real[1] = (Real)c.ctrl_cost; }
  M.envdt = (Real)(c.dt * c.frame_skip);
  for (int d = 0; d < c.ndofs; d++) { M.spd_kp[d] = (Real)c.spd_kp[d]; M.spd_kd[d] = (Real)c.spd_kd[d]; }
  if (c.task == DART_TASK_DOG) {   // aux_real = {alive, velocity weight, ctrl cost, max side deviation, height lo, height hi}
    M.aux_link[0] = body_link[c.aux_body[0]];
    const double ar[6] = {c.alive_bonus, c.aux_real[0], c.ctrl_cost, c.aux_real[1], c.height_lo, c.height_hi};
    for (int k = 0; k < 6; k++) M.aux_real[k] = (Real)ar[k];
  }
  if (c.task == DART_TASK_WALKER3D_SPD) {   // same epilogue as Walker3d: reward = aux_real2[2] dx/dt + alive - ctrl sum a^2 - dev |z|
    M.aux_link[0] = body_link[c.aux_body[0]]; M.aux_link[1] = -1; M.aux_link[2] = -1;
    const double ar[7] = {c.alive_bonus, c.ctrl_cost, 0.0, c.aux_real[0], c.height_lo, c.height_hi, c.penalty_margin};
    for (int k = 0; k < 7; k++) M.aux_real[k] = (Real)ar[k];
    M.aux_real2[2] = (Real)c.aux_real[1];   // velocity-reward weight 0.45
  }
  if (c.task == DART_TASK_WALKER3D) {
    M.aux_real2[2] = (Real)1;
    M.aux_link[0] = body_link[c.aux_body[0]]; M.aux_link[1] = c.aux_body[1]; M.aux_link[2] = c.aux_body[2];
    if (c.aux_body[1] < 0 || c.aux_body[1] >= c.ndofs || c.aux_body[2] < 0 || c.aux_body[2] >= c.ndofs) return "penalty dof index";
    const double ar[7] = {c.alive_bonus, c.ctrl_cost, c.limit_penalty, c.aux_real[0], c.height_lo, c.height_hi, c.penalty_margin};
    for (int k = 0; k < 7; k++) M.aux_real[k] = (Real)ar[k];
  }
  M.s_max = (Real)c.state_abs_max; M.v_clip = (Real)c.obs_vel_clip; M.noise = (Real)c.reset_noise; M.noise_v = (Real)c.reset_noise_vel;
  M.inv_envdt = (Real)(1.0 / (c.dt * c.frame_skip));
  M.solver_iters = 600; M.pgs_fallback_sweeps = 600; M.stats = nullptr; M.dbg = nullptr; M.creport = nullptr; M.creport_count = nullptr; M.cf_report = nullptr;
  if (c.task == DART_TASK_NONE && c.obs_dim != 2 * c.ndofs) return "physics-only obs must be [q, dq]";
  return "";
}

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
    lds = sp_lds_bytes(M.nl, M.n, sizeof(Real), M.maxm, M.maxcp);
    // lean / pairs / extras instantiations of the step kernel (see sp_world_step)
    pairs = M.npairs > 0;
    const void* fns[4] = {(const void*)sp_step_kernel<Real, false, false>, (const void*)sp_step_kernel<Real, true, false>,
                          (const void*)sp_step_kernel<Real, false, true>, (const void*)sp_step_kernel<Real, true, true>};
    for (const void* fn : fns)
      if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)sp_reset_kernel<Real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    return hipSuccess;
  }
  void release() override {
    if (dM) (void)hipFree(dM); if (init_h) (void)hipFree(init_h); if (d_ext) (void)hipFree(d_ext); if (d_cf) (void)hipFree(d_cf);
    if (d_creport) (void)hipFree(d_creport); if (d_ccount) (void)hipFree(d_ccount); if (d_cfrep) (void)hipFree(d_cfrep); d_creport = nullptr; d_ccount = nullptr; d_cfrep = nullptr;
    dM = nullptr; init_h = nullptr; d_ext = nullptr; d_cf = nullptr;
  }
  void upload() { if (dM) (void)hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice); }
  hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act, float* obs,
                  float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) override {
#define SP_LAUNCH(P, X, R)                                                                                              \
  hipLaunchKernelGGL((sp_step_kernel<Real, P, X, R>), dim3((unsigned)n), dim3(64), lds, s, dM, n, (Real*)q, (Real*)dq, init_h, el, ep, \
                     act, obs, rew, done, trunc, autoreset, seed, off)
    if (M.creport) SP_LAUNCH(true, true, true);   // contact reporting lives in the most general instantiation only
    else if (pairs) { if (extras) SP_LAUNCH(true, true, false); else SP_LAUNCH(true, false, false); }
    else { if (extras) SP_LAUNCH(false, true, false); else SP_LAUNCH(false, false, false); }
#undef SP_LAUNCH
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
  bool pairs = false, extras = false;   // which instantiation of the step kernel this model runs
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
  hipError_t debug_dump(double* out) override { return dbg ? hipMemcpy(out, dbg, sizeof(double) * 160 * (size_t)nenv, hipMemcpyDeviceToHost) : hipErrorInvalidValue; }
  int slots() const override { return M.maxm; }
  int max_contacts() const override { return M.maxcp; }
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
      p->P.stats = nullptr;
      p->is_static = true;
      return p;
    }
  }
  auto p = std::make_unique<ImplT<Real, T>>();
  p->P = R;
  return p;
}

template <class Real>
std::unique_ptr<Impl> make_impl(const DartModelCard& c, std::string& why, bool allow_static) {
  const char* fs = getenv("DART_FORCE_SPATIAL");   // testing aid: run planar models through the general kernel
  const bool force_spatial = (fs && fs[0] == '1') || c.generic_kernel != 0;
  why = "hopper-chain: ";
  if (!force_spatial)
  if (auto p = make_for_topology<Real, HopperTopo, HopperStatic<Real>>(c, why, allow_static)) return p;
  why += "; hopper-chain, all capsules: ";
  if (!force_spatial)
  if (auto p = make_for_topology<Real, HopperAllTopo, void>(c, why, allow_static)) return p;
  why += "; walker2d-tree: ";
  if (!force_spatial)
  if (auto p = make_for_topology<Real, Walker2dTopo, Walker2dStatic<Real>>(c, why, allow_static)) return p;
  why += "; spatial: ";
  {
    auto p = std::make_unique<SpatialImplT<Real>>();
    std::string w = fill_spatial<Real>(c, p->M, false, p->body_link_map);
    p->extras = c.generic_kernel != 0 || c.task == DART_TASK_SNAKE || c.task == DART_TASK_WALKER3D_SPD || p->M.has_joint_friction != 0 ||
                p->M.free_root != 0;
    if (w.empty()) return p;
    why += w;
  }
  return nullptr;
}

}  // namespace

struct DartStepper {
  DartModelCard card;
  int64_t n = 0;
  int device = 0, precision = 32;
  hipStream_t stream = nullptr;
  std::unique_ptr<Impl> impl;
  void *q = nullptr, *dq = nullptr;
  int32_t* elapsed = nullptr;
  uint32_t* episode = nullptr;
  float *d_act = nullptr, *d_obs = nullptr, *d_rew = nullptr;
  uint8_t *d_done = nullptr, *d_trunc = nullptr, *d_mask = nullptr;
  double *d_qn = nullptr, *d_vn = nullptr;
  unsigned long long* d_stats = nullptr;
  uint32_t* mt = nullptr;        // MT19937 bank [624][N] (dart_seed_mt19937)
  int32_t* mt_pos = nullptr;
  double *d_init_pos = nullptr, *d_init_vel = nullptr;
  int noise_mode = 0;            // 0: Philox / host-supplied noise, 1: device MT19937 bank (reference-exact)
  float *h_act = nullptr, *h_obs = nullptr, *h_rew = nullptr;
  uint8_t *h_done = nullptr, *h_trunc = nullptr, *h_mask = nullptr;
  double *h_qn = nullptr, *h_vn = nullptr;
  int solver = 0, it1 = 0, it2 = 0, autoreset = 0;   // it1/it2 = 0: the implementation's default iteration cap
  uint64_t seed = 0, env_offset = 0;
  bool pending = false;
  double *d_ep_ret = nullptr, *d_last_ret = nullptr, *d_ep_tot = nullptr;   // DART_CFG_EPISODE_STATS
  int32_t *d_ep_len = nullptr, *d_last_len = nullptr;
  bool ep_stats = false;
  void* dyn_model = nullptr;     // device SpatialModel<float|double> used by dart_get_dynamics (built on first use)
  size_t dyn_lds = 0;
  double *d_dynM = nullptr, *d_dync = nullptr, *d_tstage = nullptr, *d_pose = nullptr;
  bool dyn_free_root = false;
  double* d_tvals = nullptr;     // reach targets drawn on the device (mt_draw)
  hipEvent_t ev_in = nullptr, ev_out = nullptr;   // ordering between the handle's stream and a caller-supplied one
  std::string err;
};

#define CHK(h, expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
      return DART_E_HIP;                                                                     \
    }                                                                                        \
  } while (0)

template <class Real>
static int dynamics_impl(DartStepper* h, double* mass, double* bias, double* rot = nullptr, double* pos = nullptr, double* com = nullptr) {
  const size_t N = (size_t)h->n, nd = (size_t)h->card.ndofs, nb = (size_t)h->card.nbodies;
  const bool poses = rot || pos || com;
  if (!h->dyn_model) {
    auto M = std::make_unique<SpatialModel<Real>>();
    std::string w = fill_spatial<Real>(h->card, *M, true);
    if (!w.empty()) { h->err = "dynamics getters: " + w; return DART_E_UNSUPPORTED; }
    h->dyn_free_root = M->free_root != 0;
    CHK(h, hipMalloc(&h->dyn_model, sizeof(SpatialModel<Real>)));
    CHK(h, hipMemcpy(h->dyn_model, M.get(), sizeof(SpatialModel<Real>), hipMemcpyHostToDevice));
    h->dyn_lds = sp_lds_bytes(M->nl, M->n, sizeof(Real), M->maxm, M->maxcp);
    CHK(h, hipFuncSetAttribute((const void*)sp_dynamics_kernel<Real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->dyn_lds));
    CHK(h, hipMalloc((void**)&h->d_dynM, sizeof(double) * N * nd * nd));
    CHK(h, hipMalloc((void**)&h->d_dync, sizeof(double) * N * nd));
  }
  if ((mass || bias) && h->dyn_free_root) {
    h->err = "dynamics getters: free root joint (the kernel's internal coordinates differ from DART's)"; return DART_E_UNSUPPORTED;
  }
  if (poses && !h->d_pose) CHK(h, hipMalloc((void**)&h->d_pose, sizeof(double) * N * nb * 15));
  hipLaunchKernelGGL((sp_dynamics_kernel<Real>), dim3((unsigned)N), dim3(64), h->dyn_lds, h->stream,
                     (const SpatialModel<Real>*)h->dyn_model, h->n, (const Real*)h->q, (const Real*)h->dq, h->impl->soa ? 1 : 0,
                     mass ? h->d_dynM : nullptr, bias ? h->d_dync : nullptr, poses ? h->d_pose : nullptr, (int)nb);
  CHK(h, hipGetLastError());
  if (mass) CHK(h, hipMemcpyAsync(mass, h->d_dynM, sizeof(double) * N * nd * nd, hipMemcpyDeviceToHost, h->stream));
  if (bias) CHK(h, hipMemcpyAsync(bias, h->d_dync, sizeof(double) * N * nd, hipMemcpyDeviceToHost, h->stream));
  CHK(h, hipStreamSynchronize(h->stream));
  if (poses) {
    std::vector<double> tmp(N * nb * 15);
    CHK(h, hipMemcpy(tmp.data(), h->d_pose, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N * nb; i++) {
      const double* t = tmp.data() + i * 15;
      if (rot) for (int k = 0; k < 9; k++) rot[i * 9 + k] = t[k];
      if (pos) for (int k = 0; k < 3; k++) pos[i * 3 + k] = t[9 + k];
      if (com) for (int k = 0; k < 3; k++) com[i * 3 + k] = t[12 + k];
    }
  }
  return DART_OK;
}

extern "C" {

const char* dart_last_error(const DartStepper* h) { return h ? h->err.c_str() : g_err.c_str(); }

int dart_create(const DartModelCard* card, int64_t num_envs, int device, int precision, DartStepper** out) {
  if (!out) { g_err = "out is NULL"; return DART_E_INVALID; }
  *out = nullptr;
  if (!card || card->version != DART_CARD_VERSION || card->struct_bytes != (int32_t)sizeof(DartModelCard)) {
    g_err = "model card version/size mismatch"; return DART_E_INVALID;
  }
  if (num_envs <= 0 || (precision != 32 && precision != 64)) { g_err = "bad num_envs/precision"; return DART_E_INVALID; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_err = "no HIP device available (this library has no CPU path)"; return DART_E_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_err = "device index out of range"; return DART_E_INVALID; }
  std::string why;
  const char* gen = getenv("DART_GENERIC_KERNEL");   // debugging aid: force the runtime-parameter kernel
  bool allow_static = !(gen && gen[0] == '1');
  std::unique_ptr<Impl> impl = precision == 32 ? make_impl<float>(*card, why, allow_static)
                                                : make_impl<double>(*card, why, allow_static);
  if (!impl) { g_err = "no compiled kernel for this model: " + why; return DART_E_UNSUPPORTED; }
  auto h = new DartStepper();
  h->card = *card; h->n = num_envs; h->device = device; h->precision = precision; h->impl = std::move(impl);
  h->impl->set_solver(h->solver, h->it1, h->it2);
  int rc = [&]() -> int {
    CHK(h, hipSetDevice(device));
    CHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CHK(h, h->impl->prepare(num_envs));
    size_t rs = precision == 32 ? 4 : 8, nd = (size_t)card->ndofs, N = (size_t)num_envs;
    CHK(h, hipMalloc(&h->q, rs * nd * N));
    CHK(h, hipMalloc(&h->dq, rs * nd * N));
    CHK(h, hipMalloc((void**)&h->elapsed, 4 * N));
    CHK(h, hipMalloc((void**)&h->episode, 4 * N));
    CHK(h, hipMalloc((void**)&h->d_act, 4 * N * card->act_dim));
    CHK(h, hipMalloc((void**)&h->d_obs, 4 * N * card->obs_dim));
    CHK(h, hipMalloc((void**)&h->d_rew, 4 * N));
    CHK(h, hipMalloc((void**)&h->d_done, N));
    CHK(h, hipMalloc((void**)&h->d_trunc, N));
    CHK(h, hipMalloc((void**)&h->d_mask, N));
    CHK(h, hipMalloc((void**)&h->d_qn, 8 * N * nd));
    CHK(h, hipMalloc((void**)&h->d_vn, 8 * N * nd));
    CHK(h, hipHostMalloc((void**)&h->h_act, 4 * N * card->act_dim));
    CHK(h, hipHostMalloc((void**)&h->h_obs, 4 * N * card->obs_dim));
    CHK(h, hipHostMalloc((void**)&h->h_rew, 4 * N));
    CHK(h, hipHostMalloc((void**)&h->h_done, N));
    CHK(h, hipHostMalloc((void**)&h->h_trunc, N));
    CHK(h, hipHostMalloc((void**)&h->h_mask, N));
    CHK(h, hipHostMalloc((void**)&h->h_qn, 8 * N * nd));
    CHK(h, hipHostMalloc((void**)&h->h_vn, 8 * N * nd));
    CHK(h, hipMemsetAsync(h->elapsed, 0, 4 * N, h->stream));
    CHK(h, hipMemsetAsync(h->episode, 0, 4 * N, h->stream));
    // initial state = init_pos / init_vel for every env
    for (size_t i = 0; i < N; i++)
      for (size_t d = 0; d < nd; d++) { h->h_qn[i * nd + d] = card->init_pos[d]; h->h_vn[i * nd + d] = card->init_vel[d]; }
    CHK(h, hipMemcpyAsync(h->d_qn, h->h_qn, 8 * N * nd, hipMemcpyHostToDevice, h->stream));
    CHK(h, hipMemcpyAsync(h->d_vn, h->h_vn, 8 * N * nd, hipMemcpyHostToDevice, h->stream));
    CHK(h, h->impl->state_io(h->stream, h->n, h->q, h->dq, h->d_qn, h->d_vn, 1));
    CHK(h, hipStreamSynchronize(h->stream));
    return DART_OK;
  }();
  if (rc != DART_OK) { g_err = h->err; dart_destroy(h); return rc; }
  *out = h;
  return DART_OK;
}

int dart_destroy(DartStepper* h) {
  if (!h) return DART_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->impl) h->impl->release();
  void* dev[] = {h->q, h->dq, h->elapsed, h->episode, h->d_act, h->d_obs, h->d_rew, h->d_done, h->d_trunc, h->d_mask, h->d_qn, h->d_vn, h->d_stats, h->mt, h->mt_pos, h->d_init_pos, h->d_init_vel, h->dyn_model, h->d_dynM, h->d_dync, h->d_tstage, h->d_pose, h->d_tvals, h->d_ep_ret, h->d_last_ret, h->d_ep_tot, h->d_ep_len, h->d_last_len};
  for (void* p : dev) if (p) hipFree(p);
  void* host[] = {h->h_act, h->h_obs, h->h_rew, h->h_done, h->h_trunc, h->h_mask, h->h_qn, h->h_vn};
  for (void* p : host) if (p) hipHostFree(p);
  if (h->ev_in) hipEventDestroy(h->ev_in);
  if (h->ev_out) hipEventDestroy(h->ev_out);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return DART_OK;
}

int dart_query(const DartStepper* h, int what, int64_t* out) {
  if (!h || !out) return DART_E_INVALID;
  switch (what) {
    case DART_Q_NUM_ENVS: *out = h->n; break;
    case DART_Q_NDOFS: *out = h->card.ndofs; break;
    case DART_Q_OBS_DIM: *out = h->card.obs_dim; break;
    case DART_Q_ACT_DIM: *out = h->card.act_dim; break;
    case DART_Q_FRAME_SKIP: *out = h->card.frame_skip; break;
    case DART_Q_PRECISION: *out = h->precision; break;
    case DART_Q_DEVICE: *out = h->device; break;
    case DART_Q_LCP_SLOTS: *out = h->impl->slots(); break;
    case DART_Q_STATIC_KERNEL: *out = h->impl->is_static ? 1 : 0; break;
    case DART_Q_MAX_CONTACTS: *out = h->impl->max_contacts(); break;
    default: return DART_E_INVALID;
  }
  return DART_OK;
}

int dart_configure(DartStepper* h, int key, double value) {
  if (!h) return DART_E_INVALID;
  switch (key) {
    case DART_CFG_SOLVER: h->solver = (int)value; break;
    case DART_CFG_ITERS_STAGE1: h->it1 = (int)value; break;
    case DART_CFG_ITERS_STAGE2: h->it2 = (int)value; break;
    case DART_CFG_AUTORESET: h->autoreset = value != 0; break;
    case DART_CFG_SEED: h->seed = (uint64_t)value; break;
    case DART_CFG_ENV_OFFSET: h->env_offset = (uint64_t)value; break;
    case DART_CFG_STATS:
      if (value != 0 && !h->d_stats) {
        CHK(h, hipMalloc((void**)&h->d_stats, 64 * sizeof(unsigned long long)));
        CHK(h, hipMemset(h->d_stats, 0, 64 * sizeof(unsigned long long)));
      }
      h->impl->set_stats(value != 0 ? h->d_stats : nullptr);
      break;
    case DART_CFG_CONTACT_REPORT: {
      const int rc = h->impl->set_contact_report(value != 0, h->n);
      if (rc == DART_E_UNSUPPORTED) h->err = "contact reporting: only the generic kernel implements it (card.generic_kernel = 1)";
      if (rc != DART_OK) return rc;
    } break;
    case DART_CFG_EPISODE_STATS:
      if (value != 0 && !h->d_ep_ret) {
        const size_t N = (size_t)h->n;
        CHK(h, hipMalloc((void**)&h->d_ep_ret, 8 * N)); CHK(h, hipMalloc((void**)&h->d_last_ret, 8 * N));
        CHK(h, hipMalloc((void**)&h->d_ep_len, 4 * N)); CHK(h, hipMalloc((void**)&h->d_last_len, 4 * N));
        CHK(h, hipMalloc((void**)&h->d_ep_tot, 8 * 3));
        CHK(h, hipMemset(h->d_ep_ret, 0, 8 * N)); CHK(h, hipMemset(h->d_last_ret, 0, 8 * N));
        CHK(h, hipMemset(h->d_ep_len, 0, 4 * N)); CHK(h, hipMemset(h->d_last_len, 0, 4 * N));
        CHK(h, hipMemset(h->d_ep_tot, 0, 8 * 3));
      }
      h->ep_stats = value != 0;
      break;
    case DART_CFG_BLOCK_THREADS:
      if (value != 64 && value != 32 && value != 16) { h->err = "block threads must be 16, 32 or 64"; return DART_E_INVALID; }
      h->impl->block_threads = (int)value; break;
    default: h->err = "unknown configure key"; return DART_E_INVALID;
  }
  if (h->solver < 0 || h->solver > 1 || h->it1 < 0 || h->it2 < 0) { h->err = "bad solver setting"; return DART_E_INVALID; }
  h->impl->set_solver(h->solver, h->it1, h->it2);
  return DART_OK;
}

// episode return / length accumulators (DART_CFG_EPISODE_STATS): after every step, on the step's stream
static int episode_accumulate(DartStepper* h, hipStream_t s, const float* d_rew, const uint8_t* d_done) {
  if (!h->ep_stats) return DART_OK;
  hipLaunchKernelGGL(episode_stats_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, h->n, d_rew, d_done, h->d_ep_ret,
                     h->d_ep_len, h->d_last_ret, h->d_last_len, h->d_ep_tot);
  CHK(h, hipGetLastError());
  return DART_OK;
}
static int episode_restart(DartStepper* h, hipStream_t s, const uint8_t* d_mask) {
  if (!h->ep_stats) return DART_OK;
  hipLaunchKernelGGL(episode_reset_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, h->n, d_mask, h->d_ep_ret, h->d_ep_len);
  CHK(h, hipGetLastError());
  return DART_OK;
}

// MT19937 mode: draw reference-exact reset noise on the device for the masked envs into d_qn / d_vn
// reset_model() of the masked envs from the MT19937 bank: the two noise vectors every env draws (hopper.py:78-79) and, for the
// tasks whose reset_model draws more from the same stream, that as well (swing-up sign, reach targets -> device task state)
static int mt_draw(DartStepper* h, hipStream_t s, const uint8_t* d_mask) {
  const double r = h->card.reset_noise, rv = h->card.reset_noise_vel;
  const int extra = h->card.task == DART_TASK_CARTPOLE_SWINGUP ? MT_EXTRA_SWINGUP
                  : h->card.task == DART_TASK_REACHER2D ? MT_EXTRA_REACHER2D
                  : h->card.task == DART_TASK_REACHER3D ? MT_EXTRA_REACHER3D : MT_EXTRA_NONE;
  const bool targets = extra == MT_EXTRA_REACHER2D || extra == MT_EXTRA_REACHER3D;
  if (targets && !h->d_tvals) CHK(h, hipMalloc((void**)&h->d_tvals, sizeof(double) * 4 * (size_t)h->n));
  dim3 grid((unsigned)((h->n + 127) / 128)), block(128);
  hipLaunchKernelGGL(mt_draw_kernel, grid, block, 0, s, h->n, (int)h->card.ndofs, h->mt, h->mt_pos, d_mask, -r, r - (-r), -rv,
                     rv - (-rv), h->d_init_pos, h->d_init_vel, h->d_qn, h->d_vn, extra, h->d_tvals);
  CHK(h, hipGetLastError());
  if (targets) return h->impl->set_task_state(s, d_mask, h->d_tvals, h->n);   // before the reset kernel computes the observation
  return DART_OK;
}

int dart_seed_mt19937(DartStepper* h, const uint32_t* keys, const int32_t* key_len) {
  if (!h || !keys || !key_len) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  const size_t N = (size_t)h->n, nd = (size_t)h->card.ndofs;
  if (!h->mt) {
    CHK(h, hipMalloc((void**)&h->mt, sizeof(uint32_t) * 624 * N));
    CHK(h, hipMalloc((void**)&h->mt_pos, sizeof(int32_t) * N));
    CHK(h, hipMalloc((void**)&h->d_init_pos, sizeof(double) * nd));
    CHK(h, hipMalloc((void**)&h->d_init_vel, sizeof(double) * nd));
    CHK(h, hipMemcpy(h->d_init_pos, h->card.init_pos, sizeof(double) * nd, hipMemcpyHostToDevice));
    CHK(h, hipMemcpy(h->d_init_vel, h->card.init_vel, sizeof(double) * nd, hipMemcpyHostToDevice));
  }
  uint32_t* dk = nullptr; int32_t* dl = nullptr;
  CHK(h, hipMalloc((void**)&dk, sizeof(uint32_t) * 2 * N));
  CHK(h, hipMalloc((void**)&dl, sizeof(int32_t) * N));
  CHK(h, hipMemcpy(dk, keys, sizeof(uint32_t) * 2 * N, hipMemcpyHostToDevice));
  CHK(h, hipMemcpy(dl, key_len, sizeof(int32_t) * N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mt_seed_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, h->stream, h->n, h->mt, h->mt_pos, dk, dl);
  CHK(h, hipGetLastError());
  CHK(h, hipStreamSynchronize(h->stream));
  (void)hipFree(dk); (void)hipFree(dl);
  h->noise_mode = 1;
  return DART_OK;
}

int dart_reset(DartStepper* h, const uint8_t* mask, const double* qpos_noise, const double* qvel_noise, float* obs_out) {
  if (!h) return DART_E_INVALID;
  if ((qpos_noise == nullptr) != (qvel_noise == nullptr)) { h->err = "give both noise arrays or neither"; return DART_E_INVALID; }
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  size_t N = (size_t)h->n, nd = (size_t)h->card.ndofs;
  const uint8_t* dmask = nullptr;
  if (mask) {
    memcpy(h->h_mask, mask, N);
    CHK(h, hipMemcpyAsync(h->d_mask, h->h_mask, N, hipMemcpyHostToDevice, h->stream));
    dmask = h->d_mask;
  }
  const double *dqn = nullptr, *dvn = nullptr;
  if (qpos_noise) {
    // world.reset() puts q, dq at init_pos/init_vel before the noise is added (hopper.py:77-79)
    for (size_t i = 0; i < N; i++) {
      if (mask && !mask[i]) continue;
      for (size_t d = 0; d < nd; d++) {
        h->h_qn[i * nd + d] = h->card.init_pos[d] + qpos_noise[i * nd + d];
        h->h_vn[i * nd + d] = h->card.init_vel[d] + qvel_noise[i * nd + d];
      }
    }
    CHK(h, hipMemcpyAsync(h->d_qn, h->h_qn, 8 * N * nd, hipMemcpyHostToDevice, h->stream));
    CHK(h, hipMemcpyAsync(h->d_vn, h->h_vn, 8 * N * nd, hipMemcpyHostToDevice, h->stream));
    dqn = h->d_qn; dvn = h->d_vn;
  } else if (h->noise_mode == 1) {
    int rc = mt_draw(h, h->stream, dmask);
    if (rc != DART_OK) return rc;
    dqn = h->d_qn; dvn = h->d_vn;
  }
  CHK(h, h->impl->reset(h->stream, h->n, h->q, h->dq, h->elapsed, h->episode, dmask, dqn, dvn,
                        obs_out ? h->d_obs : nullptr, h->seed, h->env_offset));
  { int rc = episode_restart(h, h->stream, dmask); if (rc != DART_OK) return rc; }
  if (obs_out) CHK(h, hipMemcpyAsync(h->h_obs, h->d_obs, 4 * N * h->card.obs_dim, hipMemcpyDeviceToHost, h->stream));
  CHK(h, hipStreamSynchronize(h->stream));
  if (obs_out) memcpy(obs_out, h->h_obs, 4 * N * h->card.obs_dim);
  return DART_OK;
}

// A caller-supplied stream shares the handle's state buffers with the handle's own stream: make it wait for the work already
// enqueued there (ext_begin) and make the handle's stream wait for what the caller's stream was just given (ext_end).
static int ext_begin(DartStepper* h, hipStream_t s) {
  if (s == h->stream) return DART_OK;
  if (!h->ev_in) { CHK(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming)); CHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming)); }
  CHK(h, hipEventRecord(h->ev_in, h->stream));
  CHK(h, hipStreamWaitEvent(s, h->ev_in, 0));
  return DART_OK;
}
static int ext_end(DartStepper* h, hipStream_t s) {
  if (s == h->stream) return DART_OK;
  CHK(h, hipEventRecord(h->ev_out, s));
  CHK(h, hipStreamWaitEvent(h->stream, h->ev_out, 0));
  return DART_OK;
}

int dart_reset_device(DartStepper* h, const uint8_t* d_mask, float* d_obs, void* hip_stream) {
  if (!h) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  { int rc = ext_begin(h, s); if (rc != DART_OK) return rc; }
  const double *dqn = nullptr, *dvn = nullptr;
  if (h->noise_mode == 1) {
    int rc = mt_draw(h, s, d_mask);
    if (rc != DART_OK) return rc;
    dqn = h->d_qn; dvn = h->d_vn;
  }
  CHK(h, h->impl->reset(s, h->n, h->q, h->dq, h->elapsed, h->episode, d_mask, dqn, dvn, d_obs, h->seed, h->env_offset));
  { int rc = episode_restart(h, s, d_mask); if (rc != DART_OK) return rc; }
  return ext_end(h, s);
}

static int state_copy(DartStepper* h, double* q, double* dq, int to_device) {
  if (!h || !q || !dq) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  size_t bytes = 8 * (size_t)h->n * h->card.ndofs;
  if (to_device) {
    memcpy(h->h_qn, q, bytes); memcpy(h->h_vn, dq, bytes);
    CHK(h, hipMemcpyAsync(h->d_qn, h->h_qn, bytes, hipMemcpyHostToDevice, h->stream));
    CHK(h, hipMemcpyAsync(h->d_vn, h->h_vn, bytes, hipMemcpyHostToDevice, h->stream));
    CHK(h, h->impl->state_io(h->stream, h->n, h->q, h->dq, h->d_qn, h->d_vn, 1));
    CHK(h, hipStreamSynchronize(h->stream));
  } else {
    CHK(h, h->impl->state_io(h->stream, h->n, h->q, h->dq, h->d_qn, h->d_vn, 0));
    CHK(h, hipMemcpyAsync(h->h_qn, h->d_qn, bytes, hipMemcpyDeviceToHost, h->stream));
    CHK(h, hipMemcpyAsync(h->h_vn, h->d_vn, bytes, hipMemcpyDeviceToHost, h->stream));
    CHK(h, hipStreamSynchronize(h->stream));
    memcpy(q, h->h_qn, bytes); memcpy(dq, h->h_vn, bytes);
  }
  return DART_OK;
}
int dart_set_state(DartStepper* h, const double* q, const double* dq) { return state_copy(h, (double*)q, (double*)dq, 1); }
int dart_get_state(DartStepper* h, double* q, double* dq) { return state_copy(h, q, dq, 0); }

int dart_step_async(DartStepper* h, const float* actions) {
  if (!h || !actions) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async called while a step is pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  size_t N = (size_t)h->n;
  memcpy(h->h_act, actions, 4 * N * h->card.act_dim);
  CHK(h, hipMemcpyAsync(h->d_act, h->h_act, 4 * N * h->card.act_dim, hipMemcpyHostToDevice, h->stream));
  const bool mt_reset = h->autoreset && h->noise_mode == 1;
  CHK(h, h->impl->step(h->stream, h->n, h->q, h->dq, h->elapsed, h->episode, h->d_act, h->d_obs, h->d_rew, h->d_done,
                       h->d_trunc, mt_reset ? 0 : h->autoreset, h->seed, h->env_offset));
  { int rc = episode_accumulate(h, h->stream, h->d_rew, h->d_done); if (rc != DART_OK) return rc; }
  if (mt_reset) {   // done envs: MT19937 noise, reset, post-reset observation (sync_vector_env.py:77-78)
    int rc = mt_draw(h, h->stream, h->d_done);
    if (rc != DART_OK) return rc;
    CHK(h, h->impl->reset(h->stream, h->n, h->q, h->dq, h->elapsed, h->episode, h->d_done, h->d_qn, h->d_vn, h->d_obs, h->seed,
                          h->env_offset, 1));
  }
  CHK(h, hipMemcpyAsync(h->h_obs, h->d_obs, 4 * N * h->card.obs_dim, hipMemcpyDeviceToHost, h->stream));
  CHK(h, hipMemcpyAsync(h->h_rew, h->d_rew, 4 * N, hipMemcpyDeviceToHost, h->stream));
  CHK(h, hipMemcpyAsync(h->h_done, h->d_done, N, hipMemcpyDeviceToHost, h->stream));
  CHK(h, hipMemcpyAsync(h->h_trunc, h->d_trunc, N, hipMemcpyDeviceToHost, h->stream));
  h->pending = true;
  return DART_OK;
}

int dart_step_wait(DartStepper* h, float* obs_out, double* reward_out, uint8_t* done_out, uint8_t* truncated_out) {
  if (!h) return DART_E_INVALID;
  if (!h->pending) { h->err = "step_wait called without step_async"; return DART_E_NOT_PENDING; }
  h->pending = false;
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  size_t N = (size_t)h->n;
  if (obs_out) memcpy(obs_out, h->h_obs, 4 * N * h->card.obs_dim);
  if (reward_out) for (size_t i = 0; i < N; i++) reward_out[i] = (double)h->h_rew[i];
  if (done_out) memcpy(done_out, h->h_done, N);
  if (truncated_out) memcpy(truncated_out, h->h_trunc, N);
  return DART_OK;
}

int dart_host_views(DartStepper* h, const float** obs, const float** reward_f32, const uint8_t** done, const uint8_t** truncated) {
  if (!h) return DART_E_INVALID;
  if (obs) *obs = h->h_obs;
  if (reward_f32) *reward_f32 = h->h_rew;
  if (done) *done = h->h_done;
  if (truncated) *truncated = h->h_trunc;
  return DART_OK;
}

int dart_step(DartStepper* h, const float* actions, float* obs_out, double* reward_out, uint8_t* done_out,
              uint8_t* truncated_out) {
  int rc = dart_step_async(h, actions);
  if (rc != DART_OK) return rc;
  return dart_step_wait(h, obs_out, reward_out, done_out, truncated_out);
}

int dart_step_device(DartStepper* h, const float* d_actions, float* d_obs, float* d_reward, uint8_t* d_done,
                     uint8_t* d_truncated, void* hip_stream) {
  if (!h || !d_actions) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  { int rc = ext_begin(h, s); if (rc != DART_OK) return rc; }
  const bool mt_reset = h->autoreset && h->noise_mode == 1;
  float* o = d_obs ? d_obs : h->d_obs;
  uint8_t* dn = d_done ? d_done : h->d_done;
  CHK(h, h->impl->step(s, h->n, h->q, h->dq, h->elapsed, h->episode, d_actions, o, d_reward ? d_reward : h->d_rew, dn,
                       d_truncated ? d_truncated : h->d_trunc, mt_reset ? 0 : h->autoreset, h->seed, h->env_offset));
  { int rc = episode_accumulate(h, s, d_reward ? d_reward : h->d_rew, dn); if (rc != DART_OK) return rc; }
  if (mt_reset) {
    int rc = mt_draw(h, s, dn);
    if (rc != DART_OK) return rc;
    CHK(h, h->impl->reset(s, h->n, h->q, h->dq, h->elapsed, h->episode, dn, h->d_qn, h->d_vn, o, h->seed, h->env_offset, 1));
  }
  return ext_end(h, s);
}

int dart_get_counters(DartStepper* h, int32_t* elapsed, uint32_t* episode) {
  if (!h) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  if (elapsed) CHK(h, hipMemcpy(elapsed, h->elapsed, 4 * (size_t)h->n, hipMemcpyDeviceToHost));
  if (episode) CHK(h, hipMemcpy(episode, h->episode, 4 * (size_t)h->n, hipMemcpyDeviceToHost));
  return DART_OK;
}

int dart_get_stats(DartStepper* h, uint64_t* hist64, int clear) {
  if (!h || !hist64) return DART_E_INVALID;
  if (!h->d_stats) { h->err = "enable DART_CFG_STATS first"; return DART_E_INVALID; }
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  CHK(h, hipMemcpy(hist64, h->d_stats, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (clear) CHK(h, hipMemset(h->d_stats, 0, 64 * sizeof(unsigned long long)));
  return DART_OK;
}

int dart_debug_dump(DartStepper* h, double* out160) {
  if (!h || !out160) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  CHK(h, h->impl->debug_dump(out160));
  return DART_OK;
}

int dart_set_task_state(DartStepper* h, const uint8_t* mask, const double* values) {
  if (!h || !values) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  const size_t N = (size_t)h->n;
  if (!h->d_tstage) CHK(h, hipMalloc((void**)&h->d_tstage, 8 * 4 * N));   // staging buffer, kept for the handle's lifetime
  double* dv = h->d_tstage;
  CHK(h, hipMemcpy(dv, values, 8 * 4 * N, hipMemcpyHostToDevice));
  const uint8_t* dmask = nullptr;
  if (mask) { memcpy(h->h_mask, mask, N); CHK(h, hipMemcpy(h->d_mask, h->h_mask, N, hipMemcpyHostToDevice)); dmask = h->d_mask; }
  int rc = h->impl->set_task_state(h->stream, dmask, dv, h->n);
  hipError_t e = hipStreamSynchronize(h->stream);
  if (rc == DART_E_UNSUPPORTED) h->err = "this model's kernel keeps no per-env task state";
  if (rc != DART_OK) return rc;
  CHK(h, e);
  return DART_OK;
}

int dart_set_ext_force(DartStepper* h, int body, const double* force) {
  if (!h) return DART_E_INVALID;
  if (force && (body < 0 || body >= h->card.nbodies)) { h->err = "body index"; return DART_E_INVALID; }
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  int rc = h->impl->set_ext_force(body, force, h->n);
  if (rc == DART_E_UNSUPPORTED) h->err = "external body forces need the generic kernel: set card.generic_kernel = 1 before dart_create";
  else if (rc != DART_OK) h->err = "dart_set_ext_force: HIP error";
  return rc;
}

int dart_get_episode_stats(DartStepper* h, double* last_return, int32_t* last_length, double* totals3, int clear_totals) {
  if (!h) return DART_E_INVALID;
  if (!h->d_ep_ret) { h->err = "enable DART_CFG_EPISODE_STATS first"; return DART_E_INVALID; }
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  const size_t N = (size_t)h->n;
  if (last_return) CHK(h, hipMemcpy(last_return, h->d_last_ret, 8 * N, hipMemcpyDeviceToHost));
  if (last_length) CHK(h, hipMemcpy(last_length, h->d_last_len, 4 * N, hipMemcpyDeviceToHost));
  if (totals3) CHK(h, hipMemcpy(totals3, h->d_ep_tot, 8 * 3, hipMemcpyDeviceToHost));
  if (clear_totals) CHK(h, hipMemset(h->d_ep_tot, 0, 8 * 3));
  return DART_OK;
}

int dart_get_dynamics(DartStepper* h, double* mass_matrix, double* coriolis_gravity) {
  if (!h || (!mass_matrix && !coriolis_gravity)) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  return h->precision == 32 ? dynamics_impl<float>(h, mass_matrix, coriolis_gravity) : dynamics_impl<double>(h, mass_matrix, coriolis_gravity);
}

int dart_get_body_poses(DartStepper* h, double* rotation, double* origin, double* com) {
  if (!h || (!rotation && !origin && !com)) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  CHK(h, hipSetDevice(h->device));
  return h->precision == 32 ? dynamics_impl<float>(h, nullptr, nullptr, rotation, origin, com)
                            : dynamics_impl<double>(h, nullptr, nullptr, rotation, origin, com);
}

int dart_get_contacts(DartStepper* h, int32_t* count, int32_t* bodies, double* point_force, int32_t max_contacts) {
  if (!h || !count || max_contacts < 0) return DART_E_INVALID;
  if (h->pending) { h->err = "dart_get_contacts while a step is pending"; return DART_E_PENDING; }
  const int rc = h->impl->get_contacts(h->stream, h->n, count, bodies, point_force, max_contacts);
  if (rc == DART_E_INVALID) h->err = "dart_get_contacts: enable DART_CFG_CONTACT_REPORT before stepping";
  if (rc == DART_E_UNSUPPORTED) h->err = "contact reporting: only the generic kernel implements it (card.generic_kernel = 1)";
  return rc;
}

int dart_get_constraint_forces(DartStepper* h, double* constraint_forces) {
  if (!h || !constraint_forces) return DART_E_INVALID;
  if (h->pending) { h->err = "dart_get_constraint_forces while a step is pending"; return DART_E_PENDING; }
  const int rc = h->impl->get_constraint_forces(h->stream, h->n, constraint_forces);
  if (rc == DART_E_INVALID) h->err = "dart_get_constraint_forces: enable DART_CFG_CONTACT_REPORT before stepping";
  if (rc == DART_E_UNSUPPORTED) h->err = "constraint forces: only the generic kernel reports them (card.generic_kernel = 1)";
  return rc;
}

// everything that persists between steps, in a fixed order
static void snapshot_buffers(DartStepper* h, std::vector<std::pair<void*, size_t>>& v) {
  const size_t N = (size_t)h->n, nd = (size_t)h->card.ndofs, rs = h->precision == 32 ? 4 : 8;
  v.push_back({h->q, rs * nd * N}); v.push_back({h->dq, rs * nd * N});
  v.push_back({h->elapsed, 4 * N}); v.push_back({h->episode, 4 * N});
  if (h->mt) { v.push_back({h->mt, 4 * 624 * N}); v.push_back({h->mt_pos, 4 * N}); }
  if (h->d_ep_ret) {
    v.push_back({h->d_ep_ret, 8 * N}); v.push_back({h->d_last_ret, 8 * N}); v.push_back({h->d_ep_len, 4 * N});
    v.push_back({h->d_last_len, 4 * N}); v.push_back({h->d_ep_tot, 8 * 3});
  }
  h->impl->persistent(v, h->n);
}

int dart_snapshot(DartStepper* h, void* buf, uint64_t* nbytes) {
  if (!h || !nbytes) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  std::vector<std::pair<void*, size_t>> v;
  snapshot_buffers(h, v);
  uint64_t total = 16;
  for (auto& b : v) total += b.second;
  if (!buf) { *nbytes = total; return DART_OK; }
  if (*nbytes < total) { h->err = "dart_snapshot: buffer too small"; *nbytes = total; return DART_E_INVALID; }
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  unsigned char* p = (unsigned char*)buf;
  const uint64_t head[2] = {0x44415254534e4150ull /* "DARTSNAP" */, total};
  memcpy(p, head, 16); p += 16;
  for (auto& b : v) { CHK(h, hipMemcpy(p, b.first, b.second, hipMemcpyDeviceToHost)); p += b.second; }
  *nbytes = total;
  return DART_OK;
}

int dart_restore(DartStepper* h, const void* buf, uint64_t nbytes) {
  if (!h || !buf) return DART_E_INVALID;
  if (h->pending) { h->err = "step_async pending"; return DART_E_PENDING; }
  std::vector<std::pair<void*, size_t>> v;
  snapshot_buffers(h, v);
  uint64_t total = 16;
  for (auto& b : v) total += b.second;
  uint64_t head[2];
  if (nbytes < 16) { h->err = "dart_restore: not a snapshot"; return DART_E_INVALID; }
  memcpy(head, buf, 16);
  if (head[0] != 0x44415254534e4150ull || head[1] != total || nbytes < total) {
    h->err = "dart_restore: snapshot of a different handle configuration (model, num_envs, precision, seeding / statistics modes)";
    return DART_E_INVALID;
  }
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  const unsigned char* p = (const unsigned char*)buf + 16;
  for (auto& b : v) { CHK(h, hipMemcpy(b.first, p, b.second, hipMemcpyHostToDevice)); p += b.second; }
  return DART_OK;
}

int dart_sync(DartStepper* h) {
  if (!h) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  CHK(h, hipStreamSynchronize(h->stream));
  return DART_OK;
}

int dart_time_steps(DartStepper* h, const float* d_actions, int action_batches, float* d_obs, float* d_reward,
                    uint8_t* d_done, uint8_t* d_truncated, int steps, double* ms_per_step) {
  if (!h || !d_actions || action_batches <= 0 || steps <= 0 || !ms_per_step) return DART_E_INVALID;
  CHK(h, hipSetDevice(h->device));
  hipEvent_t e0, e1;
  CHK(h, hipEventCreate(&e0));
  CHK(h, hipEventCreate(&e1));
  size_t stride = (size_t)h->n * h->card.act_dim;
  CHK(h, hipEventRecord(e0, h->stream));
  for (int i = 0; i < steps; i++) {
    int rc = dart_step_device(h, d_actions + (size_t)(i % action_batches) * stride, d_obs, d_reward, d_done, d_truncated, nullptr);
    if (rc != DART_OK) return rc;
  }
  CHK(h, hipEventRecord(e1, h->stream));
  CHK(h, hipEventSynchronize(e1));
  float ms = 0;
  CHK(h, hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  *ms_per_step = (double)ms / steps;
  return DART_OK;
}

}  // extern "C"
