// spatial_impl.hpp -- host side of the tree kernel: card -> SpatialModel (multi-dof joints expanded into 1-dof links), launches,
// contact report / external force / task state plumbing, and the dynamics-getter model.
// Included by spatial_f32.hip / spatial_f64.hip only.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "impl_iface.hpp"
#include "spatial_kernel.hpp"

namespace dartk {

// ------------------------------------------------------------------ general 3-D skeletons (spatial_kernel.hpp)
static inline void mat4_to_Rp(const double* T, double* R, double* p) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j]; p[i] = T[4 * i + 3]; }
}
static inline void inv_Rp(const double* R, const double* p, double* Ri, double* pi) {
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
    pairs = M.npairs > 0;
    big = M.n >= 20 && !M.free_root;
    pattern = matches_pattern<HumanWalkerPattern>() ? 1 : 0;
    is_static = pattern != 0 && uses_big() && !extras && !pairs;   // DART_Q_STATIC_KERNEL: the step kernel is specialised for this model's tree at compile time
    choose_lds();
    (void)hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice);
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
    if (dM) (void)hipFree(dM); if (init_h) (void)hipFree(init_h); if (d_ext) (void)hipFree(d_ext); if (d_cf) (void)hipFree(d_cf);
    if (d_creport) (void)hipFree(d_creport); if (d_ccount) (void)hipFree(d_ccount); if (d_cfrep) (void)hipFree(d_cfrep); d_creport = nullptr; d_ccount = nullptr; d_cfrep = nullptr;
    dM = nullptr; init_h = nullptr; d_ext = nullptr; d_cf = nullptr;
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
    return true;
  }
  // register-LCP models without contact reporting drop the LDS solver's workspace (sp_carve): smaller block, more workgroups per CU
  bool uses_big() const {   // which step-kernel instantiation step() launches
    if (M.creport) return false;
    return pairs ? (extras || big) : (!extras && big);
  }
  void choose_lds() {
    M.reg_lcp = (uses_big() && M.npairs == 0 && M.maxm <= 40) ? 1 : 0;
    lds = sp_lds_bytes(M.nl, M.n, sizeof(Real), M.maxm, M.maxcp, M.reg_lcp);
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
    else if (big && pattern == 1)   // the factor's sparsity is known at compile time (tree_patterns.hpp)
      hipLaunchKernelGGL((sp_step_kernel<Real, false, false, false, true, HumanWalkerPattern>), dim3((unsigned)n), dim3(64), lds, s, dM, n, (Real*)q,
                         (Real*)dq, init_h, el, ep, act, obs, rew, done, trunc, autoreset, seed, off);
    else if (big) SP_LAUNCH(false, false, false, true);
    else SP_LAUNCH(false, false, false, false);
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
