// spatial_build.hpp -- host code: DartModelCard -> SpatialModel (multi-dof joints expanded into chains of 1-dof links with massless
// carriers).  Shared by the tree kernel's host side (spatial_impl.hpp) and the 3-D chain lane kernel (chain3d_kernel.hpp through
// planar_impl.hpp); no device code in here.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/dart_stepper.h"
#include "spatial_model.hpp"

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
  M.n_mpairs = 0;
  for (int i = 0; i < nl; i++) {   // the dofs that move link i
    M.anc_dofs[i] = 0u;
    for (int j = i; j >= 0; j = M.parent[j]) if (M.dof[j] >= 0) M.anc_dofs[i] |= 1u << M.dof[j];
  }
  for (int d = 0; d < c.ndofs; d++) {   // mass-matrix entries (d, dj): dj on the path of d's link to the root
    const int i = M.dof_link[d];
    for (int j = i; j >= 0; j = M.parent[j]) {
      if (M.dof[j] < 0) continue;
      const int a = c.ndofs - 1 - d, b = c.ndofs - 1 - M.dof[j];
      M.mpair_off[M.n_mpairs] = (uint16_t)HI(a, b);      // the padded dense rows; SpatialImplT::choose_lds re-addresses them for a pattern kernel's skyline
      M.mpairs[M.n_mpairs++] = (uint32_t)i | ((uint32_t)j << 8) | ((uint32_t)d << 16) | ((uint32_t)M.dof[j] << 24);
    }
  }
  M.hreals = HR(sp_npad(c.ndofs));
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
  M.impulse_M = c.impulse_inertia == DART_IMPULSE_MASS ? 1 : 0;
  M.has_implicit = 0;
  M.fd_passes = 2;
  for (int d = 0; d < c.ndofs; d++) if (c.damping[d] != 0.0 || c.stiffness[d] != 0.0) M.has_implicit = 1;
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
  M.solver_iters = 600; M.pgs_fallback_sweeps = 600; M.stats = nullptr; M.sched_perm = nullptr; M.sched_cost = nullptr; M.dbg = nullptr; M.creport = nullptr; M.creport_count = nullptr; M.cf_report = nullptr;
  if (c.task == DART_TASK_NONE && c.obs_dim != 2 * c.ndofs) return "physics-only obs must be [q, dq]";
  return "";
}

}  // namespace dartk
