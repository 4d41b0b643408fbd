// spatial_tasks.hpp -- task epilogues of the tree kernel: reward / done / observation of every served env id.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include "spatial_world_step.hpp"

namespace dartk {

// ------------------------------------------------------------------ task epilogues (lane 0, after a fresh kinematics pass)
// HumanWalker: reward / done / obs (human_walker.py:75-149).  Returns done.
template <class Real>
__device__ __forceinline__ bool sp_humanwalker_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before,
                                                        Real abs_a_sum, Real init_height, const int* cflags,
                                                        Real& reward_out) {
  const V3<Real> roff = ld3(S.misc);
  const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
  const Real* Lh = S.link + Md.aux_link[1] * SP_LINKF;
  const Real pos_after = Lb[LK_C] + roff.x;
  const Real height = Lh[LK_C + 1] + roff.y, side = Lh[LK_C + 2] + roff.z;
  const Real* R = Lh + LK_R;
  const V3<Real> up = v3<Real>(R[1], R[4], R[7]), fw = v3<Real>(R[0], R[3], R[6]);
  const Real ang_u = acos(fmin(fmax(up.y / sqrt(dot(up, up)), Real(-1)), Real(1)));
  const Real ang_f = acos(fmin(fmax(fw.x / sqrt(dot(fw, fw)), Real(-1)), Real(1)));
  const Real vel = (pos_after - pos_before) * Md.inv_envdt;
  const Real tv = Md.aux_real[0];
  Real rew = Real(2) * (tv - fabs(tv - vel)) + Md.aux_real[1] - Md.aux_real[2] * abs_a_sum - Md.aux_real[3] * fabs(side);
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  const Real dh = height - init_height;
  ok = ok && (dh > Md.aux_real[4]) && (dh < Md.aux_real[5]) && (fabs(ang_u) < Md.aux_real2[1]) && (fabs(ang_f) < Md.aux_real2[1]) &&
       (fabs(S.q[3]) < Md.aux_real[6]) && (fabs(S.q[5]) < Md.aux_real[7]) && (fabs(side) < Md.aux_real2[0]);
  if (!ok) rew = Real(0);
  reward_out = rew;
  (void)cflags;
  return !ok;
}

// Walker3d: reward / done (walker3d.py:44-97).  Progress, height, side deviation and the up / forward angles are those
// of bodynodes[0] (aux_link[0]); aux_link[1..2] are the two penalised dofs; aux_real = {alive, ctrl_cost, limit_penalty,
// deviation_pen, height_lo, height_hi, penalty_margin}.  Returns done.
template <class Real>
__device__ __forceinline__ bool sp_walker3d_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before,
                                                     Real sq_a_sum, Real& reward_out) {
  const V3<Real> roff = ld3(S.misc);
  const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
  const Real pos_after = Lb[LK_C] + roff.x, height = Lb[LK_C + 1] + roff.y, side = Lb[LK_C + 2] + roff.z;
  const Real* R = Lb + LK_R;
  const V3<Real> up = v3<Real>(R[1], R[4], R[7]), fw = v3<Real>(R[0], R[3], R[6]);
  const Real ang_u = acos(fmin(fmax(up.y / sqrt(dot(up, up)), Real(-1)), Real(1)));
  const Real ang_f = acos(fmin(fmax(fw.x / sqrt(dot(fw, fw)), Real(-1)), Real(1)));
  Real pen = Real(0);
  for (int k = 1; k <= 2; k++) {
    const int j = Md.aux_link[k];
    if (j < 0) continue;
    if ((Md.lower[j] - S.q[j]) > -Md.aux_real[6]) pen += Real(1.5);
    if ((Md.upper[j] - S.q[j]) < Md.aux_real[6]) pen += Real(1.5);
  }
  Real rew = Md.aux_real2[2] * ((pos_after - pos_before) * Md.inv_envdt) + Md.aux_real[0];   // weight 1 (Walker3d) / 0.45 (SPD)
  rew -= Md.aux_real[1] * sq_a_sum;
  rew -= Md.aux_real[2] * pen;
  rew -= Md.aux_real[3] * fabs(side);
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  ok = ok && (height > Md.aux_real[4]) && (height < Md.aux_real[5]) && (fabs(ang_u) < Md.aux_real2[1]) && (fabs(ang_f) < Md.aux_real2[1]);
  if (!ok && Md.task == 3) rew = Real(0);   // the SPD variant (task 12) keeps the reward of the terminal step
  reward_out = rew;
  return !ok;
}

// Hopper / Walker2d task logic for cards the planar kernels do not take (e.g. every capsule collidable): hopper.py:36-65,
// walker2d.py:22-62.  aux_link = {height body link, penalty dof or -1}; aux_real = {alive, ctrl_cost, limit_penalty, -,
// height_lo, height_hi, penalty_margin}.  Returns done; *height_out feeds observation[0].
template <class Real>
__device__ __forceinline__ bool sp_planar_task_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before, Real sq_a_sum,
                                                        Real& reward_out) {
  const Real height = S.link[Md.aux_link[0] * SP_LINKF + LK_C + 1] + S.misc[1];
  Real pen = Real(0);
  const int j = Md.aux_link[1];
  if (j >= 0) {
    if ((Md.lower[j] - S.q[j]) > -Md.aux_real[6]) pen += Real(1.5);
    if ((Md.upper[j] - S.q[j]) < Md.aux_real[6]) pen += Real(1.5);
  }
  Real rew = (S.q[0] - pos_before) * Md.inv_envdt;
  rew += Md.aux_real[0];
  rew -= Md.aux_real[1] * sq_a_sum;
  rew -= Md.aux_real[2] * pen;
  reward_out = rew;
  bool ok = true;
  for (int i = 0; i < Md.n; i++) {
    ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
    if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
  }
  ok = ok && (height > Md.aux_real[4]) && (height < Md.aux_real[5]) && (fabs(S.q[2]) < Md.aux_real2[1]);
  return !ok;
}

// Reacher tip: to_world(aux body, aux_real[0..2]) in absolute coordinates (no floating base in these models)
template <class Real>
__device__ __forceinline__ V3<Real> sp_reacher_tip(const SpatialModel<Real>& Md, SpLds<Real>& S) {
  const Real* L = S.link + Md.aux_link[0] * SP_LINKF;
  return ld3(L + LK_P) + mulR(L + LK_R, v3<Real>(Md.aux_real[0], Md.aux_real[1], Md.aux_real[2])) + ld3(S.misc);
}

// CartPole (cart_pole.py:12-24): reward 1, done when the observation is not finite or |q[1]| > angle_max.
// HalfCheetah (half_cheetah.py:43-63): aux_real = {alive, ctrl_cost}; reward zeroed when the state broke.
template <class Real>
__device__ __forceinline__ bool sp_simple_epilogue(const SpatialModel<Real>& Md, SpLds<Real>& S, Real pos_before, Real sq_a_sum,
                                                   Real& reward_out) {
  bool fin = true, bounded = true;
  for (int i = 0; i < Md.n; i++) {
    fin = fin && isfinite(S.q[i]) && isfinite(S.dq[i]);
    bounded = bounded && (fabs(S.dq[i]) < Md.s_max) && (i < 2 || fabs(S.q[i]) < Md.s_max);
  }
  if (Md.task == 5) {
    reward_out = Md.aux_real[0];
    return !(fin && fabs(S.q[1]) <= Md.aux_real2[1]);
  }
  if (Md.task == 9) {   // snake (snake_7link.py:72-84); aux_real = {alive, ctrl_cost, deviation cost, fluid k}
    Real rew = (S.q[0] - pos_before) * Md.inv_envdt;
    rew += Md.aux_real[0];
    rew -= Md.aux_real[1] * sq_a_sum;
    rew -= fabs(S.q[2]) * Md.aux_real[2];
    reward_out = rew;
    return !(fin && bounded && fabs(S.q[2]) < Md.aux_real2[1]);
  }
  if (Md.task == 7) {   // cart-pole swing-up (cartpole_swingup.py:22-31); sq_a_sum = a^2 of the single action
    reward_out = Md.aux_real[0] - fabs(S.q[1]) - Md.aux_real[1] * sq_a_sum - Md.aux_real[2] * fabs(S.q[0]);
    return (fabs(S.q[1]) > Md.aux_real[3]) || (fabs(S.dq[1]) > Md.aux_real[4]) || (fabs(S.q[0]) > Md.aux_real[5]);
  }
  if (Md.task == 8) {   // double inverted pendulum (inverted_double_pendulum.py:27-42): tip height above the cart
    const Real base = S.link[Md.aux_link[0] * SP_LINKF + LK_P + 1], raw = S.link[Md.aux_link[1] * SP_LINKF + LK_P + 1];
    const Real height = Real(2) * (raw - base - Md.aux_real[4]) / Md.aux_real[5];
    const Real dist_pen = Md.aux_real[1] * (S.q[0] * S.q[0]) + (height - Real(2)) * (height - Real(2));
    const Real vel_pen = Md.aux_real[2] * (S.dq[1] * S.dq[1]) + Md.aux_real[3] * (S.dq[2] * S.dq[2]);
    reward_out = Md.aux_real[0] - dist_pen - vel_pen;
    return height <= Real(1);
  }
  const bool ok = fin && bounded;
  Real rew = (S.q[0] - pos_before) * Md.inv_envdt + Md.aux_real[0];
  rew -= Md.aux_real[1] * sq_a_sum;
  reward_out = ok ? rew : Real(0);
  return !(ok && fabs(S.q[2]) < Md.aux_real2[1]);
}

template <class Real>
__device__ __forceinline__ void sp_write_obs(const SpatialModel<Real>& Md, SpLds<Real>& S, const int* cflags, float* __restrict__ o, int lane,
                                             int n_ = -1, int task_ = -1) {
  // n_ / task_: the caller's compile-time copies of Md.n / Md.task (pattern kernels), -1 = read the model
  const int n = n_ >= 0 ? n_ : Md.n, task = task_ >= 0 ? task_ : Md.task;
  if (task == 10 || task == 11) {   // reachers: cos q, sin q, target (2-D: x, z), dq, tip - target (reacher.py:38-42)
    const V3<Real> tgt = ld3(S.misc + 4);       // staged by the caller from the per-env task state
    int o0 = 2 * n;
    if (lane < n) { Real sn, cs; sincos_<Real>(S.q[lane], sn, cs); o[lane] = (float)cs; o[n + lane] = (float)sn; }
    if (lane == 0) {
      if (task == 10) { o[o0] = (float)tgt.x; o[o0 + 1] = (float)tgt.z; }
      else { o[o0] = (float)tgt.x; o[o0 + 1] = (float)tgt.y; o[o0 + 2] = (float)tgt.z; }
    }
    o0 += (task == 10) ? 2 : 3;
    if (lane < n) o[o0 + lane] = (float)S.dq[lane];
    if (lane == 0) {
      const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - tgt;
      o[o0 + n] = (float)vec.x; o[o0 + n + 1] = (float)vec.y; o[o0 + n + 2] = (float)vec.z;
    }
    return;
  }
  if (task == 8) {   // double pendulum: [q0, sin q1, sin q2, cos q1, cos q2, dq] (inverted_double_pendulum.py:45-51)
    if (lane == 0) o[0] = (float)S.q[0];
    if (lane == 1 || lane == 2) { Real sn, cs; sincos_<Real>(S.q[lane], sn, cs); o[lane] = (float)sn; o[lane + 2] = (float)cs; }
    if (lane < 3) o[5 + lane] = (float)S.dq[lane];
    return;
  }
  if (task == 0 || task == 5 || task == 7) {   // physics only, CartPole, swing-up: [q, dq]
    if (lane < n) { o[lane] = (float)S.q[lane]; o[n + lane] = (float)S.dq[lane]; }
    return;
  }
  if (lane >= 1 && lane < n) o[lane - 1] = (float)S.q[lane];
  if (lane < n) o[n - 1 + lane] = (float)fmin(fmax(S.dq[lane], -Md.v_clip), Md.v_clip);
  if (task == 4 && lane < 2) o[2 * n - 1 + lane] = (float)cflags[lane];   // foot-contact flags (human_walker.py:146)
  if ((task == 1 || task == 2) && lane == 1)   // observation[0] = COM height of the root body (hopper.py:72)
    o[0] = (float)(S.link[Md.aux_link[0] * SP_LINKF + LK_C + 1] + S.misc[1]);
}

}  // namespace dartk
