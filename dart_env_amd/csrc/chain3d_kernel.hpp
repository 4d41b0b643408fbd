// chain3d_kernel.hpp -- gfx950 device code for a fixed-base serial chain of revolute links in 3-D: the five-dof reacher
// (reacher.skel: universal - revolute - universal joints, expanded into five 1-dof links by spatial_build.hpp).
//
// Replaces, one env per lane, what the reference does per env in Python + DART:
//   DartReacherEnv.step / _get_obs               reference gym/envs/dart/reacher.py:17-42
//   DartEnv.do_simulation, TimeLimit.step, SyncVectorEnv auto-reset (as planar_kernel.hpp)
// The reach target is per-env task state (dart_set_task_state; reset_model resamples it, reacher.py:44-57).
//
// The recursions are the tree kernel's (spatial_dynamics.hpp: world-frame quantities about each link's joint origin, velocity-product
// accelerations, composite bodies, H = M + dt D + dt^2 K) written out link after link for ONE lane, everything in registers;
// explicit H^-1 and the boxed LCP by block principal pivoting as in the planar kernels (one joint-limit row per dof, and one
// Coulomb-friction row per dof in the FRIC instantiation).  This model used to run on the wave-per-env tree kernel with 5 of 64
// lanes busy.
#pragma once
#include "planar_kernel.hpp"
#include "spatial_model.hpp"

namespace dartk {

template <class Real, int NL>
struct Chain3Params {
  static constexpr int N = NL;
  Real dt, limit_erp_dt, max_erv, cfm1, g[3];
  // per link (spatial_build.hpp): joint frame in the parent link frame, child link frame in the moved joint frame, axis in the joint
  // frame and in the child link frame, joint origin in the child link frame, COM and inertia about the COM in the link frame
  Real Rpre[NL][9], ppre[NL][3], Rpost[NL][9], ppost[NL][3], axis[NL][3], axr[NL][3], cpost[NL][3], com[NL][3], inertia[NL][9], mass[NL];
  Real lo[NL], hi[NL], damp[NL], stiff[NL], rest[NL], q0[NL], dq0[NL], fric_dt[NL];
  Real sqe[NL];     // sqrt(dt damp + dt^2 stiff) per dof (planar_kernel.hpp: implicit_accel)
  int impulse_M;    // card.impulse_inertia (A3): 1 = impulses act on M (DART 6), 0 = on M + dt D + dt^2 K
  Real act_scale[NL], act_lo[NL], act_hi[NL];
  Real tip[3];            // the finger tip in the last link's frame (reacher.py:21)
  Real ctrl_w, done_dist; // reacher.py:25-33: reward = -dist - ctrl_w sum tau^2, done when the distance was below done_dist
  Real noise, noise_v;
  int frame_skip, max_steps, task, iters;
  Real* tstate;           // [n_envs][4] per-env task state: the reach target x, y, z
};

// world pose of every link: joint origins, axes, and the last link's frame (for the tip)
template <class Real, int NL>
__device__ __forceinline__ V3<Real> chain3d_tip(const Chain3Params<Real, NL>& P, const Real (&q)[NL]) {
  Real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3<Real> pp = v3<Real>(0, 0, 0);
  sfor<0, NL>([&](auto K) {
    constexpr int k = K;
    const V3<Real> ax = ld3(P.axis[k]);
    Real sn, cs;
    sincos_<Real>(q[k], sn, cs);
    const Real v = Real(1) - cs;
    const Real Rq[9] = {ax.x * ax.x * v + cs,        ax.x * ax.y * v - ax.z * sn, ax.x * ax.z * v + ax.y * sn,
                        ax.y * ax.x * v + ax.z * sn, ax.y * ax.y * v + cs,        ax.y * ax.z * v - ax.x * sn,
                        ax.z * ax.x * v - ax.y * sn, ax.z * ax.y * v + ax.x * sn, ax.z * ax.z * v + cs};
    Real T[9], Rl[9], R[9];
    mulRR(Rq, P.Rpost[k], T);
    const V3<Real> t = mulR(Rq, ld3(P.ppost[k]));
    mulRR(P.Rpre[k], T, Rl);
    const V3<Real> pl = ld3(P.ppre[k]) + mulR(P.Rpre[k], t);
    mulRR(Rp, Rl, R);
    pp = pp + mulR(Rp, pl);
    for (int c = 0; c < 9; c++) Rp[c] = R[c];
  });
  return pp + mulR(Rp, ld3(P.tip));
}

// one world step: q, dq in/out
template <class Real, int NL, bool FRIC>
__device__ __forceinline__ void chain3d_world_step(const Chain3Params<Real, NL>& P, Real (&q)[NL], Real (&dq)[NL], const Real (&tau)[NL]) {
  constexpr int N = NL;
  V3<Real> pj[N], a[N], F[N], Nm[N], Hm[N];
  Real MC[N], IC[N][6];
  {
    Real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3<Real> pp = v3<Real>(0, 0, 0), omp = pp, alp = pp, aop = pp;   // parent link: frame, origin, angular velocity / acceleration, origin acceleration
    const V3<Real> grav = ld3(P.g);
    sfor<0, N>([&](auto K) {
      constexpr int k = K;
      const V3<Real> ax = ld3(P.axis[k]);
      Real sn, cs;
      sincos_<Real>(q[k], sn, cs);
      const Real v = Real(1) - cs;
      const Real Rq[9] = {ax.x * ax.x * v + cs,        ax.x * ax.y * v - ax.z * sn, ax.x * ax.z * v + ax.y * sn,
                          ax.y * ax.x * v + ax.z * sn, ax.y * ax.y * v + cs,        ax.y * ax.z * v - ax.x * sn,
                          ax.z * ax.x * v - ax.y * sn, ax.z * ax.y * v + ax.x * sn, ax.z * ax.z * v + cs};
      Real T[9], Rl[9], R[9];
      mulRR(Rq, P.Rpost[k], T);
      const V3<Real> t = mulR(Rq, ld3(P.ppost[k]));
      mulRR(P.Rpre[k], T, Rl);
      const V3<Real> pl = ld3(P.ppre[k]) + mulR(P.Rpre[k], t);
      mulRR(Rp, Rl, R);
      const V3<Real> p = pp + mulR(Rp, pl);
      a[k] = mulR(R, ld3(P.axr[k]));
      pj[k] = p - mulR(R, ld3(P.cpost[k]));
      const V3<Real> c = p + mulR(R, ld3(P.com[k]));
      // angular velocity, velocity-product angular acceleration, velocity-product acceleration of the link origin
      const V3<Real> w = a[k] * dq[k];
      const V3<Real> om = omp + w;
      const V3<Real> al = alp + cross(omp, w);
      const V3<Real> r = pj[k] - pp, sv = p - pj[k];
      const V3<Real> ao = aop + cross(alp, r) + cross(omp, cross(omp, r)) + cross(al, sv) + cross(om, cross(om, sv));
      // wrench and composite seeds about the joint origin
      const Real m = P.mass[k];
      const V3<Real> dj = c - pj[k];
      Real RI[9], Iw[9];
      mulRR(R, P.inertia[k], RI);
      for (int x = 0; x < 3; x++)
        for (int y = 0; y < 3; y++) Iw[3 * x + y] = RI[3 * x] * R[3 * y] + RI[3 * x + 1] * R[3 * y + 1] + RI[3 * x + 2] * R[3 * y + 2];
      const V3<Real> dc = c - p;
      const V3<Real> ac = ao + cross(al, dc) + cross(om, cross(om, dc));
      const V3<Real> f = (ac - grav) * m;
      const V3<Real> nrm = mulR(Iw, al) + cross(om, mulR(Iw, om));
      F[k] = f; Nm[k] = nrm + cross(dj, f); MC[k] = m; Hm[k] = dj * m;
      const Real d2 = dot(dj, dj);
      IC[k][0] = Iw[0] + m * (d2 - dj.x * dj.x); IC[k][1] = Iw[1] - m * dj.x * dj.y; IC[k][2] = Iw[2] - m * dj.x * dj.z;
      IC[k][3] = Iw[4] + m * (d2 - dj.y * dj.y); IC[k][4] = Iw[5] - m * dj.y * dj.z; IC[k][5] = Iw[8] + m * (d2 - dj.z * dj.z);
      for (int cc = 0; cc < 9; cc++) Rp[cc] = R[cc];
      pp = p; omp = om; alp = al; aop = ao;
    });
  }
  // backward pass: fold every link's wrench and composite body into its parent, shifting the reference point to the parent's joint origin
  sfor_rev<1, N>([&](auto K) {
    constexpr int k = K, p = k - 1;
    const V3<Real> o = pj[k] - pj[p];
    Nm[p] = Nm[p] + Nm[k] + cross(o, F[k]);
    F[p] = F[p] + F[k];
    const Real mc = MC[k];
    const V3<Real> h = Hm[k];
    const Real diag = Real(2) * dot(o, h) + mc * dot(o, o);
    IC[p][0] += IC[k][0] + diag - Real(2) * h.x * o.x - mc * o.x * o.x;
    IC[p][1] += IC[k][1] - (h.x * o.y + o.x * h.y) - mc * o.x * o.y;
    IC[p][2] += IC[k][2] - (h.x * o.z + o.x * h.z) - mc * o.x * o.z;
    IC[p][3] += IC[k][3] + diag - Real(2) * h.y * o.y - mc * o.y * o.y;
    IC[p][4] += IC[k][4] - (h.y * o.z + o.y * h.z) - mc * o.y * o.z;
    IC[p][5] += IC[k][5] + diag - Real(2) * h.z * o.z - mc * o.z * o.z;
    Hm[p] = Hm[p] + h + o * mc;
    MC[p] += mc;
  });
  // H = M + dt D + dt^2 K (lower triangle), rhs = tau - C - D dq - K (q + dt dq - rest)
  Real H[N * (N + 1) / 2], rhs[N];
  sfor<0, N>([&](auto K) {
    constexpr int k = K;
    const V3<Real> Lm = cross(a[k], Hm[k]);
    const V3<Real> Kv = v3<Real>(IC[k][0] * a[k].x + IC[k][1] * a[k].y + IC[k][2] * a[k].z,
                                 IC[k][1] * a[k].x + IC[k][3] * a[k].y + IC[k][4] * a[k].z,
                                 IC[k][2] * a[k].x + IC[k][4] * a[k].y + IC[k][5] * a[k].z);
    sfor<0, k + 1>([&](auto J) {
      constexpr int j = J;
      H[tri(k, j)] = dot(a[j], Kv + cross(pj[k] - pj[j], Lm));
    });
    rhs[k] = tau[k] - dot(a[k], Nm[k]) - P.damp[k] * dq[k] - P.stiff[k] * (q[k] + P.dt * dq[k] - P.rest[k]);
    if (!P.impulse_M) H[tri(k, k)] += P.dt * P.damp[k] + P.dt * P.dt * P.stiff[k];
  });
  spd_inverse<Real, N>(H);   // inverse of the impulse inertia: M (DART 6) or M + E (card.impulse_inertia = 0)
  Real vs[N];
  {
    Real acc[N];
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real t = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; t += H[tri(i, j)] * rhs[j]; });
      acc[i] = t;
    });
    if (P.impulse_M) implicit_accel<Real, N, false, AllDofs<N>>(P, H, acc);   // qdd = (M + E)^-1 rhs from M^-1
    sfor<0, N>([&](auto I) { constexpr int i = I; vs[i] = dq[i] + P.dt * acc[i]; });
  }
  // LCP rows: joint limits at q_t (rows 0..N-1) and, with FRIC, Coulomb joint friction (rows N..2N-1); each acts on one dof
  constexpr int M = FRIC ? 2 * N : N;
  Real A[M * (M + 1) / 2], b[M], lo[M], hi[M], x[M];
  bool act[M], any = false;
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    const bool low = q[i] <= P.lo[i], up = (!low) && (q[i] >= P.hi[i]);
    const Real viol = low ? (q[i] - P.lo[i]) : (q[i] - P.hi[i]);
    const Real bounce = fmin(fmax(-viol * P.limit_erp_dt, -P.max_erv), P.max_erv);
    act[i] = low || up;
    b[i] = act[i] ? (bounce - vs[i]) : Real(0);
    lo[i] = low ? Real(0) : (up ? -inf_<Real>() : Real(0));
    hi[i] = low ? inf_<Real>() : Real(0);
    any = any || act[i];
    if constexpr (FRIC) {
      const bool fr = P.fric_dt[i] > Real(0);
      act[N + i] = fr;
      b[N + i] = fr ? -vs[i] : Real(0);
      lo[N + i] = fr ? -P.fric_dt[i] : Real(0);
      hi[N + i] = fr ? P.fric_dt[i] : Real(0);
      any = any || fr;
    }
  });
  if (__any(any)) {
    sfor<0, M>([&](auto I) {
      constexpr int i = I, di = i % N;
      sfor<0, i + 1>([&](auto J) {
        constexpr int j = J, dj = j % N;
        A[tri(i, j)] = (i == j) ? (act[i] ? H[tri(di, di)] * P.cfm1 : Real(1)) : ((act[i] && act[j]) ? H[tri(di, dj)] : Real(0));
      });
    });
    uint32_t pinmask = 0, Fs = 0, Us = 0;
    Real bmax0 = Real(0);
    sfor<0, M>([&](auto I) { bmax0 = fmax(bmax0, fabs(b[I])); });
    const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      x[i] = Real(0);
      const bool pinned = !(lo[i] < hi[i]);
      const bool upper = !(lo[i] == Real(0));
      const bool start_free = !pinned && (upper ? (b[i] < -tol0) : (b[i] > tol0));
      pinmask |= pinned ? (1u << i) : 0u;
      Fs |= start_free ? (1u << i) : 0u;
      Us |= (upper && !start_free) ? (1u << i) : 0u;
    });
    blcp_bpp<Real, M, !FRIC>(A, b, lo, hi, pinmask, Fs, Us, x, P.iters, nullptr);
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real dv = Real(0);
      sfor<0, N>([&](auto J) {
        constexpr int j = J;
        if constexpr (FRIC) dv += H[tri(i, j)] * (x[j] + x[N + j]);
        else dv += H[tri(i, j)] * x[j];
      });
      vs[i] += dv;
    });
  }
  sfor<0, N>([&](auto I) { constexpr int i = I; dq[i] = vs[i]; q[i] += P.dt * vs[i]; });
}

// observation (reacher.py:36-42): cos q, sin q, target, dq, tip - target
template <class Real, int NL>
__device__ __forceinline__ void chain3d_write_obs(const Chain3Params<Real, NL>& P, const Real (&q)[NL], const Real (&dq)[NL], const Real (&tgt)[3],
                                                  float* __restrict__ o) {
  sfor<0, NL>([&](auto K) { constexpr int k = K; Real sn, cs; sincos_<Real>(q[k], sn, cs); o[k] = (float)cs; o[NL + k] = (float)sn; });
  o[2 * NL] = (float)tgt[0]; o[2 * NL + 1] = (float)tgt[1]; o[2 * NL + 2] = (float)tgt[2];
  sfor<0, NL>([&](auto K) { constexpr int k = K; o[2 * NL + 3 + k] = (float)dq[k]; });
  const V3<Real> tip = chain3d_tip<Real, NL>(P, q);
  o[3 * NL + 3] = (float)(tip.x - tgt[0]); o[3 * NL + 4] = (float)(tip.y - tgt[1]); o[3 * NL + 5] = (float)(tip.z - tgt[2]);
}
template <int NL> __device__ __host__ constexpr int chain3d_obs_dim() { return 3 * NL + 6; }
// task 0 (round 5): a physics-only card of this shape (envs.DartEnv on a user's .skel): torques as given, obs = [q, dq], reward 0, never done
template <int NL> __device__ __host__ constexpr int chain3d_obs_dim_rt(int task) { return task == 0 ? 2 * NL : chain3d_obs_dim<NL>(); }
template <class Real, int NL>
__device__ __forceinline__ void chain3d_write_obs_rt(const Chain3Params<Real, NL>& P, const Real (&q)[NL], const Real (&dq)[NL], const Real (&tgt)[3], float* __restrict__ o) {
  if (P.task == 0) { sfor<0, NL>([&](auto K) { constexpr int k = K; o[k] = (float)q[k]; o[NL + k] = (float)dq[k]; }); return; }
  chain3d_write_obs<Real, NL>(P, q, dq, tgt, o);
}

template <class Real, int NL, bool FRIC>
__global__ void __launch_bounds__(64) chain3d_step_kernel(Chain3Params<Real, NL> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                           int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                           const float* __restrict__ actions, float* __restrict__ obs,
                                                           float* __restrict__ reward, uint8_t* __restrict__ done,
                                                           uint8_t* __restrict__ truncated, int autoreset, uint64_t seed, uint64_t env_offset) {
  constexpr int N = NL;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = e < n_envs;
  const int64_t ec = valid ? e : n_envs - 1;   // tail lanes shadow the last env so wave votes stay uniform
  Real q[N], dq[N], tau[N], tgt[3];
  sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + ec]; dq[i] = dqs[(int64_t)i * n_envs + ec]; });
  sfor<0, 3>([&](auto I) { constexpr int i = I; tgt[i] = P.tstate[4 * ec + i]; });
  int el_in = elapsed[ec];            // fetched with the state: a load issued in the epilogue would be a bare HBM round trip
  uint32_t ep_in = episode[ec];
  Real tau2 = Real(0);
  sfor<0, N>([&](auto K) {
    constexpr int k = K;
    const Real av = (Real)actions[ec * N + k];
    Real cl = (av > P.act_hi[k]) ? P.act_hi[k] : av;   // comparison clamp (reacher.py:17-22): a NaN action stays NaN
    cl = (cl < P.act_lo[k]) ? P.act_lo[k] : cl;
    tau[k] = P.task == 0 ? av : cl * P.act_scale[k];
    tau2 += tau[k] * tau[k];                           // reacher.py:26: the control cost takes the scaled, clamped torque
  });
  DART_PIN_VGPR(el_in); DART_PIN_VGPR(ep_in);   // pinned where the state loads are awaited anyway: the compiler must not sink them
  // reacher.py:23-33: reward and done use the tip-target distance BEFORE the step
  Real dist0;
  {
    const V3<Real> tip0 = chain3d_tip<Real, NL>(P, q);
    const Real vx = tip0.x - tgt[0], vy = tip0.y - tgt[1], vz = tip0.z - tgt[2];
    dist0 = sqrt(vx * vx + vy * vy + vz * vz);
  }
#pragma unroll 1
  for (int f = 0; f < P.frame_skip; ++f) chain3d_world_step<Real, NL, FRIC>(P, q, dq, tau);
  bool fin = true;
  sfor<0, N>([&](auto I) { constexpr int i = I; fin = fin && isfinite(q[i]) && isfinite(dq[i]); });
  const Real rew = P.task == 0 ? Real(0) : -dist0 - tau2 * P.ctrl_w;
  const bool task_done = P.task != 0 && !(fin && (dist0 > P.done_dist));
  int el = el_in + 1;
  const bool trunc = (P.max_steps > 0) && (el >= P.max_steps);
  const bool dn = task_done || trunc;
  if (autoreset && dn) {
    const uint32_t ep = ep_in + 1;
    reset_noise<Real, N>(seed, env_offset + (uint64_t)ec, ep, P.noise, P.noise_v, q, dq);
    sfor<0, N>([&](auto I) { constexpr int i = I; q[i] += P.q0[i]; dq[i] += P.dq0[i]; });
    el = 0;
    if (valid) episode[e] = ep;
  }
  if (valid) {
    sfor<0, N>([&](auto I) { constexpr int i = I; qs[(int64_t)i * n_envs + e] = q[i]; dqs[(int64_t)i * n_envs + e] = dq[i]; });
    elapsed[e] = el;
    chain3d_write_obs_rt<Real, NL>(P, q, dq, tgt, obs + e * chain3d_obs_dim_rt<NL>(P.task));
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = (trunc && !task_done) ? 1 : 0;
  }
}

template <class Real, int NL>
__global__ void __launch_bounds__(256) chain3d_reset_kernel(Chain3Params<Real, NL> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                             int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                             const uint8_t* __restrict__ mask, const double* __restrict__ qnoise,
                                                             const double* __restrict__ vnoise, float* __restrict__ obs, uint64_t seed,
                                                             uint64_t env_offset, int obs_masked_only) {
  constexpr int N = NL;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  Real q[N], dq[N], tgt[3];
  const bool m = (mask == nullptr) || mask[e];
  if (m) {
    if (qnoise) {
      sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = (Real)qnoise[e * N + i]; dq[i] = (Real)vnoise[e * N + i]; });
    } else {
      const uint32_t ep = episode[e] + 1;
      reset_noise<Real, N>(seed, env_offset + (uint64_t)e, ep, P.noise, P.noise_v, q, dq);
      sfor<0, N>([&](auto I) { constexpr int i = I; q[i] += P.q0[i]; dq[i] += P.dq0[i]; });
      episode[e] = ep;
    }
    sfor<0, N>([&](auto I) { constexpr int i = I; qs[(int64_t)i * n_envs + e] = q[i]; dqs[(int64_t)i * n_envs + e] = dq[i]; });
    elapsed[e] = 0;
  } else {
    sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + e]; dq[i] = dqs[(int64_t)i * n_envs + e]; });
  }
  if (obs && (m || !obs_masked_only)) {
    sfor<0, 3>([&](auto I) { constexpr int i = I; tgt[i] = P.tstate[4 * e + i]; });
    chain3d_write_obs_rt<Real, NL>(P, q, dq, tgt, obs + e * chain3d_obs_dim_rt<NL>(P.task));
  }
}

}  // namespace dartk
