// arm_kernel.hpp -- gfx950 device code for a fixed-base planar arm moving in the horizontal x-z plane: the two-link reacher
// (reacher2d.skel).
//
// Replaces, one env per lane, what the reference does per env in Python + DART:
//   DartReacher2dEnv.step / _get_obs             reference gym/envs/dart/reacher2d.py:17-45
//   DartEnv.do_simulation, TimeLimit.step, SyncVectorEnv auto-reset (as planar_kernel.hpp)
// The reach target is per-env task state (dart_set_task_state; reset_model resamples it, reacher2d.py:47-60).
//
// Same formulation as planar_kernel.hpp / cart_kernel.hpp (composite bodies about each link's joint origin, H = M + dt D + dt^2 K,
// explicit H^-1, boxed LCP by block principal pivoting) with the root fixed in the world.  In-plane coordinates (X, Y) = (x, z): a
// turn about +y is clockwise there, hence sigma = -1 for a +y axis; gravity is normal to the plane and drops out.  The LCP holds
// one joint-limit row and one Coulomb-friction row per dof (DART's JointCoulombFrictionConstraint: joint velocity -> 0 with an
// impulse within +-mu dt; reacher2d.skel is the asset that has <friction> on its joints).  This 2-dof model used to run on the
// wave-per-env tree kernel with 2 of 64 lanes busy.
#pragma once
#include "planar_kernel.hpp"

namespace dartk {

template <class Real, int NP>
struct ArmParams {
  static constexpr int N = NP;
  Real dt, limit_erp_dt, max_erv, cfm1;
  Real height;                      // y of the plane of motion (the tip's y in the observation)
  Real sigma[NP], mass[NP], cx[NP], cy[NP], izz[NP], jx[NP], jy[NP];   // link k; joint position in the parent frame (link 0: in the world)
  Real lo[NP], hi[NP];              // +-inf: no limit on that dof
  Real damp[NP], stiff[NP], rest[NP], q0[NP], dq0[NP];
  Real sqe[NP];     // sqrt(dt damp + dt^2 stiff) per dof (planar_kernel.hpp: implicit_accel)
  int impulse_M;    // card.impulse_inertia (A3): 1 = impulses act on M (DART 6), 0 = on M + dt D + dt^2 K
  Real fric_dt[NP];                 // Coulomb joint friction * dt (0 = none)
  Real tipx, tipy;                  // the finger tip (COM of the last body, reacher2d.py:31) in the last link's frame
  Real act_scale[NP], act_lo[NP], act_hi[NP];
  Real noise, noise_v;
  int frame_skip, max_steps, task, iters;
  Real* tstate;                     // [n_envs][4] per-env task state: the reach target x, y, z
};

// one world step: q, dq in/out
template <class Real, int NP>
__device__ __forceinline__ void arm_world_step(const ArmParams<Real, NP>& P, Real (&q)[NP], Real (&dq)[NP], const Real (&tau)[NP]) {
  constexpr int N = NP;
  Real c[N], s[N], px[N], py[N], lx[N], ly[N], om[N], apx[N], apy[N];
  Real mc[N], dcx[N], dcy[N], Ip[N], Fx[N], Fy[N], Nz[N];
  sfor<0, N>([&](auto K) {
    constexpr int k = K;
    Real sj, cj;
    sincos_<Real>(q[k], sj, cj);
    sj *= P.sigma[k];
    if constexpr (k == 0) {
      c[0] = cj; s[0] = sj; lx[0] = P.jx[0]; ly[0] = P.jy[0]; px[0] = lx[0]; py[0] = ly[0];
      om[0] = P.sigma[0] * dq[0]; apx[0] = Real(0); apy[0] = Real(0);
    } else {
      constexpr int p = k - 1;
      c[k] = c[p] * cj - s[p] * sj;
      s[k] = s[p] * cj + c[p] * sj;
      lx[k] = c[p] * P.jx[k] - s[p] * P.jy[k]; ly[k] = s[p] * P.jx[k] + c[p] * P.jy[k];
      px[k] = px[p] + lx[k]; py[k] = py[p] + ly[k];
      om[k] = om[p] + P.sigma[k] * dq[k];
      const Real w2p = om[p] * om[p];
      apx[k] = apx[p] - w2p * lx[k]; apy[k] = apy[p] - w2p * ly[k];
    }
    const Real ox = c[k] * P.cx[k] - s[k] * P.cy[k], oy = s[k] * P.cx[k] + c[k] * P.cy[k];
    const Real w2 = om[k] * om[k];
    const Real fx = P.mass[k] * (apx[k] - w2 * ox), fy = P.mass[k] * (apy[k] - w2 * oy);
    mc[k] = P.mass[k]; dcx[k] = P.mass[k] * ox; dcy[k] = P.mass[k] * oy;
    Ip[k] = P.izz[k] + P.mass[k] * (ox * ox + oy * oy);
    Fx[k] = fx; Fy[k] = fy; Nz[k] = ox * fy - oy * fx;
  });
  sfor_rev<1, N>([&](auto K) {
    constexpr int k = K, p = k - 1;
    Ip[p] += Ip[k] + Real(2) * (lx[k] * dcx[k] + ly[k] * dcy[k]) + mc[k] * (lx[k] * lx[k] + ly[k] * ly[k]);
    dcx[p] += dcx[k] + mc[k] * lx[k]; dcy[p] += dcy[k] + mc[k] * ly[k];
    mc[p] += mc[k];
    Nz[p] += Nz[k] + (lx[k] * Fy[k] - ly[k] * Fx[k]);
    Fx[p] += Fx[k]; Fy[p] += Fy[k];
  });
  Real H[N * (N + 1) / 2], rhs[N];
  sfor<0, N>([&](auto K) {
    constexpr int k = K;
    sfor<0, k + 1>([&](auto J) {
      constexpr int j = J;
      H[tri(k, j)] = P.sigma[k] * P.sigma[j] * (Ip[k] + dcx[k] * (px[k] - px[j]) + dcy[k] * (py[k] - py[j]));
    });
    rhs[k] = tau[k] - P.sigma[k] * Nz[k] - P.damp[k] * dq[k] - P.stiff[k] * (q[k] + P.dt * dq[k] - P.rest[k]);
    if (!P.impulse_M) H[tri(k, k)] += P.dt * P.damp[k] + P.dt * P.dt * P.stiff[k];
  });
  spd_inverse<Real, N>(H);   // inverse of the impulse inertia: M (DART 6) or M + E (card.impulse_inertia = 0)
  Real vs[N];
  {
    Real acc[N];
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real a = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; a += H[tri(i, j)] * rhs[j]; });
      acc[i] = a;
    });
    if (P.impulse_M) implicit_accel<Real, N, false, AllDofs<N>>(P, H, acc);   // qdd = (M + E)^-1 rhs from M^-1
    sfor<0, N>([&](auto I) { constexpr int i = I; vs[i] = dq[i] + P.dt * acc[i]; });
  }
  // LCP rows: joint limits at q_t (rows 0..N-1), Coulomb joint friction (rows N..2N-1); both act on a single dof
  constexpr int M = 2 * N;
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
    const bool fr = P.fric_dt[i] > Real(0);
    act[N + i] = fr;
    b[N + i] = fr ? -vs[i] : Real(0);
    lo[N + i] = fr ? -P.fric_dt[i] : Real(0);
    hi[N + i] = fr ? P.fric_dt[i] : Real(0);
    any = any || act[i] || fr;
  });
  if (__any(any)) {
    sfor<0, M>([&](auto I) {
      constexpr int i = I, di = i % N;
      sfor<0, i + 1>([&](auto J) {
        constexpr int j = J, dj = j % N;
        A[tri(i, j)] = (i == j) ? (act[i] ? H[tri(di, di)] * P.cfm1 : Real(1)) : ((act[i] && act[j]) ? H[tri(di, dj)] : Real(0));
      });
    });
    uint32_t pinmask = 0, F = 0, U = 0;
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
      F |= start_free ? (1u << i) : 0u;
      U |= (upper && !start_free) ? (1u << i) : 0u;
    });
    blcp_bpp<Real, M, false>(A, b, lo, hi, pinmask, F, U, x, P.iters, nullptr);
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real dv = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; dv += H[tri(i, j)] * (x[j] + x[N + j]); });
      vs[i] += dv;
    });
  }
  sfor<0, N>([&](auto I) { constexpr int i = I; dq[i] = vs[i]; q[i] += P.dt * vs[i]; });
}

// finger tip in the plane (X, Y) = (x, z)
template <class Real, int NP>
__device__ __forceinline__ void arm_tip(const ArmParams<Real, NP>& P, const Real (&q)[NP], Real& tx, Real& ty) {
  Real cc = Real(1), ss = Real(0), x = Real(0), y = Real(0);
  sfor<0, NP>([&](auto K) {
    constexpr int k = K;
    x += cc * P.jx[k] - ss * P.jy[k]; y += ss * P.jx[k] + cc * P.jy[k];
    Real sj, cj;
    sincos_<Real>(q[k], sj, cj);
    sj *= P.sigma[k];
    const Real cn = cc * cj - ss * sj, sn = ss * cj + cc * sj;
    cc = cn; ss = sn;
  });
  tx = x + cc * P.tipx - ss * P.tipy; ty = y + ss * P.tipx + cc * P.tipy;
}

// observation (reacher2d.py:40-43): cos q, sin q, target x and z, dq, tip - target
template <class Real, int NP>
__device__ __forceinline__ void arm_write_obs(const ArmParams<Real, NP>& P, const Real (&q)[NP], const Real (&dq)[NP], const Real (&tgt)[3],
                                              float* __restrict__ o) {
  sfor<0, NP>([&](auto K) { constexpr int k = K; Real sn, cs; sincos_<Real>(q[k], sn, cs); o[k] = (float)cs; o[NP + k] = (float)sn; });
  o[2 * NP] = (float)tgt[0]; o[2 * NP + 1] = (float)tgt[2];
  sfor<0, NP>([&](auto K) { constexpr int k = K; o[2 * NP + 2 + k] = (float)dq[k]; });
  Real tx, ty;
  arm_tip<Real, NP>(P, q, tx, ty);
  o[3 * NP + 2] = (float)(tx - tgt[0]); o[3 * NP + 3] = (float)(P.height - tgt[1]); o[3 * NP + 4] = (float)(ty - tgt[2]);
}
template <int NP> __device__ __host__ constexpr int arm_obs_dim() { return 3 * NP + 5; }
// task 0 (round 5): a physics-only card of this shape (envs.DartEnv on a user's .skel, dart_env.py:28-175): torques as given, obs = [q, dq],
// reward 0, never done
template <int NP> __device__ __host__ constexpr int arm_obs_dim_rt(int task) { return task == 0 ? 2 * NP : arm_obs_dim<NP>(); }
template <class Real, int NP>
__device__ __forceinline__ void arm_write_obs_rt(const ArmParams<Real, NP>& P, const Real (&q)[NP], const Real (&dq)[NP], const Real (&tgt)[3], float* __restrict__ o) {
  if (P.task == 0) { sfor<0, NP>([&](auto K) { constexpr int k = K; o[k] = (float)q[k]; o[NP + k] = (float)dq[k]; }); return; }
  arm_write_obs<Real, NP>(P, q, dq, tgt, o);
}

template <class Real, int NP>
__global__ void __launch_bounds__(64) arm_step_kernel(ArmParams<Real, NP> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                       int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                       const float* __restrict__ actions, float* __restrict__ obs,
                                                       float* __restrict__ reward, uint8_t* __restrict__ done,
                                                       uint8_t* __restrict__ truncated, int autoreset, uint64_t seed, uint64_t env_offset) {
  constexpr int N = NP;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = e < n_envs;
  const int64_t ec = valid ? e : n_envs - 1;   // tail lanes shadow the last env so wave votes stay uniform
  Real q[N], dq[N], tau[N], tgt[3];
  sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + ec]; dq[i] = dqs[(int64_t)i * n_envs + ec]; });
  sfor<0, 3>([&](auto I) { constexpr int i = I; tgt[i] = P.tstate[4 * ec + i]; });
  int el_in = elapsed[ec];            // fetched with the state: a load issued in the epilogue would be a bare HBM round trip
  uint32_t ep_in = episode[ec];
  Real a2 = Real(0);
  sfor<0, N>([&](auto K) {
    constexpr int k = K;
    const Real a = (Real)actions[ec * N + k];
    a2 += a * a;                                  // reacher2d.py:34: the control cost takes the action as given
    Real cl = (a > P.act_hi[k]) ? P.act_hi[k] : a;   // comparison clamp (reacher2d.py:18-23): a NaN action stays NaN
    cl = (cl < P.act_lo[k]) ? P.act_lo[k] : cl;
    tau[k] = P.task == 0 ? a : cl * P.act_scale[k];
  });
  DART_PIN_VGPR(el_in); DART_PIN_VGPR(ep_in);   // pinned where the state loads are awaited anyway: the compiler must not sink them
#pragma unroll 1
  for (int f = 0; f < P.frame_skip; ++f) arm_world_step<Real, NP>(P, q, dq, tau);
  Real tx, ty;
  arm_tip<Real, NP>(P, q, tx, ty);
  const Real vx = tx - tgt[0], vy = P.height - tgt[1], vz = ty - tgt[2];
  const Real rew = P.task == 0 ? Real(0) : -sqrt(vx * vx + vy * vy + vz * vz) - a2;      // reacher2d.py:31-35; the task itself never ends an episode
  int el = el_in + 1;
  const bool trunc = (P.max_steps > 0) && (el >= P.max_steps);
  const bool dn = trunc;
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
    arm_write_obs_rt<Real, NP>(P, q, dq, tgt, obs + e * arm_obs_dim_rt<NP>(P.task));
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = trunc ? 1 : 0;
  }
}

template <class Real, int NP>
__global__ void __launch_bounds__(256) arm_reset_kernel(ArmParams<Real, NP> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                         int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                         const uint8_t* __restrict__ mask, const double* __restrict__ qnoise,
                                                         const double* __restrict__ vnoise, float* __restrict__ obs, uint64_t seed,
                                                         uint64_t env_offset, int obs_masked_only) {
  constexpr int N = NP;
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
    arm_write_obs_rt<Real, NP>(P, q, dq, tgt, obs + e * arm_obs_dim_rt<NP>(P.task));
  }
}

// per-env task state (reach targets): masked copy of (N, 4) doubles
template <class Real>
__global__ void arm_task_state_kernel(int64_t n_envs, const uint8_t* __restrict__ mask, const double* __restrict__ values, Real* __restrict__ tstate) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs || (mask && !mask[e])) return;
  for (int k = 0; k < 4; k++) tstate[4 * e + k] = (Real)values[4 * e + k];
}

}  // namespace dartk
