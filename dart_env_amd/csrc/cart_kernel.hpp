// cart_kernel.hpp -- gfx950 device code for the fixed-base planar chains of the reference: a cart on a slider carrying one or
// two pendulum links (cartpole.skel, cartpole_swingup.skel, inverted_double_pendulum.skel).
//
// Replaces, one env per lane, what the reference does per env in Python + DART:
//   DartCartPoleEnv.step / _get_obs              reference gym/envs/dart/cart_pole.py:12-30
//   DartCartPoleSwingUpEnv.step                  reference gym/envs/dart/cartpole_swingup.py:14-33
//   DartDoubleInvertedPendulumEnv.step / _get_obs reference gym/envs/dart/inverted_double_pendulum.py:19-53
//   DartEnv.do_simulation, TimeLimit.step, SyncVectorEnv auto-reset (as planar_kernel.hpp)
//
// Same formulation as planar_kernel.hpp (composite bodies about each link's joint origin, H = M + dt D + dt^2 K, explicit H^-1,
// joint limits as a boxed LCP by block principal pivoting) with the floating root replaced by a slider: generalized coordinates
// (x, theta_1 [, theta_2]).  These 2-3-dof models used to run on the wave-per-env tree kernel with 2-4 of 64 lanes busy; here a
// lane owns an env and a batched step is bound by its 60-odd bytes of HBM traffic per env.
#pragma once
#include "planar_kernel.hpp"

namespace dartk {

template <class Real, int NP>
struct CartParams {
  static constexpr int N = 1 + NP;
  Real dt, g, limit_erp_dt, max_erv, cfm1;
  Real cart_mass;
  Real sigma[NP], mass[NP], cx[NP], cy[NP], izz[NP], jx[NP], jy[NP];   // pendulum link k = 1..NP at index k - 1; joint position in the parent frame
  Real lo[N], hi[N];                                                 // +-inf: no limit on that dof
  Real damp[N], stiff[N], rest[N], q0[N], dq0[N];
  Real sqe[N];      // sqrt(dt damp + dt^2 stiff) per dof (planar_kernel.hpp: implicit_accel)
  int impulse_M;    // card.impulse_inertia (A3): 1 = impulses act on M (DART 6), 0 = on M + dt D + dt^2 K
  Real tipx, tipy;                  // inverted_double_pendulum.py:27-32: the 'weight' body's origin in the last link's frame
  Real act_scale, act_lo, act_hi;   // cart_pole.py:16: tau[0] = a[0] * scale (act_lo / act_hi = -/+inf: no clamp)
  Real aux[8], angle_max, s_max, noise, noise_v;
  int frame_skip, max_steps, task, iters;
};

// one world step: q, dq in/out, tau = generalized forces (the three tasks drive the slider only; a physics-only card every dof)
template <class Real, int NP>
__device__ __forceinline__ void cart_world_step(const CartParams<Real, NP>& P, Real (&q)[1 + NP], Real (&dq)[1 + NP], const Real (&tau)[1 + NP]) {
  constexpr int N = 1 + NP, NL = N;   // link 0 = the cart (translates only)
  Real c[NL], s[NL], px[NL], py[NL], lx[NL], ly[NL], om[NL], apx[NL], apy[NL];
  Real mc[NL], dcx[NL], dcy[NL], Ip[NL], Fx[NL], Fy[NL], Nz[NL];
  c[0] = Real(1); s[0] = Real(0); px[0] = Real(0); py[0] = Real(0); lx[0] = Real(0); ly[0] = Real(0); om[0] = Real(0);
  apx[0] = Real(0); apy[0] = Real(0);
  mc[0] = P.cart_mass; dcx[0] = Real(0); dcy[0] = Real(0); Ip[0] = Real(0); Fx[0] = Real(0); Fy[0] = P.cart_mass * P.g; Nz[0] = Real(0);
  sfor<1, NL>([&](auto K) {
    constexpr int k = K, p = k - 1, i = k - 1;
    Real sj, cj;
    sincos_<Real>(q[k], sj, cj);
    sj *= P.sigma[i];
    c[k] = c[p] * cj - s[p] * sj;
    s[k] = s[p] * cj + c[p] * sj;
    lx[k] = c[p] * P.jx[i] - s[p] * P.jy[i]; ly[k] = s[p] * P.jx[i] + c[p] * P.jy[i];
    px[k] = px[p] + lx[k]; py[k] = py[p] + ly[k];
    om[k] = om[p] + P.sigma[i] * dq[k];
    const Real w2p = om[p] * om[p];
    apx[k] = apx[p] - w2p * lx[k]; apy[k] = apy[p] - w2p * ly[k];
    const Real ox = c[k] * P.cx[i] - s[k] * P.cy[i], oy = s[k] * P.cx[i] + c[k] * P.cy[i];
    const Real w2 = om[k] * om[k];
    const Real fx = P.mass[i] * (apx[k] - w2 * ox), fy = P.mass[i] * (apy[k] - w2 * oy + P.g);
    mc[k] = P.mass[i]; dcx[k] = P.mass[i] * ox; dcy[k] = P.mass[i] * oy;
    Ip[k] = P.izz[i] + P.mass[i] * (ox * ox + oy * oy);
    Fx[k] = fx; Fy[k] = fy; Nz[k] = ox * fy - oy * fx;
  });
  sfor_rev<1, NL>([&](auto K) {
    constexpr int k = K, p = k - 1;
    Ip[p] += Ip[k] + Real(2) * (lx[k] * dcx[k] + ly[k] * dcy[k]) + mc[k] * (lx[k] * lx[k] + ly[k] * ly[k]);
    dcx[p] += dcx[k] + mc[k] * lx[k]; dcy[p] += dcy[k] + mc[k] * ly[k];
    mc[p] += mc[k];
    Nz[p] += Nz[k] + (lx[k] * Fy[k] - ly[k] * Fx[k]);
    Fx[p] += Fx[k]; Fy[p] += Fy[k];
  });
  Real H[N * (N + 1) / 2], rhs[N];
  H[tri(0, 0)] = mc[0];
  rhs[0] = tau[0] - Fx[0];
  sfor<1, NL>([&](auto K) {
    constexpr int k = K, i = k - 1;
    H[tri(k, 0)] = -P.sigma[i] * dcy[k];
    sfor<1, k + 1>([&](auto J) {
      constexpr int j = J;
      H[tri(k, j)] = P.sigma[i] * P.sigma[j - 1] * (Ip[k] + dcx[k] * (px[k] - px[j]) + dcy[k] * (py[k] - py[j]));
    });
    rhs[k] = tau[k] - P.sigma[i] * Nz[k];
  });
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    rhs[i] -= P.damp[i] * dq[i] + P.stiff[i] * (q[i] + P.dt * dq[i] - P.rest[i]);
    if (!P.impulse_M) H[tri(i, i)] += P.dt * P.damp[i] + P.dt * P.dt * P.stiff[i];
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
  // joint limits at q_t (inclusive, DART's JointLimitConstraint): one LCP row per dof
  Real A[N * (N + 1) / 2], b[N], lo[N], hi[N], x[N];
  bool act[N], any = false;
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
  });
  if (__any(any)) {
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      sfor<0, i + 1>([&](auto J) {
        constexpr int j = J;
        A[tri(i, j)] = (i == j) ? (act[i] ? H[tri(i, i)] * P.cfm1 : Real(1)) : ((act[i] && act[j]) ? H[tri(i, j)] : Real(0));
      });
    });
    uint32_t pinmask = 0, F = 0, U = 0;
    Real bmax0 = Real(0);
    sfor<0, N>([&](auto I) { bmax0 = fmax(bmax0, fabs(b[I])); });
    const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      x[i] = Real(0);
      const bool pinned = !(lo[i] < hi[i]);
      const bool upper = !(lo[i] == Real(0));
      const bool start_free = !pinned && (upper ? (b[i] < -tol0) : (b[i] > tol0));
      pinmask |= pinned ? (1u << i) : 0u;
      F |= start_free ? (1u << i) : 0u;
      U |= (upper && !start_free) ? (1u << i) : 0u;
    });
    blcp_bpp<Real, N, true>(A, b, lo, hi, pinmask, F, U, x, P.iters, nullptr);
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real dv = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; dv += H[tri(i, j)] * x[j]; });
      vs[i] += dv;
    });
  }
  sfor<0, N>([&](auto I) { constexpr int i = I; dq[i] = vs[i]; q[i] += P.dt * vs[i]; });
}

// observation: [q, dq] (cart_pole.py:27-28, cartpole_swingup.py:36-37) or [q0, sin q1..2, cos q1..2, dq] (inverted_double_pendulum.py:45-51)
template <class Real, int NP>
__device__ __forceinline__ void cart_write_obs(const CartParams<Real, NP>& P, const Real (&q)[1 + NP], const Real (&dq)[1 + NP], float* __restrict__ o) {
  constexpr int N = 1 + NP;
  if (P.task == 8) {
    o[0] = (float)q[0];
    sfor<1, N>([&](auto K) { constexpr int k = K; Real sn, cs; sincos_<Real>(q[k], sn, cs); o[k] = (float)sn; o[NP + k] = (float)cs; });
    sfor<0, N>([&](auto I) { constexpr int i = I; o[1 + 2 * NP + i] = (float)dq[i]; });
  } else {
    sfor<0, N>([&](auto I) { constexpr int i = I; o[i] = (float)q[i]; o[N + i] = (float)dq[i]; });
  }
}
template <int NP> __device__ __host__ constexpr int cart_obs_dim(int task) { return task == 8 ? 2 + 3 * NP : 2 * (1 + NP); }

template <class Real, int NP>
__global__ void __launch_bounds__(64) cart_step_kernel(CartParams<Real, NP> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                        int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                        const float* __restrict__ actions, float* __restrict__ obs,
                                                        float* __restrict__ reward, uint8_t* __restrict__ done,
                                                        uint8_t* __restrict__ truncated, int autoreset, uint64_t seed, uint64_t env_offset) {
  constexpr int N = 1 + NP;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = e < n_envs;
  const int64_t ec = valid ? e : n_envs - 1;   // tail lanes shadow the last env so wave votes stay uniform
  Real q[N], dq[N];
  sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + ec]; dq[i] = dqs[(int64_t)i * n_envs + ec]; });
  int el_in = elapsed[ec];            // fetched with the state: a load issued in the epilogue would be a bare HBM round trip
  uint32_t ep_in = episode[ec];
  // task 0 (round 5): a physics-only card -- what envs.DartEnv builds from a user's .skel of this shape (dart_env.py:28-175): the action is
  // the generalized force on every dof (no clamp, no scale), the observation [q, dq], reward 0, never done
  const bool phys = P.task == 0;
  const Real a = (Real)actions[phys ? ec * N : ec];
  Real cl = (a > P.act_hi) ? P.act_hi : a;      // comparison clamp; these three tasks leave act_lo / act_hi at -/+inf
  cl = (cl < P.act_lo) ? P.act_lo : cl;
  Real tau[N];
  tau[0] = phys ? a : cl * P.act_scale;
  sfor<1, N>([&](auto K) { constexpr int k = K; tau[k] = phys ? (Real)actions[ec * N + k] : Real(0); });
  DART_PIN_VGPR(el_in); DART_PIN_VGPR(ep_in);   // pinned where the state loads are awaited anyway: the compiler must not sink them
#pragma unroll 1
  for (int f = 0; f < P.frame_skip; ++f) cart_world_step<Real, NP>(P, q, dq, tau);
  bool fin = true, bounded = true;
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    fin = fin && isfinite(q[i]) && isfinite(dq[i]);
    bounded = bounded && (fabs(dq[i]) < P.s_max) && (i < 2 || fabs(q[i]) < P.s_max);
  });
  Real rew = Real(0);
  bool task_done = false;
  if (P.task == 5) {            // cart_pole.py:13-24
    rew = P.aux[0];
    task_done = !(fin && fabs(q[1]) <= P.angle_max);
  } else if (P.task == 7) {     // cartpole_swingup.py:22-31
    rew = P.aux[0] - fabs(q[1]) - P.aux[1] * (a * a) - P.aux[2] * fabs(q[0]);
    task_done = (fabs(q[1]) > P.aux[3]) || (fabs(dq[1]) > P.aux[4]) || (fabs(q[0]) > P.aux[5]);
  } else if (P.task == 8) {     // inverted_double_pendulum.py:27-42: height of the weight above the cart
    Real cc = Real(1), ss = Real(0), y = Real(0);
    sfor<1, N>([&](auto K) {
      constexpr int k = K, i = k - 1;
      y += ss * P.jx[i] + cc * P.jy[i];
      Real sj, cj;
      sincos_<Real>(q[k], sj, cj);
      sj *= P.sigma[i];
      const Real cn = cc * cj - ss * sj, sn = ss * cj + cc * sj;
      cc = cn; ss = sn;
    });
    y += ss * P.tipx + cc * P.tipy;
    const Real height = Real(2) * (y - P.aux[4]) / P.aux[5];
    const Real dist_pen = P.aux[1] * (q[0] * q[0]) + (height - Real(2)) * (height - Real(2));
    Real vel_pen = P.aux[2] * (dq[1] * dq[1]);
    if constexpr (NP >= 2) vel_pen += P.aux[3] * (dq[2] * dq[2]);
    rew = P.aux[0] - dist_pen - vel_pen;
    task_done = height <= Real(1);
  }
  (void)bounded;
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
    cart_write_obs<Real, NP>(P, q, dq, obs + e * cart_obs_dim<NP>(P.task));
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = (trunc && !task_done) ? 1 : 0;
  }
}

template <class Real, int NP>
__global__ void __launch_bounds__(256) cart_reset_kernel(CartParams<Real, NP> P, int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs,
                                                          int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                          const uint8_t* __restrict__ mask, const double* __restrict__ qnoise,
                                                          const double* __restrict__ vnoise, float* __restrict__ obs, uint64_t seed,
                                                          uint64_t env_offset, int obs_masked_only) {
  constexpr int N = 1 + NP;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  Real q[N], dq[N];
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
  if (obs && (m || !obs_masked_only)) cart_write_obs<Real, NP>(P, q, dq, obs + e * cart_obs_dim<NP>(P.task));
}

}  // namespace dartk
