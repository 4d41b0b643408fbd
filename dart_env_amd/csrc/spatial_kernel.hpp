// spatial_kernel.hpp -- gfx950 device code for general (3-D, branching) skeletons: the tree kernel behind DartHumanWalker-v1,
// DartWalker3d-v1, DartDog-v1 and every model the planar register kernels do not take.
//
// Replaces for one env per WAVEFRONT what the reference does per env through pydart2/DART
// (reference gym/envs/dart/human_walker.py:60-165, dart_env.py:158-175).
//
// Design: the 21/29-dof models do not fit one lane's registers (H is 29x29, the contact / limit LCP
// has up to 36 rows, 64 with link-link contacts), so one 64-lane wavefront owns one environment and the per-skeleton block
// lives in LDS (18.9 KB fp32 for HumanWalker): link records, H and its Cholesky factor, the constraint Jacobian
// (overwritten by W = L^-1 J^T), the Delassus matrix A = W W^T and the pivoting solver's LDL^T workspace.  Tree recursions
// are log-depth (pointer jumping over __shfl for the forward pass, group levels for the backward pass), everything dense is
// spread over the lanes: one lane per dof for the mass-matrix rows, one lane per constraint row for Jacobians / triangular
// solves, round-robin entries for A, a systolic register Cholesky, and the LCP active-set logic is wave-uniform (row
// infeasibility flags are gathered with __ballot).  Strides of the LDS matrices are odd: row-per-lane access is bank-conflict free.
//
// Dynamics formulation: world-aligned recursive Newton-Euler + composite bodies taken about each joint origin
// (coordinates relative to the floating base translation so fp32 does not see the travelled distance) -- a different
// derivation from the oracle's body-frame spatial algebra.
//
// Files: spatial_model.hpp (constants, SpatialModel, LDS block) . spatial_dynamics.hpp (tree recursions) .
// spatial_dense.hpp (Cholesky, solves, boxed LCP) . spatial_free_root.hpp (FreeJoint) . spatial_box_box.hpp (link-link
// contacts) . spatial_world_step.hpp (one DART step) . spatial_tasks.hpp (task epilogues) . this file (the kernels).
#pragma once
#include <type_traits>
#include "spatial_tasks.hpp"

// the workgroup's dynamic LDS block (the host emulation of tests/kernel_emu/fake_wave_include supplies its own definition)
#ifndef DART_DYNAMIC_LDS
#define DART_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

namespace dartk {

// ------------------------------------------------------------------ kernels: one wavefront (64 threads) per env
// fp64 doubles the LDS block (HumanWalker: 4 workgroups per CU = one wave per SIMD): its register budget is the whole file.
// BIG kernels hold the LCP rows in registers: two waves per SIMD is what their LDS block allows anyway (HumanWalker fp32: 8
// workgroups per CU); the small-model kernels keep three.
// Waves per SIMD the register allocation is bounded for.  Small models: 3 (168 VGPRs).  BIG (register LCP solver): 2 in fp32 / 1 in
// fp64 -- except the lean kernel of a model with a compile-time factor pattern (HumanWalker), which is short enough for 3 in fp32
// (measured 6.19 -> 5.84 ms; the PAIRS kernel of Walker3d got 13 % slower at 3) and 2 in fp64 (below).
#ifndef SP_PAT_F32_WAVES
#define SP_PAT_F32_WAVES 3
#endif
#ifndef SP_BAKE_DIMS
#define SP_BAKE_DIMS 1
#endif
// fp64 pattern kernel: 2 (round 4).  Its LDS block (26 928 B after the trim in sp_carve) fits six times into a CU, so two of the four
// SIMDs can hold a second wave -- if a wave stays within 256 registers.  Measured (HumanWalker, 16 384 envs): 9.72 ms at 1 (256 VGPR + 96
// AGPR, 356 B scratch) -> 9.01 ms at 2 (256 VGPR, 816 B scratch).  The callees must be private copies for this to work (sp_blcp_t's TAG).
// fp64 kernels of the small models: 2 (round 4).  Their LDS block allows fewer workgroups per CU than three waves per SIMD would need
// anyway (the Dog: 24.7 KB = 6 per CU), so the 168-register budget only bought spills: Dog 3.65 -> 2.80 ms at 256 registers.
#ifndef SP_SMALL_F64_WAVES
#define SP_SMALL_F64_WAVES 2
#endif
#ifndef SP_PAT_F64_WAVES
#define SP_PAT_F64_WAVES 2
#endif
// The lane index is made opaque to the compiler at the top of every world step (round 5).  With `lane` a loop invariant, every per-lane
// LDS address, mask and comparison of the world step is hoisted out of the frame loop -- a hundred-odd values that then live across the
// whole loop and every call in it; the allocator spilled ~30 register pairs before the loop and reloaded 18 of them after each call to the
// LCP solver (found in the disassembly of the fp64 pattern kernel; the pre-loop stores were 80 % of its 307 MB of HBM writes per launch).
// Recomputing them per world step is a few dozen integer instructions.  Compiler remarks, VGPR spills of the eight fp32 / eight fp64
// instantiations: 162 118 80 94 137 81 89 77 -> 47 24 13 12 40 16 5 14 and 113 0 0 73 82 63 0 48 -> 21 0 0 7 13 80 0 5 (the fp64 pattern
// kernel also stores its factor as a skyline now); measured: profiles/r05_tree_kernel_ab.txt.  0 restores rounds 1-4.
#ifndef SP_OPAQUE_LANE
#define SP_OPAQUE_LANE 1
#endif
template <class Real, bool BIG, class PAT = DensePattern> __host__ __device__ constexpr int sp_min_waves() {
  if (!BIG) return sizeof(Real) == 8 ? SP_SMALL_F64_WAVES : 3;
  if (!PAT::dense) return sizeof(Real) == 8 ? SP_PAT_F64_WAVES : SP_PAT_F32_WAVES;
  return sizeof(Real) == 8 ? 1 : 2;
}
// REPORT: the contact-report variant (dart_get_contacts); only the most general instantiation <true, true, true> is built --
// the mere presence of the reporting code costs the lean kernels 2.5 % (register allocation), so they do not carry it.
template <class Real, bool PAIRS, bool EXTRAS, bool REPORT = false, bool BIG = false, class PAT = DensePattern>
__global__ void __launch_bounds__(64, (sp_min_waves<Real, BIG, PAT>())) sp_step_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                      Real* __restrict__ qs, Real* __restrict__ dqs, Real* __restrict__ tstate,
                                                      int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                      const float* __restrict__ actions, float* __restrict__ obs,
                                                      float* __restrict__ reward, uint8_t* __restrict__ done,
                                                      uint8_t* __restrict__ truncated, int autoreset, uint64_t seed,
                                                      uint64_t env_offset) {
  DART_DYNAMIC_LDS(sp_smem);
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  if ((int64_t)blockIdx.x >= n_envs) return;
  const int64_t e = Md.sched_perm ? (int64_t)Md.sched_perm[blockIdx.x] : (int64_t)blockIdx.x;
  const long long sched_t0 = Md.sched_cost ? (long long)__builtin_readcyclecounter() : 0ll;
  // BK: the pattern kernel takes the model's dimensions and task from its pattern at compile time (tree_patterns.hpp; launched only
  // for a model that carries exactly these values, SpatialImplT::matches_pattern; its launch condition fixes reg_lcp = 1)
  constexpr bool BK = (SP_BAKE_DIMS != 0) && !PAT::dense;
  const int n = BK ? PAT::n : Md.n, nl_ = BK ? PAT::nl : Md.nl;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, nl_, n, BK ? PAT::maxm : Md.maxm, BK ? PAT::maxcp : Md.maxcp, BK ? PAT::reg_lcp : Md.reg_lcp,
                                 BK ? PAT::hreals : Md.hreals);   // (a pattern kernel stores H as a skyline; the host states the same size in Md.hreals)
  const int task = BK ? PAT::task : Md.task, act_dim = BK ? PAT::act_dim : Md.act_dim, act_dof0 = BK ? PAT::act_dof0 : Md.act_dof0;
  const int frame_skip = BK ? PAT::frame_skip : Md.frame_skip, obs_dim = BK ? PAT::obs_dim : Md.obs_dim;
  int* cflags = S.imisc + 2;
  Real* sh_scal = S.misc + 8;
  if (lane < n) { S.q[lane] = qs[e * n + lane]; S.dq[lane] = dqs[e * n + lane]; S.tau[lane] = Real(0); }
  if (lane < nl_) { S.topo[lane] = (Md.parent[lane] + 1) | ((Md.dof[lane] + 1) << 8) | (Md.jtype[lane] << 16); S.ancd[lane] = (int)Md.anc_dofs[lane]; }
  if (Md.stats && lane < 10) S.ticks[lane] = 0ull;
  if (EXTRAS && task == 12 && lane < n) S.cf[lane] = Md.cf_store[e * n + lane];
  __syncthreads();
  // action: comparison clamp (a NaN stays NaN, hopper.py:25-30) and scaling, one lane per action; the reward's |a| and a^2 sums
  // by wave reductions
  Real abs_sum = Real(0), sq_sum = Real(0);
  {
    const bool has = lane < act_dim;
    const Real a = has ? (Real)actions[e * act_dim + lane] : Real(0);
    abs_sum = wave_sum<Real>(fabs(a)); sq_sum = wave_sum<Real>(a * a);
    if (has) {
      Real cl = (a > Md.act_hi[lane]) ? Md.act_hi[lane] : a;
      cl = (cl < Md.act_lo[lane]) ? Md.act_lo[lane] : cl;
      const int dd = act_dof0 + lane;
      if (EXTRAS && task == 12)   // SPD: the action is a target pose inside the joint's limits (walker3d_spd.py:68-70)
        S.tau[dd] = (cl + Real(1)) / Real(2) * (Md.upper[dd] - Md.lower[dd]) + Md.lower[dd];
      else
        S.tau[dd] = cl * Md.act_scale[lane];
    }
  }
  LinkConst<Real> lc;
  sp_load_link_const<Real>(Md, lane < nl_ ? lane : 0, lc);
  if constexpr (BK) sp_load_pattern_const<PAT, Real>(lane, lc);
  if (EXTRAS && Md.free_root) {
    __syncthreads();
    if (lane == 0) { sp_free_root_load<Real>(S); sp_free_root_to_internal<Real>(S); }   // S.q / S.dq: internal from here
  }
  // link poses are needed before the step only by the tasks that measure progress on a body (3, 4) or a tip (11), after
  // it only by the tasks whose reward / done / observation read a body pose
  if (task == 3 || task == 4 || task == 11 || task == 12 || task == 13) sp_pose_pass<Real, EXTRAS, typename std::conditional<BK, PAT, DensePattern>::type>(lc, Md, S, lane);
  else __syncthreads();
  if (lane == 0) {
    sh_scal[0] = (task == 3 || task == 4 || task == 12 || task == 13) ? S.link[Md.aux_link[0] * SP_LINKF + LK_C] + S.misc[0] : S.q[0];   // posbefore
    if (task == 11) {   // DartReacher3d: distance to the target BEFORE the step, and sum tau^2 (reacher.py:23-27)
      const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - ld3(tstate + 4 * e);
      sh_scal[0] = sqrt(dot(vec, vec));
      Real st2 = Real(0);
      for (int k = 0; k < act_dim; k++) st2 += S.tau[act_dof0 + k] * S.tau[act_dof0 + k];
      sh_scal[2] = st2;
    }
    cflags[0] = 0; cflags[1] = 0;
  }
  __syncthreads();
#pragma nounroll
  for (int f = 0; f < frame_skip; ++f) {
#if SP_OPAQUE_LANE && defined(__HIP_DEVICE_COMPILE__)
    int ln = lane;
    DART_OPAQUE(ln);
    __builtin_assume(ln >= 0 && ln < 64);
    sp_world_step<Real, PAIRS, EXTRAS, REPORT, BIG, PAT>(Md, lc, S, ln, cflags, e, REPORT && Md.creport != nullptr && f == frame_skip - 1);
#else
    sp_world_step<Real, PAIRS, EXTRAS, REPORT, BIG, PAT>(Md, lc, S, lane, cflags, e, REPORT && Md.creport != nullptr && f == frame_skip - 1);
#endif
  }
  if (Md.stats && lane < 10) atomicAdd(&Md.stats[40 + lane], S.ticks[lane]);
  bool dn = false, tr = false;
  const bool pose_last = task == 1 || task == 2 || task == 3 || task == 4 || task == 8 || task >= 10;
  const bool pose_reset = pose_last && task != 13;   // the dog's observation holds no body pose
  if (EXTRAS && Md.free_root) {
    __syncthreads();
    if (lane == 0) sp_free_root_to_internal<Real>(S);   // internal coordinates of the final pose
  }
  if (pose_last) sp_pose_pass<Real, EXTRAS, typename std::conditional<BK, PAT, DensePattern>::type>(lc, Md, S, lane);
  if (lane == 0) {
    Real rew = Real(0);
    bool task_done = false;
    Real dog_x = Real(0), dog_h = Real(0), dog_side = Real(0);
    if (task == 13) {
      const Real* Lb = S.link + Md.aux_link[0] * SP_LINKF;
      dog_x = Lb[LK_C] + S.misc[0]; dog_h = Lb[LK_C + 1] + S.misc[1]; dog_side = fabs(Lb[LK_C + 2] + S.misc[2]);
    }
    if (EXTRAS && Md.free_root) sp_free_root_store<Real>(S);          // S.q / S.dq: public coordinates again
    if (task == 13) {   // DartDogEnv.step (dog.py:28-46)
      rew = Md.aux_real[1] * (dog_x - sh_scal[0]) * Md.inv_envdt;
      rew += Md.aux_real[0];
      rew -= Md.aux_real[2] * sq_sum;
      bool ok = true;
      for (int i = 0; i < n; i++) {
        ok = ok && isfinite(S.q[i]) && isfinite(S.dq[i]) && (fabs(S.dq[i]) < Md.s_max);
        if (i >= 2) ok = ok && (fabs(S.q[i]) < Md.s_max);
      }
      task_done = !(ok && dog_h > Md.aux_real[4] && dog_h < Md.aux_real[5] && dog_side < Md.aux_real[3]);
    }
    if (task == 4) task_done = sp_humanwalker_epilogue<Real>(Md, S, sh_scal[0], abs_sum, tstate[4 * e], cflags, rew);
    else if (task == 3 || task == 12) task_done = sp_walker3d_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    else if (task >= 5 && task != 13) task_done = sp_simple_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    else if (task == 1 || task == 2) task_done = sp_planar_task_epilogue<Real>(Md, S, sh_scal[0], sq_sum, rew);
    if (task == 10 || task == 11) {   // reachers: reward from the tip-target distance (2-D: after, 3-D: before the step)
      const V3<Real> tgt = ld3(tstate + 4 * e);
      bool fin = true;
      for (int i = 0; i < n; i++) fin = fin && isfinite(S.q[i]) && isfinite(S.dq[i]);
      if (task == 11) {
        const Real dist0 = sh_scal[0];
        rew = -dist0 + -(sh_scal[2] * Md.aux_real[3]);
        task_done = !(fin && (dist0 > Md.aux_real[4]));
      } else {
        const V3<Real> vec = sp_reacher_tip<Real>(Md, S) - tgt;
        rew = -sqrt(dot(vec, vec)) + -sq_sum;
        task_done = false;
      }
    }
    int el = elapsed[e] + 1;
    const bool trunc = (Md.max_steps > 0) && (el >= Md.max_steps);
    dn = task_done || trunc; tr = trunc && !task_done;
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = tr ? 1 : 0;
    elapsed[e] = (autoreset && dn) ? 0 : el;
    sh_scal[1] = dn ? Real(1) : Real(0);
  }
  __syncthreads();
  dn = sh_scal[1] != Real(0);
  if (autoreset && dn) {
    const uint32_t ep = episode[e] + 1;
    // Philox reset noise, same stream as the planar kernels / the oracle: u[0..n) positions, u[n..2n) velocities
    if (lane < n) {
      const int iq = lane, iv = n + lane;
      uint32_t o[4];
      philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iq / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
      const Real uq = Real(o[iq % 4] >> 8) * Real(1.0 / 16777216.0);
      philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iv / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
      const Real uv = Real(o[iv % 4] >> 8) * Real(1.0 / 16777216.0);
      S.q[lane] = Md.q0[lane] + (-Md.noise + Real(2) * Md.noise * uq);
      S.dq[lane] = Md.dq0[lane] + (-Md.noise_v + Real(2) * Md.noise_v * uv);
    }
    __syncthreads();
    if (pose_reset) sp_pose_pass<Real, false, typename std::conditional<BK, PAT, DensePattern>::type>(lc, Md, S, lane);   // (`dn` is wave-uniform; no free-root model reads a pose here)
    if (lane == 0) {
      episode[e] = ep;
      if (task == 4) tstate[4 * e] = S.link[Md.aux_link[1] * SP_LINKF + LK_C + 1] + S.misc[1];
      cflags[0] = 0; cflags[1] = 0;
    }
    __syncthreads();
  }
  if (lane < n) { qs[e * n + lane] = S.q[lane]; dqs[e * n + lane] = S.dq[lane]; }
  if (EXTRAS && task == 12 && lane < n) Md.cf_store[e * n + lane] = (autoreset && dn) ? Real(0) : S.cf[lane];
  if ((task == 10 || task == 11) && lane < 3) S.misc[4 + lane] = tstate[4 * e + lane];
  __syncthreads();
  sp_write_obs<Real>(Md, S, cflags, obs + e * obs_dim, lane, BK ? n : -1, BK ? task : -1);
  if (Md.sched_cost && lane == 0) Md.sched_cost[e] = (unsigned int)((long long)__builtin_readcyclecounter() - sched_t0);
}

// Dynamics quantities of the CURRENT state (pydart2's skel.M and skel.c, reference gym/envs/dart/walker3d_spd.py:40-55):
// mass matrix (n x n, symmetric, without the implicit damping / stiffness terms) and Coriolis + gravity forces.
// One wavefront per env; `soa` tells how the owning implementation stores its state (planar kernels: q[n][N]).
template <class Real>
__global__ void __launch_bounds__(64) sp_dynamics_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                          const Real* __restrict__ qs, const Real* __restrict__ dqs, int soa,
                                                          double* __restrict__ mass_out, double* __restrict__ bias_out,
                                                          double* __restrict__ pose_out, int nbodies) {
  DART_DYNAMIC_LDS(sp_smem);
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  if (e >= n_envs) return;
  const int n = Md.n, nl = Md.nl;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, nl, n, Md.maxm, Md.maxcp, Md.reg_lcp, Md.hreals);
  if (lane < n) {
    const int64_t at = soa ? (int64_t)lane * n_envs + e : e * n + lane;
    S.q[lane] = qs[at]; S.dq[lane] = dqs[at]; S.tau[lane] = Real(0);
  }
  if (lane < nl) { S.topo[lane] = (Md.parent[lane] + 1) | ((Md.dof[lane] + 1) << 8) | (Md.jtype[lane] << 16); S.ancd[lane] = (int)Md.anc_dofs[lane]; }
  __syncthreads();
  if (pose_out) {   // bodynode world transforms / COMs: R (9), origin (3), com (3) per card body; cold path, serial kinematics
    if (lane == 0) {
      if (Md.free_root) { sp_free_root_load<Real>(S); sp_free_root_to_internal<Real>(S); }
      sp_kinematics<Real>(Md, S);
    }
    __syncthreads();
    if (lane < nl && Md.link_body[lane] >= 0) {
      const Real* L = S.link + lane * SP_LINKF;
      double* o = pose_out + ((size_t)e * nbodies + Md.link_body[lane]) * 15;
      // the root translation joints are factored out of the link records (S.misc): a carrier body of that chain has only
      // the slides up to its own joint behind it, every other body all of them
      V3<Real> off = v3<Real>(0, 0, 0);
      for (int j = 0; j < nl; j++)
        if (Md.root_trans[j] && (j <= lane || !Md.root_trans[lane])) off = off + ld3(S.link + j * SP_LINKF + LK_A) * S.q[Md.dof[j]];
      const Real offv[3] = {off.x, off.y, off.z};
      for (int k = 0; k < 9; k++) o[k] = (double)L[LK_R + k];
      for (int k = 0; k < 3; k++) { o[9 + k] = (double)(L[LK_P + k] + offv[k]); o[12 + k] = (double)(L[LK_C + k] + offv[k]); }
    }
    if (!mass_out && !bias_out) return;
    __syncthreads();
    if (lane < n) { const int64_t at = soa ? (int64_t)lane * n_envs + e : e * n + lane; S.q[lane] = qs[at]; S.dq[lane] = dqs[at]; }
    __syncthreads();
  }
  LinkConst<Real> lc;
  sp_load_link_const<Real>(Md, lane < nl ? lane : 0, lc);
  if (Md.free_root) {   // FreeJoint root: the chain's internal coordinates, chart re-centred on the pose (the host maps M, c to DART's)
    if (lane == 0) { sp_free_root_load<Real>(S); sp_free_root_to_internal<Real>(S); }
    __syncthreads();
  }
  if (lane == 0) sp_root_offset<Real>(Md, S);
  if (Md.free_root) sp_forward<Real, true>(lc, Md, S, lane); else sp_forward<Real>(lc, Md, S, lane);
  __syncthreads();
  for (int lv = Md.n_group_levels - 1; lv >= 0; lv--) {
    if (lane < nl && lc.group_level == lv) sp_gather_children<Real>(lc, Md, S, lane);
    __syncthreads();
  }
  if (lane < nl) sp_link_rhs<Real>(lc, Md, S, lane);
  __syncthreads();
  if (bias_out && lane < nl && lc.dof >= 0) {
    const Real* L = S.link + lane * SP_LINKF;
    const V3<Real> a = ld3(L + LK_A);
    const Real* D = S.ldyn + lane * SP_LDYN;
    bias_out[e * n + lc.dof] = (double)((lc.jtype == 2) ? dot(a, ld3(D + LD_N)) : dot(a, ld3(D + LD_F)));
  }
  if (mass_out) {
    if (lane < n) for (int k = 0; k <= lane; k++) S.H[HL(lane, k)] = Real(0);
    __syncthreads();
    if (lane < n) sp_mass_row<Real>(lc, Md, S, lane, false);
    __syncthreads();
    if (lane < n) {
      double* Mo = mass_out + e * n * n;
      for (int k = 0; k <= lane; k++) {
        const double v = (double)S.H[HI(n - 1 - lane, n - 1 - k)];   // reversed storage order (sp_mass_row)
        Mo[lane * n + k] = v; Mo[k * n + lane] = v;
      }
    }
  }
}

// per-env task state (reach targets): masked copy of (N, 4) doubles
template <class Real>
__global__ void sp_task_state_kernel(int64_t n_envs, const uint8_t* __restrict__ mask, const double* __restrict__ values, Real* __restrict__ tstate) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs || (mask && !mask[e])) return;
  for (int k = 0; k < 4; k++) tstate[4 * e + k] = (Real)values[4 * e + k];
}

// masked reset: q = init + noise (host rows or Philox), elapsed = 0, init height, obs
template <class Real>
__global__ void __launch_bounds__(64) sp_reset_kernel(const SpatialModel<Real>* __restrict__ Mp, int64_t n_envs,
                                                       Real* __restrict__ qs, Real* __restrict__ dqs, Real* __restrict__ tstate,
                                                       int32_t* __restrict__ elapsed, uint32_t* __restrict__ episode,
                                                       const uint8_t* __restrict__ mask, const double* __restrict__ qnoise,
                                                       const double* __restrict__ vnoise, float* __restrict__ obs,
                                                       uint64_t seed, uint64_t env_offset, int obs_masked_only) {
  DART_DYNAMIC_LDS(sp_smem);
  const SpatialModel<Real>& Md = *Mp;
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  if (e >= n_envs) return;
  const int n = Md.n;
  SpLds<Real> S = sp_carve<Real>((Real*)sp_smem, Md.nl, n, Md.maxm, Md.maxcp, Md.reg_lcp, Md.hreals);
  int* cflags = S.imisc + 2;
  const bool m = (mask == nullptr) || mask[e];
  if (lane < n) {
    if (m) {
      if (qnoise) { S.q[lane] = (Real)qnoise[e * n + lane]; S.dq[lane] = (Real)vnoise[e * n + lane]; }
      else {
        const uint32_t ep = episode[e] + 1;
        const int iq = lane, iv = n + lane;
        uint32_t o[4];
        philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iq / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const Real uq = Real(o[iq % 4] >> 8) * Real(1.0 / 16777216.0);
        philox4x32_10((uint32_t)(env_offset + e), (uint32_t)((env_offset + e) >> 32), ep, (uint32_t)(iv / 4), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const Real uv = Real(o[iv % 4] >> 8) * Real(1.0 / 16777216.0);
        S.q[lane] = Md.q0[lane] + (-Md.noise + Real(2) * Md.noise * uq);
        S.dq[lane] = Md.dq0[lane] + (-Md.noise_v + Real(2) * Md.noise_v * uv);
      }
    } else { S.q[lane] = qs[e * n + lane]; S.dq[lane] = dqs[e * n + lane]; }
  }
  __syncthreads();
  if (lane == 0) {
    cflags[0] = 0; cflags[1] = 0;
    if (m) {
      if (!qnoise) episode[e] = episode[e] + 1;
      elapsed[e] = 0;
    }
    if (m || (obs && !obs_masked_only && (Md.task == 1 || Md.task == 2 || Md.task == 10 || Md.task == 11))) {   // these observations need the pose
      sp_kinematics<Real>(Md, S);
      if (Md.task == 4) tstate[4 * e] = S.link[Md.aux_link[1] * SP_LINKF + LK_C + 1] + S.misc[1];
    }
  }
  __syncthreads();
  if (m && lane < n) { qs[e * n + lane] = S.q[lane]; dqs[e * n + lane] = S.dq[lane]; }
  if (m && Md.task == 12 && Md.cf_store && lane < n) Md.cf_store[e * n + lane] = Real(0);   // world.reset() clears them
  if ((Md.task == 10 || Md.task == 11) && lane < 3) S.misc[4 + lane] = tstate[4 * e + lane];
  __syncthreads();
  if (obs && (m || !obs_masked_only)) sp_write_obs<Real>(Md, S, cflags, obs + e * Md.obs_dim, lane);
}

// Dispatch order of the next step (SpatialModel::sched_perm, DART_CFG_LAUNCH_ORDER): env indices by descending duration of the step
// that just ran.  A launch ends when its last workgroup does and workgroups are dispatched in blockIdx order, so the expensive envs go
// first (longest-processing-time-first packing); 1 024 buckets of duration relative to the slowest env are all that needs -- the order
// inside a bucket is whatever the atomics make it.  One workgroup; ~10 us at 16 384 envs behind a step kernel of milliseconds.
template <class Real>   // (a template only so that the two precision units may both hold it)
__global__ void __launch_bounds__(1024) sp_sched_kernel(int64_t n, const unsigned int* __restrict__ cost, int* __restrict__ perm) {
  __shared__ unsigned int hist[1024], scan[1024], cmax;
  const int t = threadIdx.x;
  hist[t] = 0u;
  if (t == 0) cmax = 1u;
  __syncthreads();
  unsigned int m = 0u;
  for (int64_t i = t; i < n; i += 1024) m = cost[i] > m ? cost[i] : m;
  atomicMax(&cmax, m);
  __syncthreads();
  const unsigned long long cm = cmax;
  auto bucket = [&](unsigned int c) { return 1023u - (unsigned int)(((unsigned long long)c * 1023ull) / cm); };
  for (int64_t i = t; i < n; i += 1024) atomicAdd(&hist[bucket(cost[i])], 1u);
  __syncthreads();
  scan[t] = hist[t];
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // inclusive prefix sum over the buckets
    const unsigned int v = t >= d ? scan[t - d] : 0u;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  hist[t] = scan[t] - hist[t];           // first slot of bucket t
  __syncthreads();
  for (int64_t i = t; i < n; i += 1024) perm[atomicAdd(&hist[bucket(cost[i])], 1u)] = (int)i;
}

// (N, n) doubles <-> the kernel's AoS state
template <class Real>
__global__ void sp_state_io_kernel(int64_t count, Real* __restrict__ qs, Real* __restrict__ dqs, double* __restrict__ qh,
                                   double* __restrict__ dqh, int to_device) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (to_device) { qs[i] = (Real)qh[i]; dqs[i] = (Real)dqh[i]; }
  else { qh[i] = (double)qs[i]; dqh[i] = (double)dqs[i]; }
}

}  // namespace dartk
