// planar_kernel.hpp -- gfx950 device code of the batched planar articulated-body stepper.
//
// Replaces, for one env per lane, what the reference does per env in Python + DART:
//   DartEnv.do_simulation      reference gym/envs/dart/dart_env.py:158-175
//   DartHopperEnv.advance/step reference gym/envs/dart/hopper.py:24-74
//   DartWalker2dEnv.step       reference gym/envs/dart/walker2d.py:22-74
//   TimeLimit.step             reference gym/wrappers/time_limit.py:14-21
//   SyncVectorEnv auto-reset   reference gym/vector/sync_vector_env.py:73-84
//
// Design (MI355X-first): one env per lane, the whole skeleton block -- H (packed), H^-1,
// constraint Jacobians, Delassus matrix A -- lives in VGPRs (the models of configs 1-3,5 fit;
// LDS is left for the spatial kernel of the 21/29-dof models), state is struct-of-arrays
// `q[dof][env]` so every global access is a fully coalesced 256 B wave transaction, all
// `frame_skip` substeps + reward/done/obs + auto-reset are fused into one launch, and the
// constraint solver votes across the 64-lane wavefront (`__all`) to leave its loop early.
// Topology is a compile-time trait (loops unroll to straight-line register code); every
// numeric model parameter is a runtime kernel argument held in SGPRs.
//
// Dynamics formulation: planar composite-rigid-body algorithm in world-aligned, root-relative
// coordinates (closed forms below) -- a different derivation from the oracle's 6-D spatial
// algebra, so the two cross-check each other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "mt19937_draw.hpp"

// a branch the workloads practically never take: block frequencies steer the register allocator's spill placement -- the cold tier's live
// ranges are split around IT, not around the hot one.  Used for ONE branch, the all-limits tier of the limit-slot vote (Walker2d fp64 89.9 ->
// 84.1 us).  On the contact-tier vote and the Hopper's limit-prefix vote as well it bought nothing -- and the fp32 kernel of the physics-only hopper
// chain (whose four-capsule tier IS the common case for a fallen model) came out WRONG on the device (q off by 0.13 after one env-step,
// tests/test_generic_dartenv.py::test_fallen_user_model_stays_in_the_register_tiers[32-pogo]; exec-prologue lint clean): one more codegen hazard
// of this toolchain that only the parity tests see.  Those two branches are back to what was validated.
#ifndef DART_UNLIKELY
#define DART_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef DART_PIN_VGPR
#define DART_PIN_VGPR(x) asm volatile("" : "+v"(x))   // an optimisation barrier on a value that lives in a VGPR
#endif

namespace dartk {

// ------------------------------------------------------------------ compile-time loops
template <int B, int E, class F>
__device__ __host__ __forceinline__ void sfor(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sfor<B + 1, E>(f);
  }
}
template <int B, int E, class F>
__device__ __host__ __forceinline__ void sfor_rev(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, E - 1>{});
    sfor_rev<B, E - 1>(f);
  }
}
template <int N>
#ifdef DART_ROOT_FIRST
__device__ __host__ constexpr int rev(int i) { return i; }
#else
__device__ __host__ constexpr int rev(int i) { return N - 1 - i; }
#endif
__device__ __host__ constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
__device__ __host__ constexpr int TI(int i, int j) { return tri(i, j); }   // the tree kernel's name for it

// ------------------------------------------------------------------ topologies
// Link 0 is the floating root (dofs 0,1,2 = x, y, rot); link k>=1 hangs on a revolute joint (dof 2+k).
// NC = CANDIDATE capsules (each tested against the floor every substep).  The constraint phase works on compacted contact
// SLOTS: TIER0 slots in the first register path, TIER1 (0 = none) in a second one chosen by a wave vote when some lane has
// more contacts, and a lane with more contacts than the last tier is served alone by a loop-based solver in LDS
// (slow_constraints) -- exact for any number of contacts, and never on the path of a workload that stays within the tiers.
struct HopperTopo {  // reference assets/hopper_capsule.skel: pelvis - thigh - shin - foot; ONLY the foot collides (BASELINE config[1])
  static constexpr int NL = 4, NDOF = NL + 2, NC = 1, NA = 3, TIER0 = 1, TIER1 = 0, TIER1_F64 = 0;
  static constexpr bool WARM = false;  // warm-started active sets: measured -4 % here (short, violent episodes)
  __device__ __host__ static constexpr int parent(int k) { constexpr int P[NL] = {-1, 0, 1, 2}; return P[k]; }
  __device__ __host__ static constexpr int clink(int c) { constexpr int L[NC] = {3}; return L[c]; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};
struct HopperAllTopo {  // the same chain with EVERY capsule tested against the floor (DART's behaviour, the default card)
  static constexpr int NL = 4, NDOF = NL + 2, NC = 4, NA = 3, TIER0 = 1, TIER1 = 2, TIER1_F64 = 2;
#ifndef DART_HOPPER_LIMIT_SLOTS
#define DART_HOPPER_LIMIT_SLOTS 3
#endif
  // see topo_limit_slots.  3 = one row per limited joint.  With 2 compacted rows (a wave's maximum in 99.5 % of its substeps) the LCPs shrink
  // from 4 / 5 to 3 / 4 rows -- and the kernel got SLOWER on the device: fp64 32.0 -> 33.8 us, fp32 27.5 -> 49.9 us (profiles/r06_limit_slots.txt):
  // at these sizes the selects that build the compacted rows and the second inlined tier cost more than the smaller factorisations give back.
  static constexpr int LIMIT_SLOTS = DART_HOPPER_LIMIT_SLOTS;
#ifndef DART_HOPPER_LIMIT_PREFIX
#define DART_HOPPER_LIMIT_PREFIX 2
#endif
  static constexpr int LIMIT_PREFIX = DART_HOPPER_LIMIT_PREFIX;   // see topo_limit_prefix (3 = all limit rows always, rounds 1-5)
  static constexpr bool WARM = false;   // (round 5, host build over 64-lane groups: 13.75 pivoting solves per wave and env-step, 14.02 with warm starts)
  // (round 4 A/B: H^-1 parked in LDS across the pivoting loops, the walker's HINV_LDS_F64, makes THIS kernel slower -- 31.82 -> 33.24 us
  // fp64, 158 -> 126 AGPRs: the 21 entries cost more as LDS round trips than as accumulator-register moves)
  __device__ __host__ static constexpr int parent(int k) { constexpr int P[NL] = {-1, 0, 1, 2}; return P[k]; }
  __device__ __host__ static constexpr int clink(int c) { constexpr int L[NC] = {0, 1, 2, 3}; return L[c]; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};
struct Walker2dTopo {  // reference assets/walker2d.skel: pelvis - (thigh shin foot) x 2; only the feet collide
  static constexpr int NL = 7, NDOF = NL + 2, NC = 2, NA = 6, TIER0 = 2, TIER1 = 0, TIER1_F64 = 0;
  static constexpr bool WARM = true;   // measured +26 % (persistent double-support contacts)
  static constexpr bool WARM_FRICTION = false;   // round 5: the Schur-complement start beats the previous substep's friction state (constraint_phase)
#ifndef DART_NO_HINV_LDS
  static constexpr bool HINV_LDS_F64 = true;   // see topo_hinv_lds64
#endif
  __device__ __host__ static constexpr int parent(int k) { constexpr int P[NL] = {-1, 0, 1, 2, 0, 4, 5}; return P[k]; }
  __device__ __host__ static constexpr int clink(int c) { constexpr int L[NC] = {3, 6}; return L[c]; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};
struct Walker2dAllTopo {  // all seven capsules of walker2d.skel against the floor (DART's behaviour, the default card)
  static constexpr int NL = 7, NDOF = NL + 2, NC = 7, NA = 6, TIER0 = 2, TIER1 = 0, TIER1_F64 = 0;
#ifndef DART_WALKER2D_LIMIT_SLOTS
#define DART_WALKER2D_LIMIT_SLOTS 4
#endif
  // see topo_limit_slots (6 = one row per limited joint, rounds 1-5).  Measured, A/B on one box (profiles/r06_limit_slots.txt): 4 slots
  // fp64 101.2 -> 90.8 us, fp32 79.5 -> 72.1 us; 3 slots (a wave's maximum in 73 % of its substeps, the all-limits tier for the rest) 114.7 us.
  static constexpr int LIMIT_SLOTS = DART_WALKER2D_LIMIT_SLOTS;
  static constexpr bool WARM = true;
  static constexpr bool WARM_FRICTION = false;   // (see Walker2dTopo)
#ifndef DART_NO_HINV_LDS
  static constexpr bool HINV_LDS_F64 = true;   // see topo_hinv_lds64: 672 -> 368 B of scratch per lane, 134 -> 119 us, bitwise the same states
#endif
  __device__ __host__ static constexpr int parent(int k) { constexpr int P[NL] = {-1, 0, 1, 2, 0, 4, 5}; return P[k]; }
  __device__ __host__ static constexpr int clink(int c) { constexpr int L[NC] = {0, 1, 2, 3, 4, 5, 6}; return L[c]; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};

struct CheetahTopo {  // reference assets/half_cheetah.skel: torso (+ welded head) - (thigh shin foot) x 2, eight capsules, joint springs
  // An env beyond the two register slots -- a thrashing cheetah rests on 3 capsules in 4.5 % and on 4 in 0.23 % of its env-world-steps, 94 %
  // of the waves hold such a lane -- goes to the wave solvers: wave_constraints4, four envs per pass, one per row of 16 lanes (up to 16 rows;
  // beyond: wave_constraints, one at a time).  Rounds 2-4 had a second register tier of 4 (fp32) / 3 (fp64) slots here, 12 / 14 LCP rows,
  // first inlined (NOT TRUSTWORTHY on gfx950 / ROCm 7.2: spill copies ahead of an EXEC restore), then as a real call on
  // copies of its inputs; round 5 compiled it out of this kernel (fp64 1.383 -> 0.620, fp32 0.514 -> 0.394 ms per batched step of 65 536
  // envs, profiles/r05_halfcheetah_coop4.txt: the call and its argument block alone cost the step kernel 3.8 KB of its 6.3 KB of scratch per
  // lane), round 6 deleted it everywhere together with DART_CFG_WAVE_VOTE, the per-wave choice between it and the wave solvers (git history
  // has both).  Which solver serves an env is a function of the env alone, so a trajectory does not depend on the env's wave mates -- bitwise
  // (tests/test_gpu_spatial.py::test_half_cheetah_trajectories_do_not_depend_on_wave_mates, also with every env lying on the floor).  The
  // price: a batch in which EVERY env rests on three or four capsules pays 16 passes per world step (tools/gpu/cheetah_floor_probe.py).
  static constexpr int NL = 7, NDOF = NL + 2, NC = 8, NA = 6, TIER0 = 2, TIER1 = 0, TIER1_F64 = 0;
  static constexpr bool PLAIN_FRICTION_START = true;   // see topo_plain_friction_start
#ifndef DART_CHEETAH_LIMIT_SLOTS
#define DART_CHEETAH_LIMIT_SLOTS 3
#endif
  // see topo_limit_slots: a half cheetah's wave has at most 1 / 2 / 3 joints at their limits in 5 / 93.6 / 1.5 % of its substeps (host build,
  // 64-lane groups), never more in the sample: 3 of 6 limit rows -- the two-slot tier's LCPs shrink from 8 / 10 to 5 / 7 rows
  static constexpr int LIMIT_SLOTS = DART_CHEETAH_LIMIT_SLOTS;
  // Five touching capsules (four in fp64) are 6e-5 (2.3e-3) of the env-world-steps -- ~20 (~750) per launch of 65 536 envs -- and a
  // launch at one wave per SIMD lasts as long as its slowest wave.  WAVE_FALLBACK: such an env is served by the whole wave
  // (wave_constraints) instead of by its own lane alone, 2.50 -> 0.95 ms (fp32) / 3.43 -> 1.54 ms (fp64) per batched step; and the lanes
  // of a register tier that keep pivoting are handed to the same wave solver (blcp_bpp: coop / handoff), 0.95 -> 0.57 / 1.54 -> 1.50 ms.
  // (Opt-in per topology: with the calls in their kernels Hopper and Walker2d, whose bench workloads never need them, lose 9-12 %.)
  static constexpr bool WAVE_FALLBACK = true;
  static constexpr bool WARM = true;
  // Round 6: H^-1 out of the registers across the two-slot tier's pivoting loops, as the Walker2d kernels have it (topo_hinv_lds64) --
  // but the four-env wave solver's blocks (20.6 KB fp64) + an H^-1 block of its own (23 KB) would not fit the 40 KB a workgroup may
  // take at four workgroups per CU, so the column is parked LATE: after the wave-served phase, INTO its then idle blocks
  // (topo_hinv_lds_late; world_step).
#ifndef DART_CHEETAH_HLDS
#define DART_CHEETAH_HLDS 1
#endif
#if DART_CHEETAH_HLDS
  static constexpr bool HINV_LDS_F64 = (DART_CHEETAH_HLDS & 1) != 0, HINV_LDS_F32 = (DART_CHEETAH_HLDS & 2) != 0, HINV_LDS_LATE = true;
#endif
  __device__ __host__ static constexpr int parent(int k) { constexpr int P[NL] = {-1, 0, 1, 2, 0, 4, 5}; return P[k]; }
  __device__ __host__ static constexpr int clink(int c) { constexpr int L[NC] = {0, 0, 1, 2, 3, 4, 5, 6}; return L[c]; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};

struct SnakeTopo {  // reference assets/snake_7link.skel: seven links in a chain, sliding in the horizontal x-z plane 1 mm above the floor
  // Gravity is normal to the plane of motion and nothing can reach the floor: no contact rows at all (NC = 1 is a placeholder that
  // is never tested, TIER0 = 0), six limit rows, and the fluid force of snake_7link.py:37-47 on every body before every world step.
  static constexpr int NL = 7, NDOF = NL + 2, NC = 1, NA = 6, TIER0 = 0, TIER1 = 0, TIER1_F64 = 0;
  static constexpr bool WARM = true;
  static constexpr bool CONTACTS = false, FLUID = true, PLANE_XZ = true;
  __device__ __host__ static constexpr int parent(int k) { return k - 1; }
  __device__ __host__ static constexpr int clink(int) { return 0; }
  __device__ __host__ static constexpr bool limited(int k) { return k >= 1; }
};

// Physics-only variant of a topology (DART_TASK_NONE: what `dart_env_amd.envs.DartEnv` -- the reference's DartEnv base class,
// dart_env.py:28-175 -- runs a user's .skel on): every dof takes a generalized force (action = tau, no clamp, scale 1), the
// observation is [q, dq], reward 0, never done.  A user model whose tree matches a compiled topology then runs one env per lane
// instead of on the tree kernel (SURVEY.md 8(f)-1 "model compiler generality").
// A user model has no termination inside the library -- it may fall over and stay on the floor with every capsule touching.  The hopper
// chain keeps all four capsules in a register tier (an 11-row LCP, no fallback path at all; with the base topology's tiers 65 536 fallen
// pogo hoppers took 2.4 ms per env-step, every lane waiting its turn in the single-lane solver); the walker / cheetah trees have the two
// slots of their base and the wave solvers behind them, as the half cheetah (round 6; rounds 2-5: a second register tier of four / three
// slots as a real call, chosen per wave by DART_CFG_WAVE_VOTE -- results then depended on the wave mates in the last bits).
template <class T, class = void> struct topo_contacts { static constexpr bool value = true; };
template <class T> struct topo_contacts<T, decltype((void)T::CONTACTS)> { static constexpr bool value = T::CONTACTS; };
template <class B> struct PhysTopo : B {
  static constexpr bool ANC_TABLES_RT = true;   // see topo_anc_rt
  static constexpr int NA = B::NDOF;
  static constexpr bool PHYSICS = true;
  // (a base topology that cannot touch the floor -- the snake chain -- has no contact tiers to size)
  static constexpr int TIER1 = (topo_contacts<B>::value && B::NC <= 4) ? B::NC : 0, TIER1_F64 = TIER1;
  static constexpr bool WAVE_FALLBACK = topo_contacts<B>::value && (B::NC > 4);
  static constexpr bool PLAIN_FRICTION_START = topo_contacts<B>::value && (B::NC > 4);   // see topo_plain_friction_start
  // (a base that parks H^-1 in LDS -- the walker tree -- and gains the four-env blocks here parks it INTO them: both would not fit, see topo_hinv_lds_late)
  static constexpr bool HINV_LDS_LATE = topo_contacts<B>::value && (B::NC > 4);
};
// optional traits (default: a robot in the vertical x-y plane with capsules that can touch the floor, no fluid)
template <class T, class = void> struct topo_physics { static constexpr bool value = false; };
template <class T> struct topo_physics<T, decltype((void)T::PHYSICS)> { static constexpr bool value = T::PHYSICS; };
template <class T> __device__ __host__ constexpr int obs_dim_of() { return topo_physics<T>::value ? 2 * T::NDOF : 2 * T::NDOF - 1; }
template <class T, class = void> struct topo_fluid { static constexpr bool value = false; };
template <class T> struct topo_fluid<T, decltype((void)T::FLUID)> { static constexpr bool value = T::FLUID; };
template <class T, class = void> struct topo_plane_xz { static constexpr bool value = false; };
template <class T> struct topo_plane_xz<T, decltype((void)T::PLANE_XZ)> { static constexpr bool value = T::PLANE_XZ; };

// HINV_LDS_F64 / HINV_LDS_F32: keep H^-1 in LDS (one column of 64 lanes per packed entry) across the pivoting stages instead of in
// registers.  The 9-dof fp64 kernels need ~310 doubles live at the pivoting loops against the 256 the register file holds
// (512 32-bit registers): the allocator spilled ~84 of them to scratch (672 B per lane, 165 MB of HBM / L2 traffic per launch at
// 65 536 envs).  H^-1 (45 doubles) is needed before the loops (Delassus matrix) and after them (limit columns of the velocity
// change) but not inside: parked in LDS -- 23 KB per wave, conflict-free 8-byte columns, every entry fetched once per use --
// it leaves the register file to the loops: 672 -> 368 B of scratch, Walker2d fp64 134.1 -> 119.3 us (A/B on one box), states
// bitwise those of the register version.
template <class T, class = void> struct topo_hinv_lds64 { static constexpr bool value = false; };
template <class T> struct topo_hinv_lds64<T, decltype((void)T::HINV_LDS_F64)> { static constexpr bool value = T::HINV_LDS_F64; };
template <class T, class = void> struct topo_warm_friction { static constexpr bool value = true; };   // stage-2 start of a persisting contact's friction row from the previous substep (constraint_phase)
template <class T> struct topo_warm_friction<T, decltype((void)T::WARM_FRICTION)> { static constexpr bool value = T::WARM_FRICTION; };
// PLAIN_FRICTION_START: a friction row's start set from its own stiffness A_tt, not from the Schur complement over the contact's normal row
// (constraint_phase).  The half cheetah and the physics-only walker / cheetah trees: re-measured in round 5 on the two-slot tier, with /
// without either rule 30.0-30.6 wave solves per env-step -- nothing to gain, and their kernels were validated with this one.
template <class T, class = void> struct topo_plain_friction_start { static constexpr bool value = false; };
template <class T> struct topo_plain_friction_start<T, decltype((void)T::PLAIN_FRICTION_START)> { static constexpr bool value = T::PLAIN_FRICTION_START; };
template <class T, class = void> struct topo_wave_fallback { static constexpr bool value = false; };
template <class T> struct topo_wave_fallback<T, decltype((void)T::WAVE_FALLBACK)> { static constexpr bool value = T::WAVE_FALLBACK; };
template <class T, class = void> struct topo_hinv_lds32 { static constexpr bool value = false; };
template <class T> struct topo_hinv_lds32<T, decltype((void)T::HINV_LDS_F32)> { static constexpr bool value = T::HINV_LDS_F32; };
template <class T, class Real> __device__ __host__ constexpr bool hinv_lds() {
  return sizeof(Real) == 8 ? topo_hinv_lds64<T>::value : topo_hinv_lds32<T>::value;
}
// HINV_LDS_LATE: the column is written after the wave-served phase of the world step, into the LDS that phase used (the four-env blocks
// of wave_constraints4, idle from then on) -- topologies whose LDS budget has no room for both.  Until then H^-1 stays where it was computed.
template <class T, class = void> struct topo_hinv_lds_late { static constexpr bool value = false; };
template <class T> struct topo_hinv_lds_late<T, decltype((void)T::HINV_LDS_LATE)> { static constexpr bool value = T::HINV_LDS_LATE; };
template <class T, class Real> __device__ __host__ constexpr bool hinv_lds_late() { return hinv_lds<T, Real>() && topo_hinv_lds_late<T>::value; }
#ifndef DART_COMPILER_FENCE
#define DART_COMPILER_FENCE() asm volatile("" ::: "memory")   // no load / store of the compiler's moves across (it emits nothing)
#endif

template <class T>
__device__ __host__ constexpr bool is_anc(int j, int k) {  // j ancestor-or-self of k
  while (k >= 0) {
    if (k == j) return true;
    k = T::parent(k);
  }
  return false;
}
template <class T>
__device__ __host__ constexpr int n_limited() {
  int n = 0;
  for (int k = 0; k < T::NL; k++) n += T::limited(k) ? 1 : 0;
  return n;
}
template <class T, int NCA>
__device__ __host__ constexpr int limit_slot(int k) {  // LCP slot of link k's limit row when NCA contact slots precede the limits
  int s = 2 * NCA;
  for (int j = 0; j < k; j++) s += T::limited(j) ? 1 : 0;
  return s;
}
template <class T> __device__ __host__ constexpr int lim_ord(int k) {   // ordinal of limited link k among the limited links
  int o = 0;
  for (int j = 0; j < k; j++) o += T::limited(j) ? 1 : 0;
  return o;
}
template <class T> __device__ __host__ constexpr int lim_link(int o) {  // the o-th limited link
  int c = 0;
  for (int k = 0; k < T::NL; k++) { if (T::limited(k)) { if (c == o) return k; c++; } }
  return 0;
}
// LIMIT_SLOTS (optional trait; round 6): the register tiers carry this many joint-limit rows instead of one per limited joint -- slot s
// holds the s-th joint that IS at a limit this substep, as the contact slots hold the s-th touching capsule -- and a wave in which some
// lane has more joints at their limits runs the tier with all of them.  Measured on the host build over 64-lane groups (random-action
// rollouts, 2 048 envs): a Hopper lane has 0 / 1 / 2 / 3 of its 3 limited joints at a limit in 8 / 78 / 14 / 0.008 % of its substeps and a
// wave's maximum is 2 in 99.5 %; a Walker2d lane 0 / 1 / 2 / 3 / 4 of 6 in 36 / 44 / 17 / 2.4 / 0.7 %, never more, a wave's maximum 2 / 3 / 4 in
// 24 / 48 / 27 %.  The pivoting loops cost ~M^3: Hopper's LCPs shrink from 4 / 5 rows (frictionless / friction stage) to 3 / 4, Walker2d's
// from 8 / 10 to 6 / 8.
template <class T, class = void> struct topo_limit_slots { static constexpr int value = -1; };
template <class T> struct topo_limit_slots<T, decltype((void)T::LIMIT_SLOTS)> { static constexpr int value = T::LIMIT_SLOTS; };
// LIMIT_PREFIX (optional trait; round 6): the small tier carries the limit rows of the FIRST LIMIT_PREFIX limited joints only, each in its own
// row -- no compaction, none of its selects -- and a wave in which some lane has a LATER joint at its limit runs the tier with all of them.  For
// a chain whose last joint practically never reaches its limits: the Hopper's foot joint does in 8e-5 of the lane-substeps (a wave: 0.5 %).
template <class T, class = void> struct topo_limit_prefix { static constexpr int value = -1; };
template <class T> struct topo_limit_prefix<T, decltype((void)T::LIMIT_PREFIX)> { static constexpr int value = T::LIMIT_PREFIX; };
template <class T> __device__ __host__ constexpr int small_limit_prefix() {
  return (topo_limit_prefix<T>::value >= 0 && topo_limit_prefix<T>::value < n_limited<T>()) ? topo_limit_prefix<T>::value : n_limited<T>();
}
template <class T> __device__ __host__ constexpr int small_limit_slots() {
  return (topo_limit_slots<T>::value >= 0 && topo_limit_slots<T>::value < n_limited<T>()) ? topo_limit_slots<T>::value : n_limited<T>();
}
template <class T, class Real> __device__ __host__ constexpr int tier1() { return sizeof(Real) == 8 ? T::TIER1_F64 : T::TIER1; }
template <class T, class Real> __device__ __host__ constexpr int last_tier() { return tier1<T, Real>() > 0 ? tier1<T, Real>() : T::TIER0; }
template <class T, class Real> __device__ __host__ constexpr bool has_slow_path() { return topo_contacts<T>::value && T::NC > last_tier<T, Real>(); }
template <class T> __device__ __host__ constexpr int max_rows() { return 2 * T::NC + n_limited<T>(); }
// ancestor-or-self sets of all links, 8 bits per link (NL <= 8): bit j of byte k = is_anc(j, k)
template <class T>
__device__ __host__ constexpr unsigned long long anc_table() {
  static_assert(T::NL <= 8, "ancestor table packs 8 links");
  unsigned long long t = 0;
  for (int k = 0; k < T::NL; k++)
    for (int j = 0; j < T::NL; j++) if (is_anc<T>(j, k)) t |= 1ull << (8 * k + j);
  return t;
}
// The two tables as COMPILE-TIME values.  anc_table<T>() / T::clink(c) written into an ordinary expression are constexpr FUNCTIONS the
// compiler may evaluate at run time -- and did, in every kernel of the all-shapes topologies (found in the disassembly, round 4): the
// slot compaction of every world step walked parent tables in constant memory, one dependent s_load + s_waitcnt per hop (6 sites in the
// Hopper kernel, 56 in Walker2d's).  A static constexpr data member is a constant expression by construction.
template <class T> struct AncTable { static constexpr unsigned long long value = anc_table<T>(); };
// ANC_TABLES_RT (optional trait): the topology keeps the run-time evaluation.  The half cheetah and the physics-only variants do: with
// the compile-time tables the fp32 half-cheetah kernel's FIRST launch in a process differed from the later ones on gfx950
// (tests/test_gpu_repeatability.py; the round-2 symptom of the largest register-tier kernels, cause still unknown) -- their code
// is left exactly as it was measured and verified.
template <class T, class = void> struct topo_anc_rt { static constexpr bool value = false; };
template <class T> struct topo_anc_rt<T, decltype((void)T::ANC_TABLES_RT)> { static constexpr bool value = T::ANC_TABLES_RT; };
template <class T>
__device__ __host__ constexpr unsigned long long clink_packed() {   // link of capsule c in bits [4c, 4c + 4)
  static_assert(T::NC <= 16 && T::NL <= 16, "clink table packs 16 capsules of 4 bits");
  unsigned long long t = 0;
  for (int c = 0; c < T::NC; c++) t |= (unsigned long long)T::clink(c) << (4 * c);
  return t;
}
template <class T> struct ClinkTable { static constexpr unsigned long long value = clink_packed<T>(); };
// ancestor-or-self mask of capsule c's link, c a run-time index: shifts of two immediates, no memory
template <class T> __device__ __host__ __forceinline__ uint32_t anc_mask_of_capsule(int c) {
  if constexpr (topo_anc_rt<T>::value) return (uint32_t)((anc_table<T>() >> (8 * T::clink(c))) & 0xffull);
  const int lk = (int)((ClinkTable<T>::value >> (4 * c)) & 0xfull);
  return (uint32_t)((AncTable<T>::value >> (8 * lk)) & 0xffull);
}
// LDS words of the single-lane fallback solver (slow_constraints): H^-1 full, kinematics, candidates, rows
template <class T>
__device__ __host__ constexpr int slow_words() {
  constexpr int N = T::NDOF, M = max_rows<T>();
  return N * N + 3 * T::NL + N + 4 * T::NC + 3 * T::NL + 2 * M * N + 2 * (M * (M + 1) / 2) + 8 * M + 16;
}
// LDS words of a step kernel's constraint scratch (the fallback solver's block)
template <class T, class Real>
__device__ __host__ constexpr int constraint_lds_words() { return has_slow_path<T, Real>() ? slow_words<T>() : 1; }

// Optional per-launch extras of both parameter flavours (all null / off by default):
//   ext_force      [n_envs][3] world-frame force added at link ext_link's frame origin before every world step --
//                  bodynodes[b].add_ext_force(F), reference dart_env.py:159-172
//   creport        [n_envs][NC][8] contacts of the env-step's LAST world step {body, -1, point xyz, force on the body xyz} --
//                  world.collision_result.contacts as walker2d.py:38-41 reads it; creport_count [n_envs]; cf_report [n_envs][NDOF] =
//                  skel.constraint_forces() of that step (J^T lambda / dt)
//   mt             the per-env MT19937 bank (mt19937_draw.hpp): with it, an env that finishes draws its reset noise from ITS numpy stream in
//                  the step kernel's epilogue -- the reference-exact auto-reset, no second and third launch; null: Philox (or none)
template <class Real>
struct Extras {
  const Real* ext_force = nullptr;
  int ext_link = 0;
  Real* creport = nullptr;
  int* creport_count = nullptr;
  Real* cf_report = nullptr;
  const MtBankView* mt = nullptr;
};
// where a world step reports to (null record pointer: nothing to report)
template <class Real>
struct ReportTo { Real* rec = nullptr; int* count = nullptr; Real* cf = nullptr; };

// ------------------------------------------------------------------ runtime parameters (kernel argument -> SGPRs)
template <class Real, class T>
struct Params {
  using Topo = T;
  static constexpr bool is_static = false;  // static_models.hpp holds the compile-time variants
  __device__ __host__ static constexpr bool zero(int, int) { return false; }
  Real dt, ground_y, g, mu, erp_dt, max_erv, limit_erp_dt, cfm1;  // cfm1 = 1 + cfm (DART scales diag(A)): limit rows
  Real ccfm1;                                                     // 1 + contact cfm: contact rows (ContactConstraint's own value)
  Real root_x0, root_y0;
  Real sigma[T::NL], mass[T::NL], cx[T::NL], cy[T::NL], izz[T::NL], jx[T::NL], jy[T::NL];
  Real lo[T::NL], hi[T::NL];
  Real damp[T::NDOF], q0[T::NDOF], dq0[T::NDOF];  // joint damping; world.reset() state
  Real stiff[T::NDOF], rest[T::NDOF];             // joint springs (implicit: H += dt^2 K, rhs -= K (q + dt dq - rest))
  Real sqe[T::NDOF];                              // sqrt(dt damp + dt^2 stiff): the implicit terms E = diag(sqe^2) of the forward dynamics
  int impulse_M;                                  // card.impulse_inertia (A3): 1 = the impulse pass runs on M (DART 6), 0 = on M + E
  Real e1x[T::NC], e1y[T::NC], e2x[T::NC], e2y[T::NC], rad[T::NC];
  Real act_scale[T::NA], act_lo[T::NA], act_hi[T::NA];
  Real alive, ctrl_cost, pen_each, pen_margin, h_lo, h_hi, ang_max, s_max, v_clip, inv_envdt, noise, noise_v;
  int frame_skip, max_steps, penalty_link, task;
  int solver, iters1, iters2;  // solver 0: block principal pivoting (exact); 1: PGS sweeps
  int force_slow;              // 1: debug / test knob (DART_CFG_DEBUG_FORCE_FALLBACK), every lane with a contact takes the fallback solver
  unsigned long long* stats;   // optional [2][32] histogram of wave-level pivoting iterations per stage (debug), or null
  int cbody[T::NC];            // card body index of each candidate capsule (contact report)
  Extras<Real> ex;
  Real fluid_k;                // topologies with FLUID: force -k (v_com . n) n on every body, n = the body's in-plane normal
  Real dev_cost;               // task 9 (snake): reward -= dev_cost |q[2]|
};

// compile-time "is this model parameter exactly zero" (always false for the runtime block): lets the specialised
// kernels drop whole terms without asking the compiler for non-IEEE x*0 folding
enum { ZF_jx = 0, ZF_jy = 1, ZF_cx = 2, ZF_cy = 3, ZF_damp = 4, ZF_stiff = 5 };
#define DART_ZERO(PT, field, k) (PT::zero(ZF_##field, k))

// ------------------------------------------------------------------ math helpers
template <class Real> __device__ __forceinline__ void sincos_(Real x, Real& s, Real& c);
// fp32 sin/cos without the Payne-Hanek slow path: 3-term Cody-Waite reduction by pi/2 (exact products via fma, good
// for |x| < ~1e4 rad; joint angles are bounded by limits / termination long before that) + minimax polynomials on
// [-pi/4, pi/4] (max error ~1 ulp).  ~25 VALU instructions, branch-free.
template <> __device__ __forceinline__ void sincos_<float>(float x, float& s, float& c) {
  const float kf = rintf(x * 0.63661977236758134f);
  const int k = (int)kf;
  float r = fmaf(kf, -1.5707962512969971f, x);
  r = fmaf(kf, -7.5497894158615964e-8f, r);
  r = fmaf(kf, -5.3903029534742384e-15f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float sn = fmaf(ps * r2, r, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float cs = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const bool swap = k & 1;
  const float s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
// fp64 sin/cos without the library's Payne-Hanek path and its branches: Cody-Waite reduction by pi/2 in three 33-bit pieces
// (products with |k| < 2^20 are exact) + the fdlibm kernel polynomials on [-pi/4, pi/4] (< 1 ulp).  ~35 VALU instructions.
template <> __device__ __forceinline__ void sincos_<double>(double x, double& s, double& c) {
  const double kf = rint(x * 6.36619772367581382433e-01);
  const int k = (int)kf;
  double r = fma(kf, -1.57079632673412561417e+00, x);
  r = fma(kf, -6.07710050630396597660e-11, r);
  r = fma(kf, -2.02226624871116645580e-21, r);
  r = fma(kf, -8.47842766036889956997e-32, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double sn = fma(ps * z, r, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double cs = fma(pc * z, z, fma(z, -0.5, 1.0));
  const bool swap = k & 1;
  const double s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
// The same function with its 17 constants REMATERIALISED at every use (round 5, tree kernel).  Where the scalar registers are oversubscribed
// (the tree kernel: S pointers, model fields, wave masks) the compiler keeps these loop-invariant doubles in VGPR pairs across the whole
// frame loop and, out of registers, spills them before the loop and reloads them every world step (10 of the 30 pairs the fp64 pattern
// kernel spilled, found in the disassembly).  An `asm volatile` s_mov pair cannot be hoisted: the constant exists for the one FMA that reads
// it (an SGPR-pair operand, SALU issue).  Same arithmetic, bitwise the same results.  Host builds (tests/kernel_emu) use the plain literals.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DART_NO_REMAT64)
template <unsigned long long B> __device__ __forceinline__ double remat64_() {
  int lo, hi;
  asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "i"((int)(unsigned)(B & 0xffffffffull)));
  asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "i"((int)(unsigned)(B >> 32)));
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
#define DART_REMAT64(x) (::dartk::remat64_<__builtin_bit_cast(unsigned long long, (double)(x))>())
#else
#define DART_REMAT64(x) ((double)(x))
#endif
template <class Real> __device__ __forceinline__ void sincos_remat_(Real x, Real& s, Real& c) { sincos_<Real>(x, s, c); }
template <> __device__ __forceinline__ void sincos_remat_<double>(double x, double& s, double& c) {
  const double kf = rint(x * DART_REMAT64(6.36619772367581382433e-01));
  const int k = (int)kf;
  double r = fma(kf, DART_REMAT64(-1.57079632673412561417e+00), x);
  r = fma(kf, DART_REMAT64(-6.07710050630396597660e-11), r);
  r = fma(kf, DART_REMAT64(-2.02226624871116645580e-21), r);
  r = fma(kf, DART_REMAT64(-8.47842766036889956997e-32), r);
  const double z = r * r;
  double ps = fma(z, DART_REMAT64(1.58969099521155010221e-10), DART_REMAT64(-2.50507602534068634195e-08));
  ps = fma(ps, z, DART_REMAT64(2.75573137070700676789e-06));
  ps = fma(ps, z, DART_REMAT64(-1.98412698298579493134e-04));
  ps = fma(ps, z, DART_REMAT64(8.33333333332248946124e-03));
  ps = fma(ps, z, DART_REMAT64(-1.66666666666666324348e-01));
  const double sn = fma(ps * z, r, r);
  double pc = fma(z, DART_REMAT64(-1.13596475577881948265e-11), DART_REMAT64(2.08757232129817482790e-09));
  pc = fma(pc, z, DART_REMAT64(-2.75573143513906633035e-07));
  pc = fma(pc, z, DART_REMAT64(2.48015872894767294178e-05));
  pc = fma(pc, z, DART_REMAT64(-1.38888888888741095749e-03));
  pc = fma(pc, z, DART_REMAT64(4.16666666666666019037e-02));
  const double cs = fma(pc * z, z, fma(z, -0.5, 1.0));
  const bool swap = k & 1;
  const double s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
// reciprocal: hardware v_rcp_f32 (1 ulp) + one Newton step in fp32; v_rcp_f64 + two Newton steps in fp64
template <class Real> __device__ __forceinline__ Real rcp_(Real x);
template <> __device__ __forceinline__ float rcp_<float>(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}
template <> __device__ __forceinline__ double rcp_<double>(double x) {   // v_rcp_f64 + two Newton steps (pivots are far from the denormal range)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}
// reciprocal square root: v_rsq_f32 + one Newton step in fp32; v_rsq_f64 + two in fp64
template <class Real> __device__ __forceinline__ Real rsqrt_(Real x);
template <> __device__ __forceinline__ float rsqrt_<float>(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * fmaf(-0.5f * x * r, r, 1.5f);
}
template <> __device__ __forceinline__ double rsqrt_<double>(double x) {   // v_rsq_f64 + two Newton steps
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r * fma(-0.5 * x * r, r, 1.5);
}
// rsqrt_<Real>(x) with its dependent operations handed out one at a time -- same operations, same result -- so that a caller with
// independent work (the systolic Cholesky of the tree kernel: the previous column's remaining updates) can put it between them: a lone
// wave issues in order and would otherwise wait out every link of the chain (v_rsq_f64 + 6 dependent fp64 operations per column).
template <class Real> __device__ __forceinline__ Real rsq_seed_(Real x);
template <> __device__ __forceinline__ float rsq_seed_<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <> __device__ __forceinline__ double rsq_seed_<double>(double x) { return __builtin_amdgcn_rsq(x); }
template <class Real>
struct RsqStaged {
  Real r, h, t;
  static constexpr int NSTAGE = 3 * (sizeof(Real) == 4 ? 1 : 2);   // Newton steps of rsqrt_: 1 (fp32), 2 (fp64), three operations each
  __device__ __forceinline__ void start(Real x) { r = rsq_seed_<Real>(x); h = Real(-0.5) * x; }
  __device__ __forceinline__ void step(int s) {   // s = 0 .. NSTAGE-1, in order
    const int ph = s % 3;
    if (ph == 0) t = h * r;
    else if (ph == 1) t = fma(t, r, Real(1.5));
    else r = r * t;
  }
};
template <class Real> __device__ __forceinline__ Real inf_() { return Real(__builtin_huge_valf()); }
template <class Real> __device__ __forceinline__ Real tol_() { return sizeof(Real) == 4 ? Real(2e-6) : Real(1e-12); }

}  // namespace dartk
#if defined(__HIPCC__) || defined(DART_WAVE_EMU)   // the wave-cooperative solver (readlane, DPP): device build (and the fiber emulation of
                                                    // tests/kernel_emu/fake_wave_include); the lane-at-a-time host build of the lane
#include "wave_blcp.hpp"   // kernels (tests/kernel_emu) serves the same rows with the single-lane loops of slow_constraints
#define DART_WAVE_COOP 1
#endif
namespace dartk {

// In-place inverse of a packed symmetric positive definite matrix (lower, tri(i,j)) via LDL^T.
template <class Real, int N>
__device__ __forceinline__ void spd_inverse(Real (&a)[N * (N + 1) / 2]) {
  Real invd[N];
  // factor: a(i,j), i>j becomes L(i,j); a(j,j) becomes d_j
  sfor<0, N>([&](auto J) {
    constexpr int j = J;
    Real W[N];  // W[k] = L(j,k) d_k, shared by the pivot and every entry of column j
    Real d = a[tri(j, j)];
    sfor<0, j>([&](auto K) { constexpr int k = K; W[k] = a[tri(j, k)] * a[tri(k, k)]; d -= a[tri(j, k)] * W[k]; });
    a[tri(j, j)] = d;
    invd[j] = rcp_<Real>(d);
    sfor<j + 1, N>([&](auto I) {
      constexpr int i = I;
      Real t = a[tri(i, j)];
      sfor<0, j>([&](auto K) { constexpr int k = K; t -= a[tri(i, k)] * W[k]; });
      a[tri(i, j)] = t * invd[j];
    });
  });
  // invert the unit lower factor in place: Li(i,j) = -sum_{k=j}^{i-1} L(i,k) Li(k,j)
  sfor<0, N>([&](auto J) {
    constexpr int j = J;
    sfor<j + 1, N>([&](auto I) {
      constexpr int i = I;
      Real t = a[tri(i, j)];  // k = j term: L(i,j) * Li(j,j)=1
      sfor<j + 1, i>([&](auto K) { constexpr int k = K; t += a[tri(i, k)] * a[tri(k, j)]; });
      a[tri(i, j)] = -t;
    });
  });
  // note: the loop above must read L(i,k) (k>j) untouched and Li(k,j) (k<i) already inverted: column j is
  // finished top-down before column j+1 starts, and L(i,k) for k>j lives in later columns -> correct.
  // Hinv(i,j) = sum_{k>=i} Li(k,i) invd_k Li(k,j)   (i>=j), computed column-major so inputs stay intact
  Real out[N * (N + 1) / 2];
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    sfor<0, i + 1>([&](auto J) {
      constexpr int j = J;
      Real t = (i == j) ? invd[i] : a[tri(i, j)] * invd[i];  // k = i term (Li(i,i) = 1)
      sfor<i + 1, N>([&](auto K) { constexpr int k = K; t += (a[tri(k, i)] * invd[k]) * a[tri(k, j)]; });
      out[tri(i, j)] = t;
    });
  });
  sfor<0, N*(N + 1) / 2>([&](auto I) { a[I] = out[I]; });
}

// Solve A x = rhs for a packed SPD matrix via LDL^T (A is destroyed, x holds rhs on entry).
template <class Real, int K>
__device__ __forceinline__ void spd_solve(Real (&a)[K * (K + 1) / 2], Real (&x)[K]) {
  Real invd[K];
  sfor<0, K>([&](auto J) {
    constexpr int j = J;
    Real W[K];
    Real d = a[tri(j, j)];
    sfor<0, j>([&](auto Kk) { constexpr int k = Kk; W[k] = a[tri(j, k)] * a[tri(k, k)]; d -= a[tri(j, k)] * W[k]; });
    a[tri(j, j)] = d;
    invd[j] = rcp_<Real>(d);
    sfor<j + 1, K>([&](auto I) {
      constexpr int i = I;
      Real t = a[tri(i, j)];
      sfor<0, j>([&](auto Kk) { constexpr int k = Kk; t -= a[tri(i, k)] * W[k]; });
      a[tri(i, j)] = t * invd[j];
    });
  });
  sfor<0, K>([&](auto I) { constexpr int i = I; sfor<0, i>([&](auto Kk) { constexpr int k = Kk; x[i] -= a[tri(i, k)] * x[k]; }); });
  sfor<0, K>([&](auto I) { x[I] *= invd[I]; });
  sfor_rev<0, K>([&](auto I) { constexpr int i = I; sfor<i + 1, K>([&](auto Kk) { constexpr int k = Kk; x[i] -= a[tri(k, i)] * x[k]; }); });
}

// DART integrates joint damping and springs implicitly in the FORWARD DYNAMICS only, (M + E) qdd = rhs with E = dt D + dt^2 K,
// while its impulse pass (the constraint rows' unit-impulse responses and the final velocity change) runs on M alone
// (include/dart_model_card.h, impulse_inertia).  The lane kernels hold M^-1 explicitly for the
// constraint rows, so the acceleration comes from it through the symmetric Woodbury identity over the K dofs whose E is not zero,
//     (M + E)^-1 r = y - M^-1 S (I + S M^-1 S)^-1 S y,     y = M^-1 r,  S = sqrt(E)   (no division by E: a zero entry is harmless)
// one K x K LDL^T solve instead of a second N x N factorisation (Hopper: K = 3 of 6, Walker2d: 6 of 9).
// Sel::count / Sel::dof(a): the dofs with E != 0; REV: Minv is stored in reversed dof order.  a = y on entry, qdd on return.
template <class PT, int N> struct ImplicitDofs {
  // baked models: exactly the dofs with damping or a spring; runtime parameter block: every joint dof -- the three root dofs of a
  // planar model carry neither (fill_params declines a card that has them there)
  __device__ __host__ static constexpr bool has(int i) { return PT::is_static ? !(PT::zero(ZF_damp, i) && PT::zero(ZF_stiff, i)) : (i >= 3); }
  __device__ __host__ static constexpr int cnt() { int c = 0; for (int i = 0; i < N; i++) c += has(i) ? 1 : 0; return c; }
  static constexpr int count = cnt();
  __device__ __host__ static constexpr int dof(int a) { int c = 0; for (int i = 0; i < N; i++) if (has(i)) { if (c == a) return i; c++; } return 0; }
};
template <int N> struct AllDofs {
  static constexpr int count = N;
  __device__ __host__ static constexpr int dof(int a) { return a; }
};
template <class Real, int N, bool REV, class Sel, class PT>
__device__ __forceinline__ void implicit_accel(const PT& P, const Real (&Minv)[N * (N + 1) / 2], Real (&a)[N]) {
  constexpr int K = Sel::count;
  if constexpr (K > 0) {
    Real G[K * (K + 1) / 2], z[K];
    sfor<0, K>([&](auto A_) {
      constexpr int aa = A_, da = Sel::dof(aa), ia = REV ? N - 1 - da : da;
      sfor<0, aa + 1>([&](auto B_) {
        constexpr int bb = B_, db = Sel::dof(bb), ib = REV ? N - 1 - db : db;
        const Real g = (P.sqe[da] * P.sqe[db]) * Minv[tri(ia, ib)];
        G[tri(aa, bb)] = (aa == bb) ? g + Real(1) : g;
      });
      z[aa] = P.sqe[da] * a[da];
    });
    spd_solve<Real, K>(G, z);
    sfor<0, K>([&](auto A_) { constexpr int aa = A_; z[aa] *= P.sqe[Sel::dof(aa)]; });
    sfor<0, N>([&](auto I) {
      constexpr int i = I, ii = REV ? N - 1 - i : i;
      sfor<0, K>([&](auto A_) { constexpr int aa = A_, da = Sel::dof(aa), ia = REV ? N - 1 - da : da; a[i] -= Minv[tri(ii, ia)] * z[aa]; });
    });
  }
}

// Solve the masked symmetric system for the free set of a boxed LCP iteration.
//   free[i] in {0,1};  row i not free:  x_i = rhs_i ;  free rows:  sum_j A_ij x_j = rhs_i over free j
template <class Real, int M>
__device__ __forceinline__ void masked_solve(const Real (&A)[M * (M + 1) / 2], const Real (&fr)[M], Real (&x)[M]) {
  Real L[M * (M + 1) / 2], invd[M];
  sfor<0, M>([&](auto J) {
    constexpr int j = J;
    Real W[M];
    Real d = fr[j] * A[tri(j, j)] + (Real(1) - fr[j]);
    sfor<0, j>([&](auto K) { constexpr int k = K; W[k] = L[tri(j, k)] * L[tri(k, k)]; d -= L[tri(j, k)] * W[k]; });
    L[tri(j, j)] = d;
    invd[j] = rcp_<Real>(d);
    const Real fj = fr[j] * invd[j];
    sfor<j + 1, M>([&](auto I) {
      constexpr int i = I;
      Real t = A[tri(i, j)] * fr[i];          // rows / columns outside the free set drop out (their L entries stay 0)
      sfor<0, j>([&](auto K) { constexpr int k = K; t -= L[tri(i, k)] * W[k]; });
      L[tri(i, j)] = t * fj;
    });
  });
  sfor<0, M>([&](auto I) {  // forward: L y = rhs
    constexpr int i = I;
    sfor<0, i>([&](auto K) { constexpr int k = K; x[i] -= L[tri(i, k)] * x[k]; });
  });
  sfor<0, M>([&](auto I) { x[I] *= invd[I]; });
  sfor_rev<0, M>([&](auto I) {  // backward: L^T x = y
    constexpr int i = I;
    sfor<i + 1, M>([&](auto K) { constexpr int k = K; x[i] -= L[tri(k, i)] * x[k]; });
  });
}

// Boxed LCP:  w = A x - b,  lo <= x <= hi,  complementarity.  Block principal pivoting (Judice-Pires) with a
// least-index fallback, one lane per problem, wavefront vote to stop.  The active set is two per-lane bit masks
// (F: free rows, U: rows held at their upper bound; the rest sit at the lower bound) so a set flip is a handful of
// integer ops instead of per-row branches.  Rows in `pinmask` (lo == hi) never move.
// ZERO_BOUNDS: every finite bound is 0 (the frictionless stage) -> the bound contribution to the rhs vanishes.
// Measurement build (-DDART_WAVE_TIMING, never the product): where do the cycles of a wave go?  With DART_CFG_STATS on, the 64 counters hold
//   [0] sum, [1] max, [2] count of whole-wave cycles per launched wave; [3] / [4] / [5] sum of cycles in the fallback solver / the big
//   tier / the small tier; [8..15] waves by log2(cycles) - 14; [16..31] / [32..47] / [48..63] invocations of the fallback / big tier /
//   small tier by log2(cycles) - 8   (the pivoting histograms are not recorded in this build).
#ifdef DART_WAVE_TIMING
#define DART_CLK() ((long long)__builtin_readcyclecounter())
__device__ __forceinline__ void wave_timing_add(unsigned long long* st, int sum_slot, int hist0, int hist_shift, int nbins, long long dt) {
  if (st == nullptr || (threadIdx.x & 63) != 0) return;
  atomicAdd(&st[sum_slot], (unsigned long long)dt);
  int l2 = 63 - __clzll(dt > 1 ? dt : 1) - hist_shift;
  l2 = l2 < 0 ? 0 : (l2 >= nbins ? nbins - 1 : l2);
  atomicAdd(&st[hist0 + l2], 1ull);
}
#endif

// Wave-served solves (the hand-off of a register tier, wave_constraints, wave_constraints4) run one env -- or four -- after the other.
// Round 6: what a solve may spend is ITS OWN business -- per env and stage min(the stage's cap, DART_COOP_BUDGET) wave
// iterations, whatever its wave mates needed (every half-cheetah problem of the 65 536-env statistics converges within 60) -- so a result
// is a function of the env alone also in a wave whose lanes all lie on the floor.  (Rounds 3-5 shared ONE budget of 96 / 192 iterations
// among all the solves of a wave and world step: sized for rare fallbacks, it ran out in contact-heavy waves once the half cheetah's
// second register tier was gone -- up to 16 passes a world step -- and the late envs kept a clamped, unconverged iterate that depended
// on how much the early ones had used.)  What is left of the shared budget is a RUNAWAY GUARD of 64 x DART_COOP_BUDGET iterations per
// serving place and world step: it can bind only when the equivalent of every lane of the wave runs into its cap -- a simulation that
// is blowing up -- and then the remaining lanes get DART_COOP_MIN iterations each and keep their last iterate, clamped into the box,
// as a lane does at its own cap.
#ifndef DART_COOP_BUDGET
#define DART_COOP_BUDGET 96
#define DART_COOP_MIN 8
#endif
#define DART_COOP_GUARD (64 * DART_COOP_BUDGET)
__device__ __host__ constexpr int coop_iters(int cap, int guard) {   // iterations one wave-served solve may run
  const int c = cap < DART_COOP_BUDGET ? cap : DART_COOP_BUDGET;
  const int b = guard > DART_COOP_MIN ? guard : DART_COOP_MIN;
  return c < b ? c : b;
}

// coop / handoff (topologies with WAVE_FALLBACK, device build): after `handoff` iterations the lanes that have not converged are
// served one at a time by the whole wave -- the owner parks its problem in LDS (`coop`, coop_words<M>() Reals), the register solver of
// wave_blcp.hpp continues from the owner's current sets with lane i on row i.  A wave lasts as long as its slowest lane, and a
// launch as long as its slowest wave: the few lanes that cycle (10-40 iterations where the others need 1-3) then cost iterations of
// a 16-row wave solve (~3 k cycles) instead of iterations of the whole tier.
template <int M> __device__ __host__ constexpr int coop_words() { return M * (M + 1) / 2 + 4 * M; }
// the four LDS blocks of wave_constraints4 (four envs per pass, one per row of 16 lanes; device build, topologies with WAVE_FALLBACK whose
// candidate capsules and limits fit a row of 16 lanes) -- DART_COOP4 = 0 builds the kernels without it
#ifndef DART_COOP4
#define DART_COOP4 1
#endif
template <class T> __device__ __host__ constexpr int coop4_words() {
  constexpr int N = T::NDOF;
  return N * N + 3 * T::NL + N + 4 * T::NC + 2 * T::NL + 2 * 16 * N + 16 * 17 / 2 + 4 * 16;
}
template <class T> __device__ __host__ constexpr bool coop4_fits() { return T::NC + T::NL <= 16 && T::NDOF <= 16; }
template <class T> __device__ __host__ constexpr int coop4_lds_words() {
#if defined(DART_WAVE_COOP) && DART_COOP4
  return (topo_wave_fallback<T>::value && coop4_fits<T>()) ? 4 * coop4_words<T>() : 0;
#else
  return 0;
#endif
}
template <class Real, int M, bool ZERO_BOUNDS>
__device__ __forceinline__ void blcp_bpp(const Real (&A)[M * (M + 1) / 2], const Real (&b)[M], const Real (&lo)[M],
                                         const Real (&hi)[M], uint32_t pinmask, uint32_t& F, uint32_t& U,
                                         Real (&x)[M], int max_iter, unsigned long long* stats, Real bmax_more = Real(0),
                                         Real* coop = nullptr, int handoff = 0) {
  // feasibility tolerances scale with the problem: |b|_inf bounds the size of w and (through A^-1) of x
  // (bmax_more: |b| of rows the caller left out of this solve because they are pinned at 0)
  Real bmax = bmax_more;
  sfor<0, M>([&](auto I) { bmax = fmax(bmax, fabs(b[I])); });
  const Real tol = tol_<Real>() * (Real(1) + bmax);
  int best = M + 1, patience = 3;
  bool conv = false;
  int it = 0;
#ifdef DART_EMU_TRACE
  const uint32_t trace_F0 = F, trace_U0 = U;   // host build only (tests/kernel_emu): active-set statistics of the pivoting loop
#endif
  for (; it < max_iter; ++it) {
    Real fr[M], xb[M], r[M];
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      const bool f = (F >> i) & 1u, u = (U >> i) & 1u;
      fr[i] = f ? Real(1) : Real(0);
      xb[i] = f ? Real(0) : (u ? hi[i] : lo[i]);
    });
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      Real t = b[i];
      if constexpr (!ZERO_BOUNDS) sfor<0, M>([&](auto J) { constexpr int j = J; t -= A[tri(i, j)] * xb[j]; });
      r[i] = ((F >> i) & 1u) ? t : xb[i];
    });
    masked_solve<Real, M>(A, fr, r);
    uint32_t B = 0, GT = 0;
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      Real w = -b[i];
      sfor<0, M>([&](auto J) { constexpr int j = J; w += A[tri(i, j)] * r[j]; });
      DART_PIN_VGPR(w);  // keep w unconditional: otherwise the compiler sinks it into per-row exec-mask branches
      const bool f = (F >> i) & 1u, u = (U >> i) & 1u, pinned = (pinmask >> i) & 1u;
      const bool over = r[i] > hi[i] + tol * (Real(1) + fabs(hi[i]));
      const bool under = r[i] < lo[i] - tol * (Real(1) + fabs(lo[i]));
      const bool wbad = u ? (w > tol) : (w < -tol);
      const bool inf = f ? (over || under) : (wbad && !pinned);
      B |= inf ? (1u << i) : 0u;
      GT |= (r[i] > hi[i]) ? (1u << i) : 0u;
    });
    sfor<0, M>([&](auto I) { x[I] = conv ? x[I] : r[I]; });
    conv = conv || (B == 0u);
    if (__all(conv)) break;
    const int ninf = __popc(B);
    const bool improved = ninf < best;
    const bool single = !improved && patience == 0;
    best = improved ? ninf : best;
    patience = improved ? 3 : (patience > 0 ? patience - 1 : 0);
    uint32_t Bs = single ? (1u << (31 - __clz((int)B))) : B;   // Murty: only the highest infeasible index
    Bs = conv ? 0u : Bs;
    const uint32_t toBound = Bs & F, toFree = Bs & ~F;
    F = (F & ~toBound) | toFree;
    U = (U & ~(toFree | toBound)) | (toBound & GT);
#ifdef DART_WAVE_COOP
    if (coop != nullptr && it + 1 >= handoff) { ++it; break; }
#endif
  }
#ifdef DART_WAVE_COOP
  if (coop != nullptr) {
    static_assert(M <= 16, "hand-off uses the 16-row register solver");
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long todo = __ballot(!conv);
    int budget = DART_COOP_GUARD;   // runaway guard over all the lanes handed off here (see coop_iters)
#ifdef DART_WAVE_TIMING_FALLBACK   // [30] envs handed off, [31] cycles spent serving them
    const long long th0 = (long long)__builtin_readcyclecounter();
    if (stats && lane == 0 && todo != 0ull) atomicAdd(&stats[30], (unsigned long long)__popcll(todo));
#endif
    while (todo != 0ull) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      if (lane == owner) {
        sfor<0, M * (M + 1) / 2>([&](auto K) { coop[K] = A[K]; });
        Real* v = coop + M * (M + 1) / 2;
        sfor<0, M>([&](auto I) { constexpr int i = I; v[i] = b[i]; v[M + i] = lo[i]; v[2 * M + i] = hi[i]; v[3 * M + i] = x[i]; });
      }
      __syncthreads();
      const Real* v = coop + M * (M + 1) / 2;
      const BlcpSets res = sp_blcp_t<Real, 16, true>(coop, v, v + M, v + 2 * M, coop + M * (M + 1) / 2 + 3 * M, M, (uint64_t)__shfl(pinmask, owner),
                                               (uint64_t)__shfl(F, owner), (uint64_t)__shfl(U, owner), coop_iters(max_iter - it, budget), nullptr, lane,
                                               ZERO_BOUNDS, __shfl(bmax_more, owner));
      budget -= res.iters;
      __syncthreads();
      if (lane == owner && res.ok) {
        sfor<0, M>([&](auto I) { constexpr int i = I; x[i] = v[3 * M + i]; });
        F = (uint32_t)res.F; U = (uint32_t)res.U;
      }
      __syncthreads();
    }
#ifdef DART_WAVE_TIMING_FALLBACK
    if (stats && lane == 0) atomicAdd(&stats[31], (unsigned long long)((long long)__builtin_readcyclecounter() - th0));
#endif
  }
#endif
  // iteration cap reached without a feasible complementary point: stay in the box
  sfor<0, M>([&](auto I) { constexpr int i = I; x[i] = fmin(fmax(x[i], lo[i]), hi[i]); });
#ifndef DART_WAVE_TIMING
  if (stats && (threadIdx.x & 63) == 0) atomicAdd(&stats[it < 31 ? it : 31], 1ull);
#endif
#ifdef DART_EMU_TRACE
  dart_emu_trace(M, ZERO_BOUNDS ? 1 : 0, trace_F0, trace_U0, F, U, pinmask, it);
#endif
}

// PRESOLVE32 (fp64 kernels only): the active set is searched in fp32 first -- a copy of the problem, at most `presolve_iters` pivoting
// iterations, no hand-off -- and the fp64 loop starts from the set that search ended on: it confirms it with one solve when the fp32
// search was right, and simply continues when it was not (the fp64 loop and its tolerances stay the authority on the result).
// For the half cheetah's big fp64 tier, whose 12-row fp64 iterations are spill-bound (6 KB of scratch per lane) and whose waves need
// 2.4 + 3.7 of them per world step: 1.50 -> 1.43 ms per batched step.  Not for the small tiers: in every fp64 tier of every topology
// it cost Hopper 32.0 -> 36.4 us and Walker2d 119 -> 141 us (a wave's 2nd..4th solve is cheaper than the copies and the extra solve).
template <class Real, int M, bool ZERO_BOUNDS, bool PRESOLVE32>
__device__ __forceinline__ void blcp_bpp_mixed(const Real (&A)[M * (M + 1) / 2], const Real (&b)[M], const Real (&lo)[M],
                                               const Real (&hi)[M], uint32_t pinmask, uint32_t& F, uint32_t& U,
                                               Real (&x)[M], int max_iter, unsigned long long* stats, Real bmax_more,
                                               Real* coop, int handoff, int presolve_iters) {
  if constexpr (PRESOLVE32 && sizeof(Real) == 8) {
    float Af[M * (M + 1) / 2], bf[M], lof[M], hif[M], xf[M];
    sfor<0, M * (M + 1) / 2>([&](auto K) { Af[K] = (float)A[K]; });
    sfor<0, M>([&](auto I) { bf[I] = (float)b[I]; lof[I] = (float)lo[I]; hif[I] = (float)hi[I]; xf[I] = 0.f; });
    uint32_t Ff = F, Uf = U;
    blcp_bpp<float, M, ZERO_BOUNDS>(Af, bf, lof, hif, pinmask, Ff, Uf, xf, presolve_iters < max_iter ? presolve_iters : max_iter, nullptr,
                                    (float)bmax_more);
    F = Ff; U = Uf;
  }
  blcp_bpp<Real, M, ZERO_BOUNDS>(A, b, lo, hi, pinmask, F, U, x, max_iter, stats, bmax_more, coop, handoff);
}

template <class Real, int M>
__device__ __forceinline__ void blcp_pgs(const Real (&A)[M * (M + 1) / 2], const Real (&b)[M], const Real (&lo)[M],
                                         const Real (&hi)[M], const bool (&skip)[M], Real (&x)[M], int iters) {
  Real invd[M];
  sfor<0, M>([&](auto I) { constexpr int i = I; invd[i] = rcp_<Real>(A[tri(i, i)]); });
  for (int it = 0; it < iters; ++it) {
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      Real r = b[i];
      sfor<0, M>([&](auto J) { constexpr int j = J; r -= A[tri(i, j)] * x[j]; });
      Real xn = x[i] + r * invd[i];
      xn = fmin(fmax(xn, lo[i]), hi[i]);
      x[i] = skip[i] ? x[i] : xn;
    });
  }
}

// Active sets of the previous substep (registers, per lane): contacts and limits persist over the frame_skip substeps,
// so the pivoting solver usually starts on the right set.  Any start gives the same (unique) LCP solution.
struct WarmSets {
  uint32_t cid = 0;                // candidate capsule held by each contact slot (4 bits per slot)
  uint32_t nca = 0;                // contact slots of the tier that wrote these sets (row positions differ between tiers)
  // (limit rows: one bit per limited joint behind the 2 nca contact bits, whatever the tier's limit slots -- constraint_phase translates)
  uint32_t sig = 0, up = 0;        // which rows were active / which of them rested on their upper bound
  uint32_t F1 = 0, U1 = 0;         // final sets of the frictionless stage
  uint32_t F2 = 0, U2 = 0;         // final sets of the friction stage
};

// ------------------------------------------------------------------ constraint phase on NCA compacted contact slots
// Inputs: H^-1 (packed, reversed dof order), link origins px / py, the unconstrained velocity vs (in/out), the candidate
// contacts (con / cPx / cPy / cdep over the T::NC capsules) and the state q (limits).  A lane with `off` set takes no part
// (it is served by slow_constraints): all its rows are inactive and its vs comes back unchanged.
// HLDS: H^-1 is read from this lane's LDS column `hl` (entry k at hl[64 k]) instead of from H (topo_hinv_lds64).
// NLS: limit slots (topo_limit_slots); -1 = one row per limited joint.  LPFX (topo_limit_prefix): the NLS rows are the FIRST NLS limited joints,
// each in its own row (no compaction) -- the caller guarantees that no lane of the wave has a later joint at its limit.
template <class Real, class T, class PT, int NCA, bool EXTRAS, bool HLDS = false, int NLS = -1, bool LPFX = false>
__device__ __forceinline__ void constraint_phase(const PT& P, const Real (&q)[T::NDOF], const Real (&H)[T::NDOF * (T::NDOF + 1) / 2],
                                                 const Real (&px)[T::NL], const Real (&py)[T::NL], Real (&vs)[T::NDOF],
                                                 const bool (&con)[T::NC], const Real (&cPx)[T::NC], const Real (&cPy)[T::NC],
                                                 const Real (&cdep)[T::NC], bool off, WarmSets& warm, const ReportTo<Real>& rp,
                                                 const Real* hl = nullptr, Real* cm = nullptr) {
  constexpr int NLIM = n_limited<T>(), NLSE = (NLS < 0 || NLS >= NLIM) ? NLIM : NLS;
  constexpr bool LIDENT = NLSE == NLIM || LPFX;   // limit row o IS limited joint o (all of them, or the first NLSE: LPFX)
  constexpr int NL = T::NL, N = T::NDOF, NC = T::NC, M = 2 * NCA + NLSE;
  constexpr int NLA = NLIM > 0 ? NLIM : 1, NLSA = NLSE > 0 ? NLSE : 1;
  // hand-off of the lanes that keep pivoting to the wave solver (blcp_bpp; cm = its LDS block, null = never): after how many
  // iterations of the frictionless / the friction stage, for the big tier (whose iterations are the expensive ones) and the others
#ifndef DART_HANDOFF_BIG
#define DART_HANDOFF_BIG 3, 5
#define DART_HANDOFF_SMALL 5, 8
#endif
  constexpr int HO_[4] = {DART_HANDOFF_BIG, DART_HANDOFF_SMALL};
  constexpr bool BIGT = tier1<T, Real>() > 0 && NCA == tier1<T, Real>();
  constexpr int HO1 = BIGT ? HO_[0] : HO_[2], HO2 = BIGT ? HO_[1] : HO_[3];
  constexpr bool PRE32 = false;   // (blcp_bpp_mixed: the fp32 pre-search paid only for the 12-row fp64 tier of rounds 2-5)
  auto Hv = [&](int k) -> Real { if constexpr (HLDS) return hl[64 * k]; else return H[k]; };
  constexpr bool IDENT = (NCA == NC);   // slot s IS candidate s: links are compile-time constants
  constexpr int NS = NCA > 0 ? NCA : 1;   // array extent of the slot arrays (a tier without contact slots still declares them)
  // ---- compaction: slot s takes the s-th touching capsule (capsule order = the oracle's serial order)
  bool son[NS];
  Real sPx[NS], sPy[NS], sdep[NS];
  uint32_t samask[NS];   // ancestor-or-self set of the slot's link (bit j = link j moves the contact point)
  uint32_t cid = 0;
  if constexpr (IDENT) {
    sfor<0, NCA>([&](auto S) {
      constexpr int sl = S;
      son[sl] = con[sl] && !off; sPx[sl] = cPx[sl]; sPy[sl] = cPy[sl]; sdep[sl] = cdep[sl];
      if constexpr (topo_anc_rt<T>::value) samask[sl] = (uint32_t)((anc_table<T>() >> (8 * T::clink(sl))) & 0xffull);
      else { constexpr uint32_t am_ = (uint32_t)((AncTable<T>::value >> (8 * T::clink(sl))) & 0xffull); samask[sl] = am_; }
    });
  } else {
    sfor<0, NCA>([&](auto S) { son[S] = false; sPx[S] = Real(0); sPy[S] = Real(0); sdep[S] = Real(0); samask[S] = 0u; });
    int rank = 0;
    sfor<0, NC>([&](auto Cc) {
      constexpr int c = Cc;
      const bool hit = con[c] && !off;
      sfor<0, NCA>([&](auto S) {
        constexpr int sl = S;
        const bool take = hit && rank == sl;
        son[sl] = son[sl] || take;
        sPx[sl] = take ? cPx[c] : sPx[sl]; sPy[sl] = take ? cPy[c] : sPy[sl]; sdep[sl] = take ? cdep[c] : sdep[sl];
        if constexpr (topo_anc_rt<T>::value) samask[sl] = take ? (uint32_t)((anc_table<T>() >> (8 * T::clink(c))) & 0xffull) : samask[sl];
        else {
          constexpr uint32_t am_ = (uint32_t)((AncTable<T>::value >> (8 * T::clink(c))) & 0xffull);   // (a constant expression: see AncTable)
          samask[sl] = take ? am_ : samask[sl];
        }
        cid = take ? (cid | ((uint32_t)c << (4 * sl))) : cid;
      });
      rank += hit ? 1 : 0;
    });
  }
  Real A[M * (M + 1) / 2], b[M], lo[M], hi[M], x[M];
  bool act[M];
  Real Jn[NS][N], Jt[NS][N], Yn[NS][N], Yt[NS][N];
  bool any = false;
  sfor<0, NCA>([&](auto S) {
    constexpr int sl = S;
    Jn[sl][0] = Real(0); Jn[sl][1] = Real(1);
    Jt[sl][0] = Real(-1); Jt[sl][1] = Real(0);
    sfor<0, NL>([&](auto J) {
      constexpr int j = J;
      if constexpr (IDENT) {
        if constexpr (is_anc<T>(j, T::clink(sl))) {
          Jn[sl][2 + j] = P.sigma[j] * (sPx[sl] - px[j]);
          Jt[sl][2 + j] = P.sigma[j] * (sPy[sl] - py[j]);
        } else {
          Jn[sl][2 + j] = Real(0); Jt[sl][2 + j] = Real(0);
        }
      } else {
        const bool a = (samask[sl] >> j) & 1u;
        Jn[sl][2 + j] = a ? P.sigma[j] * (sPx[sl] - px[j]) : Real(0);
        Jt[sl][2 + j] = a ? P.sigma[j] * (sPy[sl] - py[j]) : Real(0);
      }
    });
    Real rn = Real(0), rt = Real(0);
    sfor<0, N>([&](auto I) { constexpr int i = I; rn += Jn[sl][i] * vs[i]; rt += Jt[sl][i] * vs[i]; });
    const Real bounce = fmin(sdep[sl] * P.erp_dt, P.max_erv);
    constexpr int sn = 2 * sl, stt = 2 * sl + 1;
    const bool on = son[sl];
    act[sn] = on; act[stt] = on;
    b[sn] = on ? (bounce - rn) : Real(0);
    b[stt] = on ? -rt : Real(0);
    lo[sn] = Real(0); hi[sn] = on ? inf_<Real>() : Real(0);
    lo[stt] = Real(0); hi[stt] = Real(0);  // friction rows are pinned at 0 during stage 1
    any = any || on;
  });
  uint32_t lid = 0;   // compacted limit rows: ordinal of the limited joint each slot holds
  if constexpr (LIDENT) {
  sfor<0, NL>([&](auto K) {
    constexpr int k = K;
    if constexpr (T::limited(k) && lim_ord<T>(k) < NLSE) {
      constexpr int sl = limit_slot<T, NCA>(k), i = 2 + k;
      const bool low = !off && q[i] <= P.lo[k], up = !off && (!low) && (q[i] >= P.hi[k]);
      const Real viol = low ? (q[i] - P.lo[k]) : (q[i] - P.hi[k]);
      const Real bounce = fmin(fmax(-viol * P.limit_erp_dt, -P.max_erv), P.max_erv);
      const bool on = low || up;
      act[sl] = on;
      b[sl] = on ? (bounce - vs[i]) : Real(0);
      lo[sl] = low ? Real(0) : (up ? -inf_<Real>() : Real(0));
      hi[sl] = low ? inf_<Real>() : Real(0);
      any = any || on;
    }
  });
  } else {
    // slot s takes the s-th joint that is at a limit (joint order); a lane with more such joints than slots never gets here (world_step's vote)
    sfor<0, NLSE>([&](auto S) { constexpr int r = 2 * NCA + S; act[r] = false; b[r] = Real(0); lo[r] = Real(0); hi[r] = Real(0); });
    int lrank = 0;
    sfor<0, NL>([&](auto K) {
      constexpr int k = K;
      if constexpr (T::limited(k)) {
        constexpr int o = lim_ord<T>(k), i = 2 + k;
        const bool low = !off && q[i] <= P.lo[k], up = !off && (!low) && (q[i] >= P.hi[k]);
        const Real viol = low ? (q[i] - P.lo[k]) : (q[i] - P.hi[k]);
        const Real bounce = fmin(fmax(-viol * P.limit_erp_dt, -P.max_erv), P.max_erv);
        const bool on = low || up;
        const Real bk = bounce - vs[i], lok = low ? Real(0) : -inf_<Real>(), hik = low ? inf_<Real>() : Real(0);
        sfor<0, NLSE>([&](auto S) {
          constexpr int r = 2 * NCA + S;
          const bool take = on && lrank == (int)S;
          act[r] = act[r] || take;
          b[r] = take ? bk : b[r]; lo[r] = take ? lok : lo[r]; hi[r] = take ? hik : hi[r];
          lid = take ? (lid | ((uint32_t)o << (4 * (int)S))) : lid;
        });
        lrank += on ? 1 : 0;
        any = any || on;
      }
    });
  }
  if (!__any(any)) return;

  // Y = H^-1 J^T for contact rows (limit rows: columns of H^-1), Delassus matrix A = J H^-1 J^T
  if constexpr (HLDS) {   // every packed entry of H^-1 is fetched from LDS once and serves both triangles and all slots
    sfor<0, NCA>([&](auto S) { sfor<0, N>([&](auto I) { Yn[S][I] = Real(0); Yt[S][I] = Real(0); }); });
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      sfor<0, i + 1>([&](auto J) {
        constexpr int j = J;
        const Real h = Hv(tri(rev<N>(i), rev<N>(j)));
        sfor<0, NCA>([&](auto S) {
          constexpr int sl = S;
          Yn[sl][i] += h * Jn[sl][j]; Yt[sl][i] += h * Jt[sl][j];
          if constexpr (i != j) { Yn[sl][j] += h * Jn[sl][i]; Yt[sl][j] += h * Jt[sl][i]; }
        });
      });
    });
  } else
  sfor<0, NCA>([&](auto S) {
    constexpr int sl = S;
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real a = Real(0), t = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; a += Hv(tri(rev<N>(i), rev<N>(j))) * Jn[sl][j]; t += Hv(tri(rev<N>(i), rev<N>(j))) * Jt[sl][j]; });
      Yn[sl][i] = a; Yt[sl][i] = t;
    });
  });
  sfor<0, NCA>([&](auto Ca) {
    constexpr int a = Ca;
    sfor<0, a + 1>([&](auto Cb) {
      constexpr int bb = Cb;
      Real nn = Real(0), nt = Real(0), tn = Real(0), ttv = Real(0);
      sfor<0, N>([&](auto I) {
        constexpr int i = I;
        nn += Jn[a][i] * Yn[bb][i]; nt += Jn[a][i] * Yt[bb][i];
        tn += Jt[a][i] * Yn[bb][i]; ttv += Jt[a][i] * Yt[bb][i];
      });
      A[tri(2 * a, 2 * bb)] = nn;
      A[tri(2 * a + 1, 2 * bb + 1)] = ttv;
      A[tri(2 * a + 1, 2 * bb)] = tn;
      if constexpr (a != bb) A[tri(2 * a, 2 * bb + 1)] = nt;
    });
  });
  if constexpr (LIDENT) {
  sfor<0, NL>([&](auto K) {
    constexpr int k = K;
    if constexpr (T::limited(k) && lim_ord<T>(k) < NLSE) {
      constexpr int sl = limit_slot<T, NCA>(k), i = 2 + k;
      sfor<0, NCA>([&](auto S) {
        constexpr int cs = S;
        A[tri(sl, 2 * cs)] = Yn[cs][i];
        A[tri(sl, 2 * cs + 1)] = Yt[cs][i];
      });
      sfor<0, k + 1>([&](auto J) {
        constexpr int j = J;
        if constexpr (T::limited(j)) A[tri(sl, limit_slot<T, NCA>(j))] = Hv(tri(rev<N>(i), rev<N>(2 + j)));   // (j <= k: its ordinal is below NLSE too)
      });
    }
  });
  } else {
    // compacted limit rows: the row of slot s is the unit vector of ITS joint -- a lane-varying index.  Contact couplings: a select over the
    // limited joints' entries of Y; limit-limit entries of H^-1: one indexed read of the lane's LDS column (HLDS), else a two-level select.
    sfor<0, NLSE>([&](auto S) {
      constexpr int sl = S, r = 2 * NCA + sl;
      const uint32_t os = (lid >> (4 * sl)) & 15u;
      sfor<0, NCA>([&](auto C) {
        constexpr int cs = C;
        Real yn = Yn[cs][2 + lim_link<T>(0)], yt = Yt[cs][2 + lim_link<T>(0)];
        sfor<1, NLIM>([&](auto O) { constexpr int o = O; const bool is = os == (uint32_t)o; yn = is ? Yn[cs][2 + lim_link<T>(o)] : yn; yt = is ? Yt[cs][2 + lim_link<T>(o)] : yt; });
        A[tri(r, 2 * cs)] = yn;
        A[tri(r, 2 * cs + 1)] = yt;
      });
      if constexpr (HLDS) {
        int is_ = 0;   // storage index of the slot's dof: rev<N>(2 + link)
        sfor<0, NLIM>([&](auto O) { constexpr int o = O; is_ = os == (uint32_t)o ? rev<N>(2 + lim_link<T>(o)) : is_; });
        sfor<0, sl + 1>([&](auto Tt) {
          constexpr int tl = Tt, rt = 2 * NCA + tl;
          const uint32_t ot = (lid >> (4 * tl)) & 15u;
          int it_ = 0;
          sfor<0, NLIM>([&](auto O) { constexpr int o = O; it_ = ot == (uint32_t)o ? rev<N>(2 + lim_link<T>(o)) : it_; });
          const int hi_ = is_ > it_ ? is_ : it_, lo_ = is_ > it_ ? it_ : is_;
          A[tri(r, rt)] = hl[64 * (hi_ * (hi_ + 1) / 2 + lo_)];
        });
      } else {
        Real hrow[NLA];   // H^-1 [joint of slot s][limited joint o2]
        sfor<0, NLIM>([&](auto O2) {
          constexpr int o2 = O2;
          Real v = H[tri(rev<N>(2 + lim_link<T>(0)), rev<N>(2 + lim_link<T>(o2)))];
          sfor<1, NLIM>([&](auto O1) { constexpr int o1 = O1; v = os == (uint32_t)o1 ? H[tri(rev<N>(2 + lim_link<T>(o1)), rev<N>(2 + lim_link<T>(o2)))] : v; });
          hrow[o2] = v;
        });
        sfor<0, sl + 1>([&](auto Tt) {
          constexpr int tl = Tt, rt = 2 * NCA + tl;
          const uint32_t ot = (lid >> (4 * tl)) & 15u;
          Real v = hrow[0];
          sfor<1, NLIM>([&](auto O) { constexpr int o = O; v = ot == (uint32_t)o ? hrow[o] : v; });
          A[tri(r, rt)] = v;
        });
      }
    });
  }
  // inactive slots: decouple (unit diagonal keeps the factorisations regular)
  sfor<0, M>([&](auto I) {
    constexpr int i = I;
    A[tri(i, i)] = act[i] ? A[tri(i, i)] * (i < 2 * NCA ? P.ccfm1 : P.cfm1) : Real(1);
    sfor<0, i>([&](auto J) { constexpr int j = J; if (!act[i] || !act[j]) A[tri(i, j)] = Real(0); });
  });

  // initial active set: every row at its finite bound, except rows that x = 0 already violates (w = -b has the
  // wrong sign) -- those start free, which is what the first pivoting iteration would have found
  uint32_t pinmask = 0, F = 0, U = 0, sig = 0, up = 0;
  bool has_contact = false;
  Real bmax0 = Real(0);
  sfor<0, M>([&](auto I) { bmax0 = fmax(bmax0, fabs(b[I])); });
  const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
  sfor<0, M>([&](auto I) {
    constexpr int i = I;
    x[i] = Real(0);
    const bool pinned = !(lo[i] < hi[i]);
    const bool upper = !(lo[i] == Real(0));   // (-inf, 0] rows rest on their upper bound
    const bool start_free = !pinned && (upper ? (b[i] < -tol0) : (b[i] > tol0));
    pinmask |= pinned ? (1u << i) : 0u;
    F |= start_free ? (1u << i) : 0u;
    U |= (upper && !start_free) ? (1u << i) : 0u;
    sig |= act[i] ? (1u << i) : 0u;
    up |= (act[i] && upper) ? (1u << i) : 0u;
  });
#ifdef DART_LCP_STAGE1_SWEEP
  {
    // (experiment, off by default; tests/diag/diag_lcp_active_sets.py) one projected Gauss-Seidel sweep over the rows that can move,
    // started from each violated row's own impulse b_i / A_ii: rows the others push across their bound start free, rows another
    // row already takes care of start on their bound
    Real xe[M];
    sfor<0, M>([&](auto I) { constexpr int i = I; xe[i] = ((F >> i) & 1u) ? b[i] * rcp_<Real>(A[tri(i, i)]) : Real(0); });
    sfor<0, M>([&](auto I) {
      constexpr int i = I;
      const bool pinned = (pinmask >> i) & 1u, upper = !(lo[i] == Real(0));
      Real r = b[i];
      sfor<0, M>([&](auto J) { constexpr int j = J; if constexpr (j != i) r -= A[tri(i, j)] * xe[j]; });
      Real xn = r * rcp_<Real>(A[tri(i, i)]);
      xn = upper ? fmin(xn, Real(0)) : fmax(xn, Real(0));
      xn = (pinned || !act[i]) ? Real(0) : xn;
      xe[i] = xn;
      const bool fr = upper ? (xn < Real(0)) : (xn > Real(0));
      F = fr ? (F | (1u << i)) : (F & ~(1u << i));
      U = (upper && !fr && !pinned) ? (U | (1u << i)) : (U & ~(1u << i));
    });
  }
#endif
  sfor<0, NCA>([&](auto S) { has_contact = has_contact || act[2 * S]; });
  // rows that were active on the same side in the previous substep (contact slots: and hold the same capsule) inherit
  // that substep's final set
  // The warm sets are kept in the layout of the all-limits tier -- contact rows, then ONE BIT PER LIMITED JOINT -- whatever tier wrote them, and
  // translated into this tier's slots here and back at the end: a joint's row inherits its own set across a change of slot AND across a change of
  // tier.  With that, which of the two limit layouts a wave votes for cannot change a lane's numbers: the masked factorisation of the all-limits
  // tier only ever adds exact zeros for the rows that are not there in the compacted one, the rows that are there keep their relative order (slot
  // order = joint order), and both start from the same sets -- the batch-independence tests hold bitwise across the vote.
  auto from_joint_bits = [&](uint32_t m) -> uint32_t {
    if constexpr (LIDENT) return m;
    else {
      uint32_t r = m & ((1u << (2 * NCA)) - 1u);
      sfor<0, NLSE>([&](auto S) { constexpr int sl = S; r |= (act[2 * NCA + sl] ? ((m >> (2 * NCA + ((lid >> (4 * sl)) & 15u))) & 1u) : 0u) << (2 * NCA + sl); });
      return r;
    }
  };
  auto to_joint_bits = [&](uint32_t m) -> uint32_t {
    if constexpr (LIDENT) return m;
    else {
      uint32_t r = m & ((1u << (2 * NCA)) - 1u);
      sfor<0, NLSE>([&](auto S) { constexpr int sl = S; r |= (act[2 * NCA + sl] ? ((m >> (2 * NCA + sl)) & 1u) : 0u) << (2 * NCA + ((lid >> (4 * sl)) & 15u)); });
      return r;
    }
  };
  const uint32_t wsig = from_joint_bits(warm.sig), wup = from_joint_bits(warm.up), wF1 = from_joint_bits(warm.F1), wU1 = from_joint_bits(warm.U1);
  uint32_t keep = (warm.nca == (uint32_t)NCA) ? ~0u : 0u;
  if constexpr (!IDENT) {
    sfor<0, NCA>([&](auto S) {
      constexpr int sl = S;
      const bool same_capsule = ((warm.cid >> (4 * sl)) & 15u) == ((cid >> (4 * sl)) & 15u);
      keep = same_capsule ? keep : (keep & ~(3u << (2 * sl)));
    });
  }
  const uint32_t same = T::WARM ? (sig & wsig & ~(up ^ wup) & ~pinmask & keep) : 0u;
  F = (F & ~same) | (wF1 & same);
  U = (U & ~same) | (wU1 & same);

  if (P.solver == 0) {
    if constexpr (NCA > 0) {
      // The friction rows are pinned at 0 in this stage: solve the M - NCA rows that can move (contact normals, limits) as a
      // system of their own.  Leaving the pinned rows in as identity rows gives the same numbers bit for bit (a masked row only
      // ever contributes exact zeros) at the cost of the full-size factorisation.
      constexpr int M1 = M - NCA;
      auto full = [](int k) constexpr { return k < NCA ? 2 * k : NCA + k; };   // row k of the small system in the full one
      Real A1[M1 * (M1 + 1) / 2], b1[M1], lo1[M1], hi1[M1], x1[M1];
      uint32_t pin1 = 0, F1 = 0, U1 = 0;
      Real bt = Real(0);
      sfor<0, NCA>([&](auto S) { bt = fmax(bt, fabs(b[2 * S + 1])); });
      sfor<0, M1>([&](auto I) {
        constexpr int i = I, fi = full(i);
        b1[i] = b[fi]; lo1[i] = lo[fi]; hi1[i] = hi[fi]; x1[i] = Real(0);
        pin1 |= ((pinmask >> fi) & 1u) << i; F1 |= ((F >> fi) & 1u) << i; U1 |= ((U >> fi) & 1u) << i;
        sfor<0, i + 1>([&](auto J) { constexpr int j = J; A1[tri(i, j)] = A[tri(fi, full(j))]; });
      });
      blcp_bpp_mixed<Real, M1, true, PRE32>(A1, b1, lo1, hi1, pin1, F1, U1, x1, P.iters1, P.stats, bt, cm, HO1, 4);
      F = 0; U = 0;
      sfor<0, M1>([&](auto I) {
        constexpr int i = I, fi = full(i);
        x[fi] = x1[i];
        F |= ((F1 >> i) & 1u) << fi; U |= ((U1 >> i) & 1u) << fi;
      });
    } else {
      blcp_bpp_mixed<Real, M, true, PRE32>(A, b, lo, hi, pinmask, F, U, x, P.iters1, P.stats, Real(0), cm, HO1, 4);
    }
  }
  else {
    bool skip[M];
    sfor<0, M>([&](auto I) { skip[I] = (pinmask >> I) & 1u; });
    blcp_pgs<Real, M>(A, b, lo, hi, skip, x, P.iters1);
  }

  warm.F1 = to_joint_bits(F); warm.U1 = to_joint_bits(U);
  if (__any(has_contact)) {
    // ODE/DART friction bounds: +-mu * (normal impulse of the frictionless solve), then the full problem
    uint32_t fric = 0;
    sfor<0, NCA>([&](auto S) {
      constexpr int sn = 2 * S, stt = 2 * S + 1;
      Real hb = act[sn] ? fabs(P.mu * x[sn]) : Real(0);
      hi[stt] = hb; lo[stt] = -hb;
      const bool pinned = !(hb > Real(0));
      pinmask = pinned ? (pinmask | (1u << stt)) : (pinmask & ~(1u << stt));
      // start of the friction row: the impulse that would stop the tangential velocity left by the frictionless solve, with
      // every other row held -- inside the bounds the row starts free (sticking), beyond them on that bound (sliding)
      Real wt = -b[stt];
      sfor<0, M>([&](auto J) { constexpr int j = J; wt += A[tri(stt, j)] * x[j]; });
      // ... and with the contact's own normal row free to respond: the tangential stiffness the friction row sees is then the Schur
      // complement A_tt - A_tn^2 / A_nn.  (tests/diag/diag_lcp_active_sets.py: with the plain A_tt 92 % of the wrong stage-2 guesses
      // were friction rows guessed sticking that ended up sliding; with the complement a Hopper lane needs one stage-2 solve in 98 %
      // instead of 84 % of the substeps, which takes a wave -- the maximum over its lanes -- from 3 solves to 2 or 1: 34.1 -> 31.2 us.)
      // (topo_plain_friction_start: the half cheetah and the physics-only walker / cheetah trees keep the plain A_tt)
      Real att = A[tri(stt, stt)];
      if constexpr (!topo_plain_friction_start<T>::value) {
        const bool nfree = (F >> sn) & 1u;
        att = nfree ? att - A[tri(stt, sn)] * A[tri(stt, sn)] * rcp_<Real>(A[tri(sn, sn)]) : att;
      }
      const Real xe = -wt * rcp_<Real>(att);
      const bool up = xe > hb, dn = xe < -hb;
      F = (pinned || up || dn) ? (F & ~(1u << stt)) : (F | (1u << stt));
      U = (!pinned && up) ? (U | (1u << stt)) : (U & ~(1u << stt));
      fric |= pinned ? 0u : (1u << stt);
    });
    // ... unless the same contact was sliding/sticking a substep ago: start from that state
    // Round 5, measured on the host build over 64-lane groups (tests/diag/diag_wave_solves.py): where the Schur-complement start above is in
    // force it predicts a friction row's state BETTER than the contact's state a substep ago -- Walker2d: a lane needs a second stage-2
    // solve in 1.5 % of the substeps with it, in 14 % with the warm start on top, and since a wave iterates until its slowest lane is done
    // that is 1.3 against 2.75 stage-2 solves per wave and substep (15.2 against 19.4 solves per wave and env-step in all).  The half
    // cheetah (plain A_tt start, 4.3 stage-2 solves per wave either way) keeps the warm start: 33.5 against 34.7.  WARM_FRICTION, per topology.
    const uint32_t samef = topo_warm_friction<T>::value ? (fric & (same << 1)) : 0u;   // friction row of a contact whose normal row persisted
    F = (F & ~samef) | (warm.F2 & samef);
    U = (U & ~samef) | (warm.U2 & samef);
    if (P.solver == 0) blcp_bpp_mixed<Real, M, false, PRE32>(A, b, lo, hi, pinmask, F, U, x, P.iters2, P.stats ? P.stats + 32 : nullptr, Real(0), cm, HO2, 6);
    else {
      bool skip[M];
      sfor<0, M>([&](auto I) { skip[I] = !has_contact; });   // per-env semantics: no contact -> no second stage
      blcp_pgs<Real, M>(A, b, lo, hi, skip, x, P.iters2);
    }
    warm.F2 = F; warm.U2 = U;
  }
  warm.sig = to_joint_bits(sig); warm.up = to_joint_bits(up); warm.cid = cid; warm.nca = (uint32_t)NCA;
  // impulse of every limited joint's row (compacted rows: scattered back to their joints)
  Real xl[NLA];
  sfor<0, NLIM>([&](auto O) {
    constexpr int o = O;
    if constexpr (LIDENT) { if constexpr (o < NLSE) xl[o] = x[2 * NCA + o]; else xl[o] = Real(0); }
    else {
      Real v = Real(0);
      sfor<0, NLSE>([&](auto S) { constexpr int sl = S; v = (act[2 * NCA + sl] && ((lid >> (4 * sl)) & 15u) == (uint32_t)o) ? x[2 * NCA + sl] : v; });
      xl[o] = v;
    }
  });
  if (EXTRAS && rp.rec != nullptr && !off) {   // contact records in capsule order, constraint forces J^T lambda / dt
    const Real idt = Real(1) / P.dt;
    int r = 0;
    sfor<0, NCA>([&](auto S) {
      constexpr int sl = S;
      if (son[sl]) {
        int body = 0;
        if constexpr (IDENT) body = P.cbody[sl];
        else sfor<0, NC>([&](auto Cc) { body = (((cid >> (4 * sl)) & 15u) == (uint32_t)Cc) ? P.cbody[Cc] : body; });
        Real* o = rp.rec + 8 * r;
        o[0] = (Real)body; o[1] = Real(-1);
        o[2] = P.root_x0 + q[0] + sPx[sl]; o[3] = P.root_y0 + q[1] + sPy[sl]; o[4] = Real(0);
        o[5] = -x[2 * sl + 1] * idt; o[6] = x[2 * sl] * idt; o[7] = Real(0);   // n = +y, t1 = z x n = -x (DART's tangent basis)
        r++;
      }
    });
    *rp.count = r;
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real f = Real(0);
      sfor<0, NCA>([&](auto S) { constexpr int cs = S; f += Jn[cs][i] * x[2 * cs] + Jt[cs][i] * x[2 * cs + 1]; });
      if constexpr (i >= 3) { if constexpr (T::limited(i - 2)) f += xl[lim_ord<T>(i - 2)]; }
      rp.cf[i] = f * idt;
    });
  }
  // velocity change  H^-1 J^T lambda: from the rows Y = H^-1 J^T kept since the Delassus matrix was built (limit rows: columns of H^-1).
  // (With H^-1 in LDS a variant that rebuilt J^T lambda from the contact points and multiplied by H^-1 afterwards -- nothing but H^-1
  // live across the pivoting loops -- was built as well: it saved no scratch (416 B against 368 B this way) and its gfx950 build
  // returned wrong states for every lane with a contact although either half of it alone is bitwise right; rebuilding Y itself
  // after the loops leaves 320 B: the rows are not what spills.  DESIGN.md section 4.1.)
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    Real dv = Real(0);
    sfor<0, NCA>([&](auto S) { constexpr int cs = S; dv += Yn[cs][i] * x[2 * cs] + Yt[cs][i] * x[2 * cs + 1]; });
    sfor<0, NL>([&](auto K) {
      constexpr int k = K;
      if constexpr (T::limited(k)) dv += Hv(tri(rev<N>(i), rev<N>(2 + k))) * xl[lim_ord<T>(k)];
    });
    vs[i] += dv;
  });
}

// ------------------------------------------------------------------ single-lane fallback: any number of contacts, loops over LDS
// Executed by ONE lane at a time (the others of its wave wait) for an env with more touching capsules than the register
// tiers hold -- the robot lying on the floor.  Same LCP, same two-stage pivoting with the same start sets and tolerances
// as blcp_bpp, written as plain loops over arrays in `mem` (slow_words<T>() Reals of LDS).  Speed is irrelevant here.
template <class Real>
__device__ inline void slow_masked_solve(int m, const Real* A, uint32_t F, Real* x, Real* L, Real* invd, Real* W) {
  for (int j = 0; j < m; j++) {
    const Real fj0 = ((F >> j) & 1u) ? Real(1) : Real(0);
    Real d = fj0 * A[tri(j, j)] + (Real(1) - fj0);
    for (int k = 0; k < j; k++) { W[k] = L[tri(j, k)] * L[tri(k, k)]; d -= L[tri(j, k)] * W[k]; }
    L[tri(j, j)] = d;
    invd[j] = rcp_<Real>(d);
    const Real fj = fj0 * invd[j];
    for (int i = j + 1; i < m; i++) {
      Real t = ((F >> i) & 1u) ? A[tri(i, j)] : Real(0);
      for (int k = 0; k < j; k++) t -= L[tri(i, k)] * W[k];
      L[tri(i, j)] = t * fj;
    }
  }
  for (int i = 0; i < m; i++) for (int k = 0; k < i; k++) x[i] -= L[tri(i, k)] * x[k];
  for (int i = 0; i < m; i++) x[i] *= invd[i];
  for (int i = m - 1; i >= 0; i--) for (int k = i + 1; k < m; k++) x[i] -= L[tri(k, i)] * x[k];
}
template <class Real>
__device__ inline void slow_blcp(int m, const Real* A, const Real* b, const Real* lo, const Real* hi, uint32_t pinmask, uint32_t& F,
                                 uint32_t& U, Real* x, int max_iter, bool zero_bounds, Real* L, Real* invd, Real* W, Real* r, Real* xb) {
  Real bmax = Real(0);
  for (int i = 0; i < m; i++) bmax = fmax(bmax, fabs(b[i]));
  const Real tol = tol_<Real>() * (Real(1) + bmax);
  int best = m + 1, patience = 3;
  bool conv = false;
  for (int it = 0; it < max_iter && !conv; ++it) {
    for (int i = 0; i < m; i++) {
      const bool f = (F >> i) & 1u, u = (U >> i) & 1u;
      xb[i] = f ? Real(0) : (u ? hi[i] : lo[i]);
    }
    for (int i = 0; i < m; i++) {
      Real t = b[i];
      if (!zero_bounds) for (int j = 0; j < m; j++) t -= A[tri(i, j)] * xb[j];
      r[i] = ((F >> i) & 1u) ? t : xb[i];
    }
    slow_masked_solve<Real>(m, A, F, r, L, invd, W);
    uint32_t B = 0, GT = 0;
    for (int i = 0; i < m; i++) {
      Real w = -b[i];
      for (int j = 0; j < m; j++) w += A[tri(i, j)] * r[j];
      const bool f = (F >> i) & 1u, u = (U >> i) & 1u, pinned = (pinmask >> i) & 1u;
      const bool over = r[i] > hi[i] + tol * (Real(1) + fabs(hi[i]));
      const bool under = r[i] < lo[i] - tol * (Real(1) + fabs(lo[i]));
      const bool wbad = u ? (w > tol) : (w < -tol);
      const bool inf = f ? (over || under) : (wbad && !pinned);
      B |= inf ? (1u << i) : 0u;
      GT |= (r[i] > hi[i]) ? (1u << i) : 0u;
    }
    for (int i = 0; i < m; i++) x[i] = r[i];
    conv = (B == 0u);
    if (conv) break;
    const int ninf = __popc(B);
    const bool improved = ninf < best;
    const bool single = !improved && patience == 0;
    best = improved ? ninf : best;
    patience = improved ? 3 : (patience > 0 ? patience - 1 : 0);
    const uint32_t Bs = single ? (1u << (31 - __clz((int)B))) : B;
    const uint32_t toBound = Bs & F, toFree = Bs & ~F;
    F = (F & ~toBound) | toFree;
    U = (U & ~(toFree | toBound)) | (toBound & GT);
  }
  for (int i = 0; i < m; i++) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
}

template <class Real, class T, class PT, bool EXTRAS>
__device__ __attribute__((noinline)) void slow_constraints(const PT& P, Real* mem, Real qx, Real qy, const ReportTo<Real>& rp) {
  constexpr int NL = T::NL, N = T::NDOF, NC = T::NC, MM = max_rows<T>();
  // layout written by the caller: Hi[N*N], px[NL], py[NL], sg[NL], vs[N], con[NC], cPx[NC], cPy[NC], cdep[NC], lim[NL] (-1 low, +1 up, 0),
  // viol[NL], clk[NL] scratch
  Real* Hi = mem; Real* px = Hi + N * N; Real* py = px + NL; Real* sg = py + NL; Real* vs = sg + NL;
  Real* con = vs + N; Real* cPx = con + NC; Real* cPy = cPx + NC; Real* cdep = cPy + NC;
  Real* lim = cdep + NC; Real* viol = lim + NL; Real* spare = viol + NL;
  Real* J = spare + NL; Real* Y = J + MM * N; Real* A = Y + MM * N; Real* L = A + MM * (MM + 1) / 2;
  Real* b = L + MM * (MM + 1) / 2; Real* lo = b + MM; Real* hi = lo + MM; Real* x = hi + MM; Real* r = x + MM; Real* xb = r + MM;
  Real* invd = xb + MM; Real* W = invd + MM;
  int m = 0, ncont = 0;
  uint32_t limrows = 0;
  for (int c = 0; c < NC; c++) {
    if (con[c] == Real(0)) continue;
    const uint32_t am = anc_mask_of_capsule<T>(c);
    Real* jn = J + m * N; Real* jt = J + (m + 1) * N;
    jn[0] = Real(0); jn[1] = Real(1); jt[0] = Real(-1); jt[1] = Real(0);
    for (int j = 0; j < NL; j++) {
      const bool a = (am >> j) & 1u;
      jn[2 + j] = a ? sg[j] * (cPx[c] - px[j]) : Real(0);
      jt[2 + j] = a ? sg[j] * (cPy[c] - py[j]) : Real(0);
    }
    Real rn = Real(0), rt = Real(0);
    for (int i = 0; i < N; i++) { rn += jn[i] * vs[i]; rt += jt[i] * vs[i]; }
    b[m] = fmin(cdep[c] * P.erp_dt, P.max_erv) - rn; lo[m] = Real(0); hi[m] = inf_<Real>();
    b[m + 1] = -rt; lo[m + 1] = Real(0); hi[m + 1] = Real(0);
    m += 2; ncont++;
  }
  for (int k = 0; k < NL; k++) {
    if (lim[k] == Real(0)) continue;
    Real* jl = J + m * N;
    for (int i = 0; i < N; i++) jl[i] = Real(0);
    jl[2 + k] = Real(1);
    const bool low = lim[k] < Real(0);
    b[m] = fmin(fmax(-viol[k] * P.limit_erp_dt, -P.max_erv), P.max_erv) - vs[2 + k];
    lo[m] = low ? Real(0) : -inf_<Real>(); hi[m] = low ? inf_<Real>() : Real(0);
    limrows |= 1u << m;
    m++;
  }
  if (m == 0) return;
  for (int rr = 0; rr < m; rr++)
    for (int i = 0; i < N; i++) {
      Real t = Real(0);
      for (int j = 0; j < N; j++) t += Hi[i * N + j] * J[rr * N + j];
      Y[rr * N + i] = t;
    }
  for (int rr = 0; rr < m; rr++)
    for (int cc = 0; cc <= rr; cc++) {
      Real t = Real(0);
      for (int i = 0; i < N; i++) t += J[rr * N + i] * Y[cc * N + i];
      if (rr == cc) t *= ((limrows >> rr) & 1u) ? P.cfm1 : P.ccfm1;
      A[tri(rr, cc)] = t;
    }
  uint32_t pinmask = 0, F = 0, U = 0;
  Real bmax0 = Real(0);
  for (int i = 0; i < m; i++) bmax0 = fmax(bmax0, fabs(b[i]));
  const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
  for (int i = 0; i < m; i++) {
    x[i] = Real(0);
    const bool pinned = !(lo[i] < hi[i]);
    const bool upper = !(lo[i] == Real(0));
    const bool start_free = !pinned && (upper ? (b[i] < -tol0) : (b[i] > tol0));
    pinmask |= pinned ? (1u << i) : 0u;
    F |= start_free ? (1u << i) : 0u;
    U |= (upper && !start_free) ? (1u << i) : 0u;
  }
  slow_blcp<Real>(m, A, b, lo, hi, pinmask, F, U, x, 4 * P.iters1 + 64, true, L, invd, W, r, xb);
  if (ncont > 0) {
    for (int c = 0; c < ncont; c++) {
      const int sn = 2 * c, stt = 2 * c + 1;
      const Real hb = fabs(P.mu * x[sn]);
      hi[stt] = hb; lo[stt] = -hb;
      const bool pinned = !(hb > Real(0));
      pinmask = pinned ? (pinmask | (1u << stt)) : (pinmask & ~(1u << stt));
      F = pinned ? (F & ~(1u << stt)) : (F | (1u << stt));
      U &= ~(1u << stt);
    }
    slow_blcp<Real>(m, A, b, lo, hi, pinmask, F, U, x, 4 * P.iters2 + 64, false, L, invd, W, r, xb);
  }
  for (int i = 0; i < N; i++) {
    Real dv = Real(0);
    for (int rr = 0; rr < m; rr++) dv += Y[rr * N + i] * x[rr];
    vs[i] += dv;
  }
  if (EXTRAS && rp.rec != nullptr) {
    const Real idt = Real(1) / P.dt;
    int r = 0;
    for (int c = 0; c < NC; c++) {
      if (con[c] == Real(0)) continue;
      Real* o = rp.rec + 8 * r;
      o[0] = (Real)P.cbody[c]; o[1] = Real(-1);
      o[2] = P.root_x0 + qx + cPx[c]; o[3] = P.root_y0 + qy + cPy[c]; o[4] = Real(0);
      o[5] = -x[2 * r + 1] * idt; o[6] = x[2 * r] * idt; o[7] = Real(0);
      r++;
    }
    *rp.count = r;
    for (int i = 0; i < N; i++) {
      Real f = Real(0);
      for (int rr = 0; rr < m; rr++) f += J[rr * N + i] * x[rr];
      rp.cf[i] = f * idt;
    }
  }
}

#ifdef DART_WAVE_COOP
// ------------------------------------------------------------------ the same fallback, served by the WHOLE wave (device build)
// One env at a time, as above, but the 64 lanes share its rows: lane c builds candidate c's Jacobian rows, lane pairs build Y = H^-1 J^T
// and the Delassus matrix, and the two pivoting solves run in registers with lane i holding row i (wave_blcp.hpp: the tree kernel's
// solver, same start sets, tolerances, patience and single-pivot rule as slow_blcp / blcp_bpp).  Same memory layout as slow_constraints;
// `owner` is the lane whose env this is (it wrote the inputs and takes vs back; only its `rp` is used).  A solve that reaches its
// iteration cap (coop_iters: min of the stage's cap and DART_COOP_BUDGET, the env's own) ends on its last iterate, clamped into the box,
// as a lane of a register tier does at its cap.
// Measured need (DartHalfCheetah-v1, 65 536 envs): 6e-5 of the env-world-steps have five touching capsules -- ~20 per launch -- and a
// launch takes as long as its slowest wave: served by one lane (~1 M cycles each) they set the kernel time, 2.5 ms instead of 0.4.
template <class Real, class T, class PT, bool EXTRAS>
__device__ __attribute__((noinline)) void wave_constraints(const PT& P, Real* mem, Real qx, Real qy, const ReportTo<Real>& rp, int owner,
                                                           int& budget) {
  constexpr int NL = T::NL, N = T::NDOF, NC = T::NC, MM = max_rows<T>();
  static_assert(MM <= 24 && NC <= 32, "wave_constraints: row capacity of the register solver variants used here");
  const int lane = (int)(threadIdx.x & 63);
  Real* Hi = mem; Real* px = Hi + N * N; Real* py = px + NL; Real* sg = py + NL; Real* vs = sg + NL;
  Real* con = vs + N; Real* cPx = con + NC; Real* cPy = cPx + NC; Real* cdep = cPy + NC;
  Real* lim = cdep + NC; Real* viol = lim + NL; Real* spare = viol + NL;
  Real* J = spare + NL; Real* Y = J + MM * N; Real* A = Y + MM * N;
  Real* b = A + 2 * (MM * (MM + 1) / 2); Real* lo = b + MM; Real* hi = lo + MM; Real* x = hi + MM;   // (the single-lane solver's factor sits between A and b)
  __syncthreads();   // the owner's inputs are in place
#ifdef DART_WAVE_TIMING_FALLBACK   // measurement build: [24] rows + Y + A, [25] stage 1, [26] stage 2, [27] tail, [28] calls, [29] rows m summed
  long long tf0 = (long long)__builtin_readcyclecounter();
  auto tf_mark = [&](int slot) { const long long t = (long long)__builtin_readcyclecounter(); if (P.stats && lane == 0) atomicAdd(&P.stats[slot], (unsigned long long)(t - tf0)); tf0 = t; };
#endif
  // ---- rows: lane c < NC owns candidate capsule c, lane NC + k owns the limit of link k
  const bool cact = lane < NC && con[lane < NC ? lane : 0] != Real(0);
  const unsigned long long cbal = __ballot(cact);
  const int ncont = __popcll(cbal);
  const int kl = lane - NC;
  const bool lact = kl >= 0 && kl < NL && lim[(kl >= 0 && kl < NL) ? kl : 0] != Real(0);
  const unsigned long long lbal = __ballot(lact);
  const int m = 2 * ncont + __popcll(lbal);
  if (m == 0) return;
  const unsigned long long below = (1ull << lane) - 1ull;
  if (cact) {
    const int c = lane, row = 2 * __popcll(cbal & below);
    const uint32_t am = anc_mask_of_capsule<T>(c);
    Real* jn = J + row * N; Real* jt = J + (row + 1) * N;
    jn[0] = Real(0); jn[1] = Real(1); jt[0] = Real(-1); jt[1] = Real(0);
    Real rn = vs[1], rt = -vs[0];
    for (int j = 0; j < NL; j++) {
      const bool a = (am >> j) & 1u;
      const Real vn = a ? sg[j] * (cPx[c] - px[j]) : Real(0), vt = a ? sg[j] * (cPy[c] - py[j]) : Real(0);
      jn[2 + j] = vn; jt[2 + j] = vt;
      rn += vn * vs[2 + j]; rt += vt * vs[2 + j];
    }
    b[row] = fmin(cdep[c] * P.erp_dt, P.max_erv) - rn; lo[row] = Real(0); hi[row] = inf_<Real>();
    b[row + 1] = -rt; lo[row + 1] = Real(0); hi[row + 1] = Real(0);
    x[row] = Real(0); x[row + 1] = Real(0);
  }
  if (lact) {
    const int row = 2 * ncont + __popcll(lbal & below);
    Real* jl = J + row * N;
    for (int i = 0; i < N; i++) jl[i] = (i == 2 + kl) ? Real(1) : Real(0);
    const bool low = lim[kl] < Real(0);
    b[row] = fmin(fmax(-viol[kl] * P.limit_erp_dt, -P.max_erv), P.max_erv) - vs[2 + kl];
    lo[row] = low ? Real(0) : -inf_<Real>(); hi[row] = low ? inf_<Real>() : Real(0);
    x[row] = Real(0);
  }
  __syncthreads();
  // ---- Y = H^-1 J^T (m x N entries) and the Delassus matrix (packed lower triangle), one entry per lane and round
  for (int p = lane; p < m * N; p += 64) {
    const int rr = p / N, i = p - rr * N;
    Real t = Real(0);
    for (int j = 0; j < N; j++) t += Hi[i * N + j] * J[rr * N + j];
    Y[p] = t;
  }
  __syncthreads();
  for (int p = lane; p < m * (m + 1) / 2; p += 64) {
    int rr = 0;
    while ((rr + 1) * (rr + 2) / 2 <= p) rr++;
    const int cc = p - rr * (rr + 1) / 2;
    Real t = Real(0);
    for (int i = 0; i < N; i++) t += J[rr * N + i] * Y[cc * N + i];
    if (rr == cc) t *= (rr >= 2 * ncont) ? P.cfm1 : P.ccfm1;
    A[p] = t;
  }
  __syncthreads();
  // ---- start sets (lane i = row i), stage 1, friction bounds, stage 2
  const bool row = lane < m;
  const int ri = row ? lane : 0;
  Real bmax0 = row ? fabs(b[ri]) : Real(0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bmax0 = fmax(bmax0, __shfl_xor(bmax0, o));
  const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
  const bool pinned0 = row && !(lo[ri] < hi[ri]);
  const bool upper = row && !(lo[ri] == Real(0));
  const bool start_free = row && !pinned0 && (upper ? (b[ri] < -tol0) : (b[ri] > tol0));
  uint64_t pinmask = __ballot(pinned0), F = __ballot(start_free), U = __ballot(upper && !start_free);
  auto solve = [&](uint64_t& Fs, uint64_t& Us, int cap, bool zero_bounds) {
    const int mi = coop_iters(cap, budget);
    const BlcpSets res = (m <= 16) ? sp_blcp_t<Real, 16, true>(A, b, lo, hi, x, m, pinmask, Fs, Us, mi, nullptr, lane, zero_bounds, Real(0), true)
                                   : sp_blcp_t<Real, 24, true>(A, b, lo, hi, x, m, pinmask, Fs, Us, mi, nullptr, lane, zero_bounds, Real(0), true);
    Fs = res.F; Us = res.U;   // (cap or budget reached: x holds the last iterate, clamped into the box)
    budget -= res.iters;
    __syncthreads();
  };
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(24);
#endif
  solve(F, U, P.iters1, true);
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(25);
#endif
  if (ncont > 0) {
    const bool fr = row && lane < 2 * ncont && (lane & 1);
    const Real hb = fr ? fabs(P.mu * x[fr ? lane - 1 : 0]) : Real(0);
    if (fr) { hi[lane] = hb; lo[lane] = -hb; }
    const bool fpin = fr && !(hb > Real(0));
    const uint64_t frm = __ballot(fr), fpm = __ballot(fpin);
    pinmask = (pinmask & ~frm) | fpm;
    F = (F & ~frm) | (frm & ~fpm);
    U &= ~frm;
    __syncthreads();
    solve(F, U, P.iters2, false);
  }
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(26);
#endif
  if (lane < N) {
    Real dv = Real(0);
    for (int rr = 0; rr < m; rr++) dv += Y[rr * N + lane] * x[rr];
    vs[lane] += dv;
  }
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(27);
  if (P.stats && lane == 0) { atomicAdd(&P.stats[28], 1ull); atomicAdd(&P.stats[29], (unsigned long long)m); }
#endif
  if (EXTRAS && lane == owner && rp.rec != nullptr) {
    const Real idt = Real(1) / P.dt;
    int rc = 0;
    for (int c = 0; c < NC; c++) {
      if (con[c] == Real(0)) continue;
      Real* o = rp.rec + 8 * rc;
      o[0] = (Real)P.cbody[c]; o[1] = Real(-1);
      o[2] = P.root_x0 + qx + cPx[c]; o[3] = P.root_y0 + qy + cPy[c]; o[4] = Real(0);
      o[5] = -x[2 * rc + 1] * idt; o[6] = x[2 * rc] * idt; o[7] = Real(0);
      rc++;
    }
    *rp.count = rc;
    for (int i = 0; i < N; i++) {
      Real f = Real(0);
      for (int rr = 0; rr < m; rr++) f += J[rr * N + i] * x[rr];
      rp.cf[i] = f * idt;
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------ FOUR such envs at a time, one per row of 16 lanes (round 5)
// An env beyond the small tier has 6-12 rows; served one at a time it keeps 16 lanes of the wave busy at best and costs ~80 k cycles, and
// the half cheetah has ~3 of them per wave and world step -- at one wave per SIMD that latency IS the kernel's time (no other wave to hide it).
// Here group g (lanes 16 g ... 16 g + 15) serves one env out of its own LDS block `gblk` (coop4_words<T>() Reals, inputs staged by the owners
// as for wave_constraints), row-per-lane from the start: lane l < NC owns candidate capsule l, lane NC + k link k's limit (NC + NL <= 16),
// then lane r holds constraint row r -- its Jacobian row in registers, Y_r = H^-1 J_r^T (N x N FMAs), the Delassus row A_r,c (c <= r) from the
// other lanes' Y rows in LDS -- and the two pivoting stages run in sp_blcp4_t (wave_blcp.hpp).  Everything a group computes is a function of
// its own block: an env's result does not depend on the group it lands in nor on the other three envs (the row-of-16 DPP operations are the
// only cross-lane traffic).  Envs with more than 16 rows (>= 6 contacts on the half cheetah) stay with wave_constraints.
// `oblk`: the block THIS lane's own env was staged in (an owner in this pass), else null -- only the owner reports (EXTRAS).
template <class Real, class T, class PT, bool EXTRAS>
__device__ __attribute__((noinline)) void wave_constraints4(const PT& P, Real* gblk, Real* oblk, Real qx, Real qy, const ReportTo<Real>& rp, int& budget) {
  constexpr int NL = T::NL, N = T::NDOF, NC = T::NC, M = 16;
  static_assert(coop4_fits<T>(), "wave_constraints4: one lane per candidate capsule and per limit inside a row of 16 lanes");
  const int lane = (int)(threadIdx.x & 63), l = lane & 15;
  // (the blocks are LDS: re-typed, or every access below compiles to flat_load / flat_store -- a lane-varying generic pointer in a real call)
  const auto Hi = DART_LDS_PTR(Real, gblk); const auto px = Hi + N * N; const auto py = px + NL; const auto sg = py + NL; const auto vs = sg + NL;
  const auto con = vs + N; const auto cPx = con + NC; const auto cPy = cPx + NC; const auto cdep = cPy + NC;
  const auto lim = cdep + NC; const auto viol = lim + NL;
  const auto J = viol + NL; const auto Y = J + M * N; const auto A = Y + M * N;
  const auto b = A + M * (M + 1) / 2; const auto lo = b + M; const auto hi = lo + M; const auto x = hi + M;
  Real* const gA = gblk + (N * N + 3 * NL + N + 4 * NC + 2 * NL + 2 * M * N);   // the solver takes generic pointers (and re-types them itself)
  __syncthreads();   // the owners' inputs are in place (idle groups: con = lim = 0)
#ifdef DART_WAVE_TIMING_FALLBACK   // measurement build: [24] rows + Y + A, [25] stage 1, [26] stage 2, [27] tail, [28] passes, [29] rows m summed over the groups
  long long tf0 = (long long)__builtin_readcyclecounter();
  auto tf_mark = [&](int slot) { const long long t = (long long)__builtin_readcyclecounter(); if (P.stats && lane == 0) atomicAdd(&P.stats[slot], (unsigned long long)(t - tf0)); tf0 = t; };
#endif
  const bool cact = l < NC && con[l < NC ? l : 0] != Real(0);
  const uint32_t cbal = row_ballot(cact, lane);
  const int ncont = __popc(cbal);
  const int kl = l - NC;
  const bool lact = kl >= 0 && kl < NL && lim[(kl >= 0 && kl < NL) ? kl : 0] != Real(0);
  const uint32_t lbal = row_ballot(lact, lane);
  const int m = 2 * ncont + __popc(lbal);   // (the caller admits m <= 16)
  if (__ballot(m > 0) == 0ull) return;
  const uint32_t below = (1u << l) - 1u;
  if (cact) {
    const int c = l, row = 2 * __popc(cbal & below);
    // (the compile-time tables, whatever ANC_TABLES_RT says for the topology's step kernel: evaluated at run time the ancestor table is a
    // walk over the parent array in constant memory, one dependent s_load per hop -- 14 such loops at the top of this function, found in
    // the disassembly.  ADVICE r5 asked why that is safe where ANC_TABLES_RT exists because compile-time tables once tripped the toolchain's
    // EXEC-prologue defect: the defect is a property of a COMPILATION, not of the construct -- every build's assembly of this function is
    // linted (tools/exec_prologue_lint.py: 0 hits in 94 functions of each planar unit), and the poison test holds the half cheetah's kernels,
    // which run this path for 5 % of their env-steps, and the physics-only walker / cheetah trees to identical bits with registers, scratch
    // and LDS disturbed, fp64 and fp32 (tests/test_gpu_first_launch.py; tests/test_gpu_spatial.py's floor batches run it for every lane).)
    const uint32_t am = (uint32_t)((AncTable<T>::value >> (8 * (int)((ClinkTable<T>::value >> (4 * c)) & 0xfull))) & 0xffull);
    const auto jn = J + row * N; const auto jt = J + (row + 1) * N;
    jn[0] = Real(0); jn[1] = Real(1); jt[0] = Real(-1); jt[1] = Real(0);
    Real rn = vs[1], rt = -vs[0];
    for (int j = 0; j < NL; j++) {
      const bool a = (am >> j) & 1u;
      const Real vn = a ? sg[j] * (cPx[c] - px[j]) : Real(0), vt = a ? sg[j] * (cPy[c] - py[j]) : Real(0);
      jn[2 + j] = vn; jt[2 + j] = vt;
      rn += vn * vs[2 + j]; rt += vt * vs[2 + j];
    }
    b[row] = fmin(cdep[c] * P.erp_dt, P.max_erv) - rn; lo[row] = Real(0); hi[row] = inf_<Real>();
    b[row + 1] = -rt; lo[row + 1] = Real(0); hi[row + 1] = Real(0);
    x[row] = Real(0); x[row + 1] = Real(0);
  }
  if (lact) {
    const int row = 2 * ncont + __popc(lbal & below);
    const auto jl = J + row * N;
    for (int i = 0; i < N; i++) jl[i] = (i == 2 + kl) ? Real(1) : Real(0);
    const bool low = lim[kl] < Real(0);
    b[row] = fmin(fmax(-viol[kl] * P.limit_erp_dt, -P.max_erv), P.max_erv) - vs[2 + kl];
    lo[row] = low ? Real(0) : -inf_<Real>(); hi[row] = low ? inf_<Real>() : Real(0);
    x[row] = Real(0);
  }
  __syncthreads();
  // ---- lane r = row r: J_r in registers, Y_r = H^-1 J_r^T, then A_r,c = J_r . Y_c for c <= r (the lower triangle, as wave_constraints builds it)
  const bool row = l < m;
  const int ri = row ? l : 0;
  Real Jr[N];
  sfor<0, N>([&](auto I) { Jr[I] = row ? J[ri * N + I] : Real(0); });
  if (row) {
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real t = Real(0);
      sfor<0, N>([&](auto Jc) { constexpr int j = Jc; t += Hi[i * N + j] * Jr[j]; });
      Y[ri * N + i] = t;
    });
  }
  __syncthreads();
  if (row) {
    for (int cc = 0; cc <= ri; cc++) {
      Real t = Real(0);
      sfor<0, N>([&](auto I) { t += Jr[I] * Y[cc * N + I]; });
      if (cc == ri) t *= (ri >= 2 * ncont) ? P.cfm1 : P.ccfm1;
      A[TI(ri, cc)] = t;
    }
  }
  __syncthreads();
  // ---- start sets, stage 1, friction bounds, stage 2 (per group: the lane kernels' rules, as in wave_constraints)
  const Real bmax0 = row_max_nonneg<Real>(row ? fabs(b[ri]) : Real(0));
  const Real tol0 = tol_<Real>() * (Real(1) + bmax0);
  const bool pinned0 = row && !(lo[ri] < hi[ri]);
  const bool upper = row && !(lo[ri] == Real(0));
  const bool start_free = row && !pinned0 && (upper ? (b[ri] < -tol0) : (b[ri] > tol0));
  uint32_t pinmask = row_ballot(pinned0, lane), F = row_ballot(start_free, lane), U = row_ballot(upper && !start_free, lane);
  auto solve = [&](int cap, bool zero_bounds) {
    const Blcp4Sets res = sp_blcp4_t<Real>(gA, gA + M * (M + 1) / 2, gA + M * (M + 1) / 2 + M, gA + M * (M + 1) / 2 + 2 * M, gA + M * (M + 1) / 2 + 3 * M, m, pinmask,
                                           F, U, coop_iters(cap, budget), lane, zero_bounds, true);
    F = res.F; U = res.U;   // (cap or budget reached: x holds the last iterate, clamped into the box)
    budget -= res.iters;
    __syncthreads();
  };
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(24);
#endif
  solve(P.iters1, true);
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(25);
#endif
  if (__ballot(ncont > 0) != 0ull) {
    const bool fr = row && l < 2 * ncont && (l & 1);
    const Real hb = fr ? fabs(P.mu * x[fr ? l - 1 : 0]) : Real(0);
    if (fr) { hi[l] = hb; lo[l] = -hb; }
    const bool fpin = fr && !(hb > Real(0));
    const uint32_t frm = row_ballot(fr, lane), fpm = row_ballot(fpin, lane);
    pinmask = (pinmask & ~frm) | fpm;
    F = (F & ~frm) | (frm & ~fpm);
    U &= ~frm;
    __syncthreads();
    solve(P.iters2, false);   // (a group without contacts repeats its converged stage-1 solve on the same sets: the same x)
  }
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(26);
#endif
  if (l < N) {
    Real dv = Real(0);
    for (int rr = 0; rr < m; rr++) dv += Y[rr * N + l] * x[rr];
    vs[l] += dv;
  }
  __syncthreads();
#ifdef DART_WAVE_TIMING_FALLBACK
  tf_mark(27);
  if (P.stats && l == 0 && m > 0) { atomicAdd(&P.stats[29], (unsigned long long)m); if (lane == 0) atomicAdd(&P.stats[28], 1ull); }
#endif
  if (EXTRAS && oblk != nullptr && rp.rec != nullptr) {   // the owner reports its env (from the block it was served in)
    const auto ocon = DART_LDS_PTR(const Real, oblk) + N * N + 3 * NL + N; const auto ocPx = ocon + NC; const auto ocPy = ocPx + NC;
    const auto oJ = DART_LDS_PTR(const Real, oblk) + N * N + 3 * NL + N + 4 * NC + 2 * NL; const auto ox = oJ + 2 * M * N + M * (M + 1) / 2 + 3 * M;
    const Real idt = Real(1) / P.dt;
    int rc = 0, om = 0;
    for (int c = 0; c < NC; c++) {
      if (ocon[c] == Real(0)) continue;
      Real* o = rp.rec + 8 * rc;
      o[0] = (Real)P.cbody[c]; o[1] = Real(-1);
      o[2] = P.root_x0 + qx + ocPx[c]; o[3] = P.root_y0 + qy + ocPy[c]; o[4] = Real(0);
      o[5] = -ox[2 * rc + 1] * idt; o[6] = ox[2 * rc] * idt; o[7] = Real(0);
      rc++;
    }
    *rp.count = rc;
    const auto olim = ocon + 4 * NC;
    om = 2 * rc;
    for (int k = 0; k < NL; k++) om += olim[k] != Real(0) ? 1 : 0;
    for (int i = 0; i < N; i++) {
      Real f = Real(0);
      for (int rr = 0; rr < om; rr++) f += oJ[rr * N + i] * ox[rr];
      rp.cf[i] = f * idt;
    }
  }
  __syncthreads();
}
#endif

// ------------------------------------------------------------------ one World::step (dt) for one env
template <class Real, class T, class PT, bool EXTRAS>
__device__ __forceinline__ void world_step(const PT& P, Real (&q)[T::NDOF], Real (&dq)[T::NDOF],
                                           const Real (&tau)[T::NDOF], WarmSets& warm, Real* slow_mem, int64_t env,
                                           const ReportTo<Real>& rp, Real* hl = nullptr) {
  constexpr int NL = T::NL, N = T::NDOF, NC = T::NC;
  Real c[NL], s[NL], px[NL], py[NL], lx[NL], ly[NL], om[NL];
  Real apx[NL], apy[NL];
  // composite-body quantities, each expressed about the link's OWN joint origin (no large-offset cancellation in
  // fp32): mass, first moment d = sum m (r - p_k), inertia Ip about p_k, force F and moment Nz about p_k
  Real mc[NL], dcx[NL], dcy[NL], Ip[NL], Fx[NL], Fy[NL], Nz[NL];
  // ---- forward pass: kinematics, velocity-product accelerations, per-link wrench
  sfor<0, NL>([&](auto K) {
    constexpr int k = K;
    Real sj, cj;
    sincos_<Real>(q[2 + k], sj, cj);
    sj *= P.sigma[k];
    if constexpr (k == 0) {
      c[0] = cj; s[0] = sj; px[0] = Real(0); py[0] = Real(0); om[0] = P.sigma[0] * dq[2];
      lx[0] = Real(0); ly[0] = Real(0);
      apx[0] = Real(0); apy[0] = Real(0);
    } else {
      constexpr int p = T::parent(k);
      c[k] = c[p] * cj - s[p] * sj;
      s[k] = s[p] * cj + c[p] * sj;
      if constexpr (DART_ZERO(PT, jx, k)) { lx[k] = -s[p] * P.jy[k]; ly[k] = c[p] * P.jy[k]; }
      else if constexpr (DART_ZERO(PT, jy, k)) { lx[k] = c[p] * P.jx[k]; ly[k] = s[p] * P.jx[k]; }
      else { lx[k] = c[p] * P.jx[k] - s[p] * P.jy[k]; ly[k] = s[p] * P.jx[k] + c[p] * P.jy[k]; }
      px[k] = px[p] + lx[k]; py[k] = py[p] + ly[k];
      om[k] = om[p] + P.sigma[k] * dq[2 + k];
      Real w2 = om[p] * om[p];
      apx[k] = apx[p] - w2 * lx[k]; apy[k] = apy[p] - w2 * ly[k];
    }
    mc[k] = P.mass[k];
    if constexpr (DART_ZERO(PT, cx, k) && DART_ZERO(PT, cy, k)) {   // COM on the joint axis
      Fx[k] = P.mass[k] * apx[k]; Fy[k] = P.mass[k] * (apy[k] + P.g);
      dcx[k] = Real(0); dcy[k] = Real(0); Ip[k] = P.izz[k]; Nz[k] = Real(0);
    } else {
      Real ox, oy;
      if constexpr (DART_ZERO(PT, cx, k)) { ox = -s[k] * P.cy[k]; oy = c[k] * P.cy[k]; }
      else if constexpr (DART_ZERO(PT, cy, k)) { ox = c[k] * P.cx[k]; oy = s[k] * P.cx[k]; }
      else { ox = c[k] * P.cx[k] - s[k] * P.cy[k]; oy = s[k] * P.cx[k] + c[k] * P.cy[k]; }
      Real w2 = om[k] * om[k];
      Real fx = P.mass[k] * (apx[k] - w2 * ox), fy = P.mass[k] * (apy[k] - w2 * oy + P.g);
      dcx[k] = P.mass[k] * ox; dcy[k] = P.mass[k] * oy;
      Ip[k] = P.izz[k] + P.mass[k] * (ox * ox + oy * oy);
      Fx[k] = fx; Fy[k] = fy; Nz[k] = ox * fy - oy * fx;
    }
  });
  if constexpr (topo_fluid<T>::value) {
    // Fluid model of snake_7link.py:37-47: every body is pushed by -k (v_com . n) n at its frame origin (= its joint origin), n =
    // the body's local z axis, which lies in the plane of motion; the +-0.05 (w x n) terms of the reference drop out of the dot
    // product.  The carrier of the second root translation takes the same force: a drag on that translation alone.
    Real vpx[NL], vpy[NL];   // velocity of the link's joint origin
    sfor<0, NL>([&](auto K) {
      constexpr int k = K;
      if constexpr (k == 0) { vpx[0] = dq[0]; vpy[0] = dq[1]; }
      else { constexpr int p = T::parent(k); vpx[k] = vpx[p] - om[p] * ly[k]; vpy[k] = vpy[p] + om[p] * lx[k]; }
      const Real ox = c[k] * P.cx[k] - s[k] * P.cy[k], oy = s[k] * P.cx[k] + c[k] * P.cy[k];
      const Real vn = -(vpx[k] - om[k] * oy) * s[k] + (vpy[k] + om[k] * ox) * c[k];
      const Real f = -P.fluid_k * vn;
      Fx[k] += f * s[k]; Fy[k] -= f * c[k];    // F holds inertial minus applied force; the force is f * n, n = (-s, c)
    });
  }
  // ---- backward pass: fold each composite into its parent, shifting the reference point by the link vector
  sfor_rev<1, NL>([&](auto K) {
    constexpr int k = K, p = T::parent(k);
    Ip[p] += Ip[k] + Real(2) * (lx[k] * dcx[k] + ly[k] * dcy[k]) + mc[k] * (lx[k] * lx[k] + ly[k] * ly[k]);
    dcx[p] += dcx[k] + mc[k] * lx[k]; dcy[p] += dcy[k] + mc[k] * ly[k];
    mc[p] += mc[k];
    Nz[p] += Nz[k] + (lx[k] * Fy[k] - ly[k] * Fx[k]);
    Fx[p] += Fx[k]; Fy[p] += Fy[k];
  });
  // ---- H = M + dt D, rhs = tau - C - D dq.  Stored with REVERSED dof order (rev<N>(i) = N-1-i) so that the LDL^T below
  // eliminates leaf joints first and the floating base last (Featherstone's LTDL order = articulated-body
  // recursion numerically): pivots are articulated inertias instead of small differences of large numbers.
  Real H[N * (N + 1) / 2], rhs[N];
  H[tri(rev<N>(0), rev<N>(0))] = mc[0]; H[tri(rev<N>(1), rev<N>(0))] = Real(0); H[tri(rev<N>(1), rev<N>(1))] = mc[0];
  rhs[0] = tau[0] - Fx[0];
  rhs[1] = tau[1] - Fy[0];
  if constexpr (topo_fluid<T>::value) rhs[1] -= P.fluid_k * dq[1];
  if constexpr (!DART_ZERO(PT, damp, 0)) rhs[0] -= P.damp[0] * dq[0];
  if constexpr (!DART_ZERO(PT, damp, 1)) rhs[1] -= P.damp[1] * dq[1];
  if constexpr (!DART_ZERO(PT, stiff, 0)) rhs[0] -= P.stiff[0] * (q[0] + P.dt * dq[0] - P.rest[0]);
  if constexpr (!DART_ZERO(PT, stiff, 1)) rhs[1] -= P.stiff[1] * (q[1] + P.dt * dq[1] - P.rest[1]);
  sfor<0, NL>([&](auto K) {
    constexpr int k = K, i = 2 + k;
    H[tri(rev<N>(i), rev<N>(0))] = -P.sigma[k] * dcy[k];
    H[tri(rev<N>(i), rev<N>(1))] = P.sigma[k] * dcx[k];
    sfor<0, k + 1>([&](auto J) {
      constexpr int j = J;
      if constexpr (is_anc<T>(j, k)) {
        Real v = Ip[k] + dcx[k] * (px[k] - px[j]) + dcy[k] * (py[k] - py[j]);
        H[tri(rev<N>(i), rev<N>(2 + j))] = P.sigma[k] * P.sigma[j] * v;
      } else {
        H[tri(rev<N>(i), rev<N>(2 + j))] = Real(0);
      }
    });
    rhs[i] = tau[i] - P.sigma[k] * Nz[k];
    if constexpr (!DART_ZERO(PT, damp, i)) rhs[i] -= P.damp[i] * dq[i];
    if constexpr (!DART_ZERO(PT, stiff, i)) rhs[i] -= P.stiff[i] * (q[i] + P.dt * dq[i] - P.rest[i]);
  });
  if (EXTRAS && P.ex.ext_force != nullptr) {   // external force at the frame origin (= joint origin) of link ext_link: generalized force J^T f
    const Real fx = P.ex.ext_force[env * 3], fy = P.ex.ext_force[env * 3 + (topo_plane_xz<T>::value ? 2 : 1)];
    rhs[0] += fx; rhs[1] += fy;
    sfor<0, NL>([&](auto K) {
      constexpr int k = K;
      if (P.ex.ext_link == k)
        sfor<0, k + 1>([&](auto J) {
          constexpr int j = J;
          if constexpr (is_anc<T>(j, k)) rhs[2 + j] += P.sigma[j] * ((px[k] - px[j]) * fy - (py[k] - py[j]) * fx);
        });
    });
  }
  // A3 (card.impulse_inertia): the forward dynamics always solves (M + E) qdd = rhs, E = dt D + dt^2 K; the impulse pass below
  // runs on M (DART 6, the default) or, with the knob at 0, on M + E like the forward dynamics.
  const bool impulse_M = P.impulse_M != 0;   // a compile-time constant of the baked models, wave-uniform otherwise
  if (!impulse_M)
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      if constexpr (!DART_ZERO(PT, damp, i)) H[tri(rev<N>(i), rev<N>(i))] += P.dt * P.damp[i];
      if constexpr (!DART_ZERO(PT, stiff, i)) H[tri(rev<N>(i), rev<N>(i))] += P.dt * P.dt * P.stiff[i];
    });
#ifdef DART_ROOT_FIRST
  constexpr bool REV = false;
#else
  constexpr bool REV = true;
#endif
  Real vs[N];
  {
    spd_inverse<Real, N>(H);  // H now holds the inverse of the impulse inertia (reversed dof order)
    Real acc[N];
    sfor<0, N>([&](auto I) {
      constexpr int i = I;
      Real a = Real(0);
      sfor<0, N>([&](auto J) { constexpr int j = J; a += H[tri(rev<N>(i), rev<N>(j))] * rhs[j]; });
      acc[i] = a;
    });
    if (impulse_M) implicit_accel<Real, N, REV, ImplicitDofs<PT, N>>(P, H, acc);
    sfor<0, N>([&](auto I) { constexpr int i = I; vs[i] = dq[i] + P.dt * acc[i]; });
  }
  constexpr bool HLDS = hinv_lds<T, Real>(), HLATE = hinv_lds_late<T, Real>();
  if constexpr (HLDS && !HLATE) {   // park H^-1 in this lane's LDS column (topo_hinv_lds64); from here on it is read from there
    sfor<0, N*(N + 1) / 2>([&](auto I) { constexpr int i = I; hl[64 * i] = H[i]; });
    DART_COMPILER_FENCE();
  }
  auto Hv = [&](int k) -> Real { if constexpr (HLDS && !HLATE) return hl[64 * k]; else return H[k]; };   // (up to the register tiers)

  // ---- candidate contacts at q_t: every capsule's lowest segment endpoint against the floor (ODE capsule-plane as DART
  // uses it: one contact, position in the middle of the penetration)
  bool con[NC];
  Real cPx[NC], cPy[NC], cdep[NC];
  int nact = 0;
  if constexpr (!topo_contacts<T>::value) { sfor<0, NC>([&](auto Cc) { con[Cc] = false; cPx[Cc] = Real(0); cPy[Cc] = Real(0); cdep[Cc] = Real(0); }); }
  else sfor<0, NC>([&](auto Cc) {
    constexpr int cidx = Cc, k = T::clink(cidx);
    Real x1 = px[k] + c[k] * P.e1x[cidx] - s[k] * P.e1y[cidx], y1 = py[k] + s[k] * P.e1x[cidx] + c[k] * P.e1y[cidx];
    Real x2 = px[k] + c[k] * P.e2x[cidx] - s[k] * P.e2y[cidx], y2 = py[k] + s[k] * P.e2x[cidx] + c[k] * P.e2y[cidx];
    bool second = y2 < y1;  // lowest endpoint; exact tie -> first (+axis) end
    Real ex = second ? x2 : x1, ey = second ? y2 : y1;
    Real d = (P.root_y0 + q[1] + ey) - P.ground_y;
    con[cidx] = d <= P.rad[cidx];
    cdep[cidx] = P.rad[cidx] - d;
    cPx[cidx] = ex; cPy[cidx] = ey - Real(0.5) * (P.rad[cidx] + d);  // ODE sphere-sphere contact position
    nact += con[cidx] ? 1 : 0;
  });

  if (EXTRAS && rp.rec != nullptr) {   // nothing touching, no limit active: an empty report
    *rp.count = 0;
    sfor<0, N>([&](auto I) { rp.cf[I] = Real(0); });
  }
  bool slow = false;
  // joints at a limit (topologies with limit slots): a lane with more of them than the small tier has slots ...
  constexpr int NLSS = small_limit_slots<T>();
  constexpr bool LIM_SLOTS = NLSS < n_limited<T>();
  // ... is served by the wave solvers where the topology has them (half cheetah, physics-only walker / cheetah trees): which solver serves an env
  // stays a function of the env alone and the kernel carries ONE register tier; elsewhere (Walker2d) the wave votes for the all-limits tier
  constexpr bool LIM_TO_WAVE = LIM_SLOTS && topo_wave_fallback<T>::value && has_slow_path<T, Real>();
  int nla = 0;
  if constexpr (LIM_SLOTS) sfor<0, NL>([&](auto K) { constexpr int k = K; if constexpr (T::limited(k)) nla += (q[2 + k] <= P.lo[k] || q[2 + k] >= P.hi[k]) ? 1 : 0; });
  if constexpr (has_slow_path<T, Real>()) {
    slow = nact > last_tier<T, Real>() || (P.force_slow > 0 && nact > 0);
    if constexpr (LIM_TO_WAVE) slow = slow || nla > NLSS;
    // (unlikely only where the fallback is the single-lane loop -- Hopper, Walker2d: a handful of lanes per launch; the half cheetah's wave
    // solvers run in 94 % of its waves)
    if (topo_wave_fallback<T>::value ? __any(slow) : DART_UNLIKELY(__any(slow))) {
      // the rare lanes whose env touches the floor with more capsules than the tiers hold: one after the other
#ifdef DART_WAVE_TIMING
      const long long tclk0 = DART_CLK();
#endif
      auto stage_inputs = [&](Real* m) {   // this lane's env -> a fallback solver's LDS block
        sfor<0, N>([&](auto I) { constexpr int i = I; sfor<0, N>([&](auto J) { constexpr int j = J; m[i * N + j] = Hv(tri(rev<N>(i), rev<N>(j))); }); });
        m += N * N;
        sfor<0, NL>([&](auto K) { m[K] = px[K]; m[NL + K] = py[K]; m[2 * NL + K] = P.sigma[K]; });
        m += 3 * NL;
        sfor<0, N>([&](auto I) { m[I] = vs[I]; });
        m += N;
        sfor<0, NC>([&](auto Cc) { m[Cc] = con[Cc] ? Real(1) : Real(0); m[NC + Cc] = cPx[Cc]; m[2 * NC + Cc] = cPy[Cc]; m[3 * NC + Cc] = cdep[Cc]; });
        m += 4 * NC;
        sfor<0, NL>([&](auto K) {
          constexpr int k = K;
          Real lim = Real(0), viol = Real(0);
          if constexpr (T::limited(k)) {
            const bool low = q[2 + k] <= P.lo[k], up = (!low) && (q[2 + k] >= P.hi[k]);
            lim = low ? Real(-1) : (up ? Real(1) : Real(0));
            viol = low ? (q[2 + k] - P.lo[k]) : (q[2 + k] - P.hi[k]);
          }
          m[k] = lim; m[NL + k] = viol;
        });
      };
      Real* mvs = slow_mem + N * N + 3 * NL;
#ifdef DART_WAVE_COOP
      if (topo_wave_fallback<T>::value && blockDim.x == 64) {   // a full wave: all 64 lanes serve the env together (wave_constraints)
        int budget = DART_COOP_GUARD;   // runaway guard over every env served in this world step (coop_iters: a solve's cap is its own)
        bool left = slow;
#if DART_COOP4
        // four envs per pass, one per row of 16 lanes, for every env of at most 16 rows (wave_constraints4); eligibility is the env's own
        // business (its contacts and active limits), so which solver serves it does not depend on its wave mates
        if constexpr (topo_wave_fallback<T>::value && coop4_fits<T>()) {
          int nlim = 0;
          sfor<0, NL>([&](auto K) { constexpr int k = K; if constexpr (T::limited(k)) nlim += (q[2 + k] <= P.lo[k] || q[2 + k] >= P.hi[k]) ? 1 : 0; });
          const bool elig = slow && 2 * nact + nlim <= 16;
          Real* blocks = slow_mem + constraint_lds_words<T, Real>() + coop_words<16>();
          const int lane = (int)(threadIdx.x & 63);
          Real* gblk = blocks + (lane >> 4) * coop4_words<T>();
          unsigned long long todo4 = __ballot(elig);
          while (todo4 != 0ull) {
            const int rank = __popcll(todo4 & ((1ull << lane) - 1ull));
            const bool served = ((todo4 >> lane) & 1ull) && rank < 4;
            const int count = __popcll(todo4) < 4 ? __popcll(todo4) : 4;
            Real* oblk = served ? blocks + rank * coop4_words<T>() : nullptr;
            if (served) stage_inputs(oblk);
            if ((lane >> 4) >= count) {   // an idle group: nothing touches, no limit active
              Real* icon = gblk + N * N + 3 * NL + N;
              if ((lane & 15) < NC) icon[lane & 15] = Real(0);
              if ((lane & 15) < NL) icon[4 * NC + (lane & 15)] = Real(0);
            }
            wave_constraints4<Real, T, PT, EXTRAS>(P, gblk, oblk, q[0], q[1], rp, budget);
#ifdef DART_WAVE_TIMING
            if (P.stats != nullptr && lane == 0) atomicAdd(&P.stats[6], 1ull);   // [6] passes of wave_constraints4, [7] envs served one at a time
#endif
            if (served) { const Real* ovs = oblk + N * N + 3 * NL; sfor<0, N>([&](auto I) { vs[I] = ovs[I]; }); }
            todo4 &= ~__ballot(served);
            __syncthreads();
          }
          left = slow && !elig;
        }
#endif
        unsigned long long todo = __ballot(left);
        while (todo != 0ull) {
          const int owner = __ffsll((long long)todo) - 1;
          todo &= todo - 1ull;
          if ((int)(threadIdx.x & 63) == owner) stage_inputs(slow_mem);
          if constexpr (topo_wave_fallback<T>::value) wave_constraints<Real, T, PT, EXTRAS>(P, slow_mem, q[0], q[1], rp, owner, budget);
#ifdef DART_WAVE_TIMING
          if (P.stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(&P.stats[7], 1ull);
#endif
          if ((int)(threadIdx.x & 63) == owner) sfor<0, N>([&](auto I) { vs[I] = mvs[I]; });
          __syncthreads();
        }
      } else
#endif
      for (int turn = 0; turn < 64; ++turn) {   // (partial waves, and the host build: the env's own lane, alone)
        if (slow && (int)(threadIdx.x & 63) == turn) {
          stage_inputs(slow_mem);
          slow_constraints<Real, T, PT, EXTRAS>(P, slow_mem, q[0], q[1], rp);
          sfor<0, N>([&](auto I) { vs[I] = mvs[I]; });
        }
      }
#ifdef DART_WAVE_TIMING
      wave_timing_add(P.stats, 3, 16, 8, 16, DART_CLK() - tclk0);
#endif
    }
  }
  if constexpr (HLATE) {   // the wave-served phase is over (its last pass ended on a barrier): its blocks take this wave's H^-1 columns
    sfor<0, N*(N + 1) / 2>([&](auto I) { constexpr int i = I; hl[64 * i] = H[i]; });
    DART_COMPILER_FENCE();
  }
  // ---- register tiers: the smallest one that holds every (remaining) lane's contacts
  const int nreg = slow ? 0 : nact;
#ifdef DART_WAVE_TIMING
  const long long tclk1 = DART_CLK();
  bool big_tier = false;
#endif
  Real* cm = nullptr;   // hand-off block of the wave solver, behind the fallback / tier memory (full waves only: it needs all 64 lanes)
#ifdef DART_WAVE_COOP
  if constexpr (topo_wave_fallback<T>::value) cm = blockDim.x == 64 ? slow_mem + constraint_lds_words<T, Real>() : nullptr;
#endif
  // limit rows: the small tier carries small_limit_slots<T>() of them (topo_limit_slots).  Topologies without wave solvers (Walker2d): a wave
  // with a lane that has more joints at their limits runs the instantiation with one row per limited joint -- the wave's vote, as for the contact
  // slots.  A lane's LCP has the same rows that can move either way and the same start sets (warm sets travel in joint layout), so on the host
  // build the two give bitwise the same numbers (tests/test_lane_kernels_as_waves.py); on the device the two instantiations round differently
  // in the last bits (measured: profiles/r06_limit_slots.txt), so THERE the vote can show in the last bits of the wave mates of a lane with five
  // or six of its six joints at their limits -- never seen in 245 k lane-substeps of random-action rollouts, nor by the 65 536-env
  // batch-independence test.  Topologies with wave solvers never vote: such a lane is `slow` (above).
  bool lim_all = false;   // (wave-uniform)
  if constexpr (LIM_SLOTS && !LIM_TO_WAVE) {
    lim_all = __any(!slow && nla > NLSS);
#ifdef DART_LIMIT_NO_FALLBACK   // (timing experiments only: the compacted tier whatever the count -- WRONG for a lane with more joints at their limits)
    lim_all = false;
#endif
  }
  // (topo_limit_prefix: the same vote on "some lane has a joint beyond the prefix at its limit")
  constexpr int NLPF = small_limit_prefix<T>();
  constexpr bool LIM_PFX = !LIM_SLOTS && NLPF < n_limited<T>();
  if constexpr (LIM_PFX) {
    bool later = false;
    sfor<0, NL>([&](auto K) { constexpr int k = K; if constexpr (T::limited(k) && lim_ord<T>(k) >= NLPF) later = later || q[2 + k] <= P.lo[k] || q[2 + k] >= P.hi[k]; });
    lim_all = __any(!slow && later);
  }
  auto small_tier = [&]() {
    if constexpr (LIM_PFX) {
      if (lim_all) constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
      else constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS, NLPF, true>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
    } else if constexpr (LIM_TO_WAVE) {
      constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS, NLSS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
    } else if constexpr (LIM_SLOTS) {
      // (inlined, although it almost never runs: as a real call on copies of its inputs -- built and measured in round 6 -- the call's mere
      // presence cost the compacted tier its registers: Walker2d fp64 89.5 -> 114.4 us, HBM traffic per launch 143 -> 326 MB)
      if (DART_UNLIKELY(lim_all)) constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
      else constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS, NLSS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
    } else {
      constraint_phase<Real, T, PT, T::TIER0, EXTRAS, HLDS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
    }
  };
  if constexpr (tier1<T, Real>() > 0) {
    if (__any(nreg > T::TIER0)) {
#ifdef DART_WAVE_TIMING
      big_tier = true;
#endif
      constraint_phase<Real, T, PT, tier1<T, Real>(), EXTRAS, HLDS>(P, q, H, px, py, vs, con, cPx, cPy, cdep, slow, warm, rp, hl, cm);
    }
    else small_tier();
  } else {
    small_tier();
  }
#ifdef DART_WAVE_TIMING
  wave_timing_add(P.stats, big_tier ? 4 : 5, big_tier ? 32 : 48, 8, 16, DART_CLK() - tclk1);
#endif
  sfor<0, N>([&](auto I) { constexpr int i = I; dq[i] = vs[i]; q[i] += P.dt * vs[i]; });
}

// ------------------------------------------------------------------ Philox4x32-10 (counter-based reset noise)
__device__ __host__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// reset noise for env `gid`, episode `ep`: q_i = -r + 2r*u(2i block...), same stream on host and device
template <class Real, int N>
__device__ __host__ inline void reset_noise(uint64_t seed, uint64_t gid, uint32_t ep, Real r, Real rv, Real (&q)[N],
                                            Real (&dq)[N]) {
  constexpr int NW = 2 * N;
  Real u[(NW + 3) / 4 * 4];
  for (int blk = 0; blk < (NW + 3) / 4; ++blk) {
    uint32_t o[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), ep, (uint32_t)blk, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    for (int j = 0; j < 4; ++j) u[4 * blk + j] = Real(o[j] >> 8) * Real(1.0 / 16777216.0);
  }
  for (int i = 0; i < N; ++i) { q[i] = -r + Real(2) * r * u[i]; dq[i] = -rv + Real(2) * rv * u[N + i]; }
}

// ------------------------------------------------------------------ observation (hopper.py:67-74, walker2d.py:67-74)
template <class Real, class T, class PT>
__device__ __forceinline__ Real root_height(const PT& P, const Real (&q)[T::NDOF]) {
  Real h = P.root_y0 + q[1];
  if constexpr (DART_ZERO(PT, cx, 0) && DART_ZERO(PT, cy, 0)) return h;
  if (P.cx[0] != Real(0) || P.cy[0] != Real(0)) {
    Real sj, cj;
    sincos_<Real>(q[2], sj, cj);
    h += P.sigma[0] * sj * P.cx[0] + cj * P.cy[0];
  }
  return h;
}
template <class Real, class T, class PT>
__device__ __forceinline__ void write_obs(const PT& P, const Real (&q)[T::NDOF], const Real (&dq)[T::NDOF],
                                          Real height, float* __restrict__ obs_row) {
  constexpr int N = T::NDOF;
  if constexpr (topo_physics<T>::value) {   // [q, dq]
    sfor<0, N>([&](auto I) { constexpr int i = I; obs_row[i] = (float)q[i]; obs_row[N + i] = (float)dq[i]; });
    return;
  }
  obs_row[0] = (float)height;
  sfor<2, N>([&](auto I) { constexpr int i = I; obs_row[i - 1] = (float)q[i]; });
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    obs_row[N - 1 + i] = (float)fmin(fmax(dq[i], -P.v_clip), P.v_clip);
  });
}

// ------------------------------------------------------------------ kernels
// One batched env.step(): clamp+scale action, frame_skip world steps, reward, done, TimeLimit, observation,
// optional on-device auto-reset (post-reset observation is returned, as SyncVectorEnv does).
// EXTRAS: the instantiation that serves an external body force and / or the contact report (Extras); the lean one carries
// none of that code (measured: its mere presence cost the fp64 Walker2d kernel 22 %).
template <class Real, class T, class PT, bool EXTRAS = false>
__global__ void __launch_bounds__(64) step_kernel(PT P, int64_t n_envs, Real* __restrict__ qs,
                                                   Real* __restrict__ dqs, int32_t* __restrict__ elapsed,
                                                   uint32_t* __restrict__ episode, const float* __restrict__ actions,
                                                   float* __restrict__ obs, float* __restrict__ reward,
                                                   uint8_t* __restrict__ done, uint8_t* __restrict__ truncated,
                                                   int autoreset, uint64_t seed, uint64_t env_offset) {
  constexpr int N = T::NDOF, NA = T::NA;
#ifdef DART_WAVE_TIMING
  const long long tclk_wave = DART_CLK();
#endif
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = e < n_envs;
  int64_t ec = valid ? e : n_envs - 1;  // tail lanes shadow the last env so wave votes stay uniform
  Real q[N], dq[N], tau[N];
  sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + ec]; dq[i] = dqs[(int64_t)i * n_envs + ec]; });
  // the episode bookkeeping the epilogue needs is fetched HERE, with the state: at one wave per SIMD nothing hides a load issued
  // at the end of the kernel (two dependent HBM round trips there cost the Hopper kernel ~10 % of its 30 us)
  int el_in = elapsed[ec];
  uint32_t ep_in = episode[ec];
  Real a2 = Real(0);
  sfor<0, N>([&](auto I) { tau[I] = Real(0); });
  sfor<0, NA>([&](auto K) {
    constexpr int k = K;
    Real a = (Real)actions[ec * NA + k];
    a2 += a * a;  // control cost uses the unclamped action (hopper.py:55)
    Real cl = (a > P.act_hi[k]) ? P.act_hi[k] : a;   // comparison clamp as hopper.py:25-30: a NaN action stays NaN
    cl = (cl < P.act_lo[k]) ? P.act_lo[k] : cl;
    tau[N - NA + k] = cl * P.act_scale[k];
  });
  DART_PIN_VGPR(el_in); DART_PIN_VGPR(ep_in);   // (pinned here, where the state loads are awaited anyway: the compiler must not sink them)
  Real x_before = q[0];
  Real dx = Real(0);
  WarmSets warm;
  // LDS of the single-lane fallback solver (only topologies with more candidate capsules than tier slots have one)
  // H^-1 of every lane, one 64-lane column per packed entry (topo_hinv_lds64; only the topologies that ask for it) -- a block of its
  // own, or (topo_hinv_lds_late) the four-env blocks behind the fallback solver's, which are idle once the wave-served phase is over
  constexpr int HL_WORDS = 64 * (N * (N + 1) / 2), SLOW_HEAD = constraint_lds_words<T, Real>() + (topo_wave_fallback<T>::value ? coop_words<16>() : 0);
  constexpr bool HL_LATE = hinv_lds_late<T, Real>();
  __shared__ Real slow_lds[SLOW_HEAD + ((HL_LATE && HL_WORDS > coop4_lds_words<T>()) ? HL_WORDS : coop4_lds_words<T>())];
  __shared__ Real hinv_lds_[(hinv_lds<T, Real>() && !HL_LATE) ? HL_WORDS : 1];
  Real* hl = (HL_LATE ? slow_lds + SLOW_HEAD : hinv_lds_) + (threadIdx.x & 63);
#pragma unroll 1
  for (int f = 0; f < P.frame_skip; ++f) {
    ReportTo<Real> rp;
    if (EXTRAS && P.ex.creport != nullptr && f == P.frame_skip - 1 && valid) {
      rp.rec = P.ex.creport + (size_t)e * T::NC * 8; rp.count = P.ex.creport_count + e; rp.cf = P.ex.cf_report + (size_t)e * N;
    }
#ifdef DART_EMU_TRACE
    dart_emu_substep((int)ec, f);   // host build of tests/diag only: which env / world step the following pivoting runs belong to
#endif
    world_step<Real, T, PT, EXTRAS>(P, q, dq, tau, warm, slow_lds, ec, rp, hl);
    dx += P.dt * dq[0];
  }
  (void)x_before;
  Real height = root_height<Real, T, PT>(P, q);
  Real ang = q[2];
  Real pen = Real(0);
  if (P.penalty_link >= 0) {
    sfor<1, T::NL>([&](auto K) {
      constexpr int k = K;
      if (P.penalty_link == k) {
        if ((P.lo[k] - q[2 + k]) > -P.pen_margin) pen += P.pen_each;
        if ((P.hi[k] - q[2 + k]) < P.pen_margin) pen += P.pen_each;
      }
    });
  }
  Real rew = dx * P.inv_envdt + P.alive - P.ctrl_cost * a2 - pen;
  if constexpr (topo_physics<T>::value) rew = Real(0);
  bool ok = true;
  sfor<0, N>([&](auto I) {
    constexpr int i = I;
    ok = ok && isfinite(q[i]) && isfinite(dq[i]) && (fabs(dq[i]) < P.s_max);
    if constexpr (i >= 2) ok = ok && (fabs(q[i]) < P.s_max);
  });
  if constexpr (topo_fluid<T>::value) {   // snake_7link.py:72-84: deviation cost on the heading, observation q[1:], dq
    if (P.task == 9) { rew -= P.dev_cost * fabs(q[2]); height = q[1]; }
  }
  if (P.task == 6) {   // half_cheetah.py:50-63: the reward is zeroed when the state broke; the observation starts with q[1] itself
    rew = ok ? rew : Real(0);
    height = q[1];
  }
  ok = ok && (height > P.h_lo) && (height < P.h_hi) && (fabs(ang) < P.ang_max);
  bool task_done = (P.task != 0) && !ok;
  int el = el_in + 1;
  bool trunc = (P.max_steps > 0) && (el >= P.max_steps);
  bool dn = task_done || trunc;
  if (autoreset && dn) {
    if (P.ex.mt != nullptr) {   // (wave-uniform) reference-exact reset: this env's own numpy stream (the episode counter keys Philox only)
      if (valid) mt_reset_draw<Real, N>(*P.ex.mt, n_envs, e, q, dq);
    } else {
      uint32_t ep = ep_in + 1;
      reset_noise<Real, N>(seed, env_offset + (uint64_t)ec, ep, P.noise, P.noise_v, q, dq);
      sfor<0, N>([&](auto I) { constexpr int i = I; q[i] += P.q0[i]; dq[i] += P.dq0[i]; });
      if (valid) episode[e] = ep;
    }
    el = 0;
    height = (P.task == 6 || P.task == 9) ? q[1] : root_height<Real, T, PT>(P, q);
  }
  if (valid) {
    sfor<0, N>([&](auto I) { constexpr int i = I; qs[(int64_t)i * n_envs + e] = q[i]; dqs[(int64_t)i * n_envs + e] = dq[i]; });
    elapsed[e] = el;
    // (Round 5 tried staging the observations through LDS so that a workgroup's (64 x obs_dim) block leaves as float4 per lane, and with it
    // a second destination in page-locked host memory -- no copy kernel behind the step kernel.  Measured, profiles/r05_host_path.txt: the
    // kernels got SLOWER -- Hopper fp64 31.8 -> 32.1 us, Walker2d fp64 101.0 -> 104.1 us: the epilogue's LDS round trip and its extra live
    // values cost more than eleven scattered dword stores that nobody waits for -- and the host path gained 7 us of 147, because every wave
    // reaches its epilogue at the same time: the 3.4 MB cross PCIe after the arithmetic either way.  Reverted.)
    write_obs<Real, T, PT>(P, q, dq, height, obs + e * obs_dim_of<T>());
    reward[e] = (float)rew;
    done[e] = dn ? 1 : 0;
    truncated[e] = (trunc && !task_done) ? 1 : 0;
  }
#ifdef DART_WAVE_TIMING
  if (P.stats != nullptr && (threadIdx.x & 63) == 0) {
    const long long dt = DART_CLK() - tclk_wave;
    wave_timing_add(P.stats, 0, 8, 14, 8, dt);
    atomicMax(&P.stats[1], (unsigned long long)dt);
    atomicAdd(&P.stats[2], 1ull);
  }
#endif
}

// Masked reset: q = init + noise (host-supplied rows, or Philox when noise pointers are null), elapsed = 0, obs.
template <class Real, class T, class PT>
__global__ void __launch_bounds__(256) reset_kernel(PT P, int64_t n_envs, Real* __restrict__ qs,
                                                     Real* __restrict__ dqs, int32_t* __restrict__ elapsed,
                                                     uint32_t* __restrict__ episode, const uint8_t* __restrict__ mask,
                                                     const double* __restrict__ qnoise, const double* __restrict__ vnoise,
                                                     float* __restrict__ obs, uint64_t seed, uint64_t env_offset,
                                                     int obs_masked_only) {
  constexpr int N = T::NDOF;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  Real q[N], dq[N];
  bool m = (mask == nullptr) || mask[e];
  if (m) {
    if (qnoise) {
      sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = (Real)qnoise[e * N + i]; dq[i] = (Real)vnoise[e * N + i]; });
    } else {
      uint32_t ep = episode[e] + 1;
      reset_noise<Real, N>(seed, env_offset + (uint64_t)e, ep, P.noise, P.noise_v, q, dq);
      sfor<0, N>([&](auto I) { constexpr int i = I; q[i] += P.q0[i]; dq[i] += P.dq0[i]; });
      episode[e] = ep;
    }
    sfor<0, N>([&](auto I) { constexpr int i = I; qs[(int64_t)i * n_envs + e] = q[i]; dqs[(int64_t)i * n_envs + e] = dq[i]; });
    elapsed[e] = 0;
  } else {
    sfor<0, N>([&](auto I) { constexpr int i = I; q[i] = qs[(int64_t)i * n_envs + e]; dq[i] = dqs[(int64_t)i * n_envs + e]; });
  }
  if (obs && (m || !obs_masked_only)) write_obs<Real, T, PT>(P, q, dq, (P.task == 6 || P.task == 9) ? q[1] : root_height<Real, T, PT>(P, q), obs + e * obs_dim_of<T>());
}

// (N, n) row-major doubles  <->  SoA state, for set_state / get_state (dart_env.py:145-148, 211-215)
template <class Real, int N>
__global__ void state_io_kernel(int64_t n_envs, Real* __restrict__ qs, Real* __restrict__ dqs, double* __restrict__ qh,
                                double* __restrict__ dqh, int to_device) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  for (int i = 0; i < N; ++i) {
    if (to_device) { qs[(int64_t)i * n_envs + e] = (Real)qh[e * N + i]; dqs[(int64_t)i * n_envs + e] = (Real)dqh[e * N + i]; }
    else { qh[e * N + i] = (double)qs[(int64_t)i * n_envs + e]; dqh[e * N + i] = (double)dqs[(int64_t)i * n_envs + e]; }
  }
}

}  // namespace dartk
