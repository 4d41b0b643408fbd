/* dart_stepper.h -- C ABI of the MI355X batched Dart stepper (libdart_stepper.so).
 *
 * The reference has no C boundary for this path: its envs drive DART through the
 * pydart2 *Python object* API.  Each entry point below names the reference call
 * sites it replaces for a whole batch of N environments at once (SURVEY.md 8b,
 * Appendix D).  Conventions mirror the reference's: the library owns the world
 * state (one "world" per env, SoA in HBM), getters/setters copy, the caller
 * serialises calls per handle (one in-flight call, like one DART world per env),
 * different handles (one per GPU) may be driven from different threads/processes.
 * Host buffers are caller-owned, row-major (N, k) exactly as numpy lays them out.
 *
 * Every function returns 0 on success or a negative DART_E_* code;
 * dart_last_error() gives the message (library-owned string).
 *
 * There is NO CPU fallback: with no HIP device dart_create fails with
 * DART_E_NO_DEVICE.
 */
#ifndef DART_STEPPER_H
#define DART_STEPPER_H

#include <stdint.h>

#include "dart_model_card.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DartStepper DartStepper;

enum {
  DART_OK = 0,
  DART_E_INVALID = -1,     /* bad argument / null handle */
  DART_E_NO_DEVICE = -2,   /* no HIP device (this library never computes on the CPU) */
  DART_E_UNSUPPORTED = -3, /* model topology has no compiled kernel */
  DART_E_HIP = -4,         /* HIP runtime error, see dart_last_error */
  DART_E_PENDING = -5,     /* step_async already pending (gym.error.AlreadyPendingCallError) */
  DART_E_NOT_PENDING = -6  /* step_wait without step_async (gym.error.NoAsyncCallError) */
};

/* dart_query keys */
enum {
  DART_Q_NUM_ENVS = 0, DART_Q_NDOFS = 1, DART_Q_OBS_DIM = 2, DART_Q_ACT_DIM = 3, DART_Q_FRAME_SKIP = 4,
  DART_Q_PRECISION = 5, DART_Q_DEVICE = 6, DART_Q_LCP_SLOTS = 7,
  DART_Q_STATIC_KERNEL = 8, /* 1: the card matched a model baked in at build time (csrc/static_models.hpp, csrc/tree_patterns.hpp) */
  DART_Q_MAX_CONTACTS = 9,  /* contact points the kernel keeps per env and world step (0: no reporting in this kernel) */
  DART_Q_LDS_BYTES = 10,    /* LDS bytes per workgroup of the step kernel (tree kernel: the per-env block; planar kernels: the fallback solver's) */
  DART_Q_LANE_KERNEL = 11   /* 1: the model runs one env per GPU lane (a register kernel whose topology the card matched); 0: the tree kernel */
};

/* dart_configure keys */
enum {
  DART_CFG_SOLVER = 0,      /* 0 = block principal pivoting (exact, default), 1 = projected Gauss-Seidel */
  DART_CFG_ITERS_STAGE1 = 1,/* iteration cap (solver 0) / sweep count (solver 1) of the frictionless stage */
  DART_CFG_ITERS_STAGE2 = 2,/* same for the friction stage */
  DART_CFG_AUTORESET = 3,   /* 1: done envs are reset inside dart_step with on-device Philox noise */
  DART_CFG_SEED = 4,        /* Philox key (low 53 bits of the double are used) */
  DART_CFG_ENV_OFFSET = 5,  /* global index of env 0 of this handle (multi-GPU sharding keeps streams distinct) */
  DART_CFG_BLOCK_THREADS = 6,/* envs (active lanes) per wave64 workgroup of the step kernel: 64, 32 or 16 */
  DART_CFG_STATS = 7,       /* 1: histogram the wave-level pivoting iteration counts (dart_get_stats) */
  DART_CFG_EPISODE_STATS = 8,/* 1: keep per-env episode return / length accumulators on the device (dart_get_episode_stats) */
  DART_CFG_CONTACT_REPORT = 9, /* 1: record the contacts of every env-step's last world step (dart_get_contacts) */
  DART_CFG_DEBUG_FORCE_FALLBACK = 10, /* planar register kernels, tests only: 1 routes every env that touches the floor through the
                               single-lane fallback solver, the path of an env with more contacts than the kernel's slot tiers hold */
  /* 12: retired (DART_CFG_WAVE_VOTE of rounds 4-5, the per-wave choice between a second register tier and the wave solvers in the physics-only
         walker / cheetah kernels).  Round 6 deleted that tier: every env beyond two touching capsules goes to the wave solvers, which solver serves
         an env -- and how many iterations it may spend -- is a function of the env alone, so results no longer depend on the wave mates.
         dart_configure returns DART_E_INVALID for it. */
  DART_CFG_HOST_DMA = 13,   /* how the host-buffer entry points (dart_step, dart_step_async[_to]) cross PCIe, a bit mask, default 3:
                               bit 0 = the step kernel reads the actions straight from page-locked host memory (no H2D copy),
                               bit 1 = the outputs return through a copy kernel writing the mapped host block (no SDMA copy),
                               bit 2 = the four outputs as four separate copies (rounds 1-2),
                               bit 3 = (round 6) the reference-exact MT19937 auto-reset as two launches behind the step kernel, as rounds 1-5
                                       ran it, also where the step kernel now does it in its epilogue (the lane kernels of the tasks whose
                                       reset_model draws the two noise vectors and nothing else).
                               0 = copy engines both ways.  Results do not depend on it; it exists for A/B measurements
                               (tools/gpu/host_path_c.py). */
  DART_CFG_LAUNCH_ORDER = 11 /* tree kernel (one env per workgroup): 1 (default) = the workgroups of a step are dispatched in the order
                               of the envs' durations at the previous step, longest first -- a launch ends when its last workgroup
                               does, and an env that was expensive (many contacts, a long pivoting run) mostly still is; 0 = index
                               order.  Results do not depend on it (each env is stepped by one workgroup either way). */
};

/* Environment: the library reads NO environment variables (round 5: the debugging switches DART_FORCE_SPATIAL, DART_GENERIC_KERNEL,
 * DART_SPLIT_D2H, DART_HOST_DMA of earlier rounds are gone -- the tree kernel is chosen with card.generic_kernel, the PCIe legs with
 * DART_CFG_HOST_DMA).  What a handle does is decided by its card and its dart_configure keys alone.
 *
 * Threading: one thread at a time per handle; different handles may be driven from different threads.
 * Library-level error text for failures that happen before a handle exists (handle == NULL): kept per calling THREAD
 * (thread_local), so two threads inside dart_create do not share it. */
const char* dart_last_error(const DartStepper* h);

/* Replaces DartEnv.__init__'s world construction for N envs: pydart.World(dt, skel), skeletons[-1], limit
 * enforcement (reference gym/envs/dart/dart_env.py:55,62-67).
 * precision: 64 = fp64 kernels, the PRODUCT DEFAULT of every layer above this ABI -- the reference computes in doubles
 *   (DART / Eigen) and this is the mode that meets the north star's tolerance (RMS state divergence vs the CPU path < 1e-4 over
 *   1 000 env-steps: measured 5e-13 untrimmed).  32 = fast mode (fp32 kernels, 15-40 % quicker): it does NOT meet that tolerance
 *   (untrimmed RMS q / dq after 1 000 env-steps: Hopper 4e-6 / 2e-4, Walker2d 1e-3 / 4e-2, HumanWalker 1e-3 / 6e-2 -- the envs
 *   whose contact or limit switched one 2 ms substep early); ask for it explicitly.
 * State starts at the model's init_pos/init_vel, elapsed = 0. */
int dart_create(const DartModelCard* card, int64_t num_envs, int device, int precision, DartStepper** out);

/* Frees device and pinned host memory.  NULL is accepted. */
int dart_destroy(DartStepper* h);

int dart_query(const DartStepper* h, int what, int64_t* out);
int dart_configure(DartStepper* h, int key, double value);

/* Replaces DartEnv.reset()/reset_model() (dart_env.py:140-143, hopper.py:76-84) for the envs with mask[i] != 0
 * (mask == NULL: all): world.reset() then q = init + qpos_noise[i], dq = init + qvel_noise[i], elapsed = 0.
 * qpos_noise/qvel_noise: (N, ndofs) float64, rows of unmasked envs are ignored; both NULL -> on-device Philox
 * U(-reset_noise, +reset_noise).  obs_out: (N, obs_dim) float32 of ALL envs (may be NULL). */
int dart_reset(DartStepper* h, const uint8_t* mask, const double* qpos_noise, const double* qvel_noise,
               float* obs_out);

/* Replaces DartEnv.seed for the whole batch (reference dart_env.py:117-119, gym/utils/seeding.py:11-19, and the
 * `s + i` fan-out of gym/vector/sync_vector_env.py:50-58): builds one MT19937 generator per env in HBM from
 * keys[i][0..key_len[i]) -- the uint32 words the reference hands to numpy's RandomState.seed(list) -- and switches the
 * handle to "MT19937 noise": dart_reset(mask, NULL, NULL, ..) and the auto-reset of dart_step then draw
 * uniform(-reset_noise, reset_noise, ndofs) for qpos and qvel from env i's own stream, bit-exact with the reference.
 * keys: (N, 2) uint32, key_len: (N,) int32 in {1, 2}. */
int dart_seed_mt19937(DartStepper* h, const uint32_t* keys, const int32_t* key_len);

/* Replaces DartEnv.set_state / state_vector (dart_env.py:145-148, 211-215) for the whole batch.
 * q, dq: (N, ndofs) float64. */
int dart_set_state(DartStepper* h, const double* q, const double* dq);
int dart_get_state(DartStepper* h, double* q, double* dq);

/* Replaces one env.step(a) per env (hopper.py:36-65 / walker2d.py:22-65 incl. do_simulation, dart_env.py:158-175,
 * and TimeLimit.step, wrappers/time_limit.py:14-21).
 *   actions        (N, act_dim) float32   host
 *   obs_out        (N, obs_dim) float32   host
 *   reward_out     (N,)         float64   host   (reference rewards are numpy float64)
 *   done_out       (N,)         uint8     host   task done OR time-limit
 *   truncated_out  (N,)         uint8     host   info['TimeLimit.truncated'] (may be NULL)
 * Without DART_CFG_AUTORESET the caller resets done envs with dart_reset(mask, ...). */
int dart_step(DartStepper* h, const float* actions, float* obs_out, double* reward_out, uint8_t* done_out,
              uint8_t* truncated_out);

/* Page-lock a caller-owned host buffer (hipHostRegister) so that dart_step / dart_step_async can DMA straight from / into it:
 * a C caller that keeps its action / observation / reward / done arrays alive across steps -- the usual way to bind this
 * library from the reference's side -- registers each of them ONCE (or one arena that holds them all) and
 * from then on every argument of dart_step that lies inside a registered range skips the library's pinned staging block and the
 * host memcpy behind it; float64 rewards are produced on the device.  Arguments outside every registered range keep the staging
 * path (correct for any pointer, ~2x slower at 65 536 envs: profiles/r04_host_path_ab.txt).  Ownership is explicit: the buffer
 * must stay mapped until dart_unregister_host_buffer or dart_destroy -- the library never registers memory on its own
 * (page-locking a buffer the caller may free behind its back is not something an ABI should do silently).  Actions are always
 * consumed before the call that takes them returns: dart_step (blocking) lets the kernel read a registered action array where it
 * lies; dart_step_async / dart_step_async_to copy the actions into the library's own pinned block at call time, so the caller may
 * refill its array between step_async and step_wait (round 5; round 4 read them in place after the call had returned). */
int dart_register_host_buffer(DartStepper* h, void* ptr, uint64_t bytes);
int dart_unregister_host_buffer(DartStepper* h, void* ptr);

/* VectorEnv.step_async / step_wait (reference gym/vector/vector_env.py:68-92): same as dart_step, split. */
int dart_step_async(DartStepper* h, const float* actions);
int dart_step_wait(DartStepper* h, float* obs_out, double* reward_out, uint8_t* done_out, uint8_t* truncated_out);

/* Device-resident variant for GPU learners / benchmarks: all pointers are HBM addresses on the handle's device,
 * reward is float32.  Enqueued on `hip_stream` (a hipStream_t, NULL = the handle's own stream); returns
 * without synchronising.  A caller-supplied stream is ordered against the handle's own stream with events in both
 * directions (it waits for work already enqueued on the handle, later calls on the handle wait for it), so mixing this
 * entry point with the host-buffer calls needs no extra synchronisation; the caller orders its own producers / consumers
 * of the argument buffers on `hip_stream`. */
int dart_step_device(DartStepper* h, const float* d_actions, float* d_obs, float* d_reward, uint8_t* d_done,
                     uint8_t* d_truncated, void* hip_stream);
int dart_reset_device(DartStepper* h, const uint8_t* d_mask, float* d_obs, void* hip_stream);

/* Per-env bookkeeping: steps since reset (TimeLimit._elapsed_steps, wrappers/time_limit.py:17,24) and the number of
 * on-device (Philox) resets so far.  Either pointer may be NULL. */
int dart_get_counters(DartStepper* h, int32_t* elapsed, uint32_t* episode);

/* Solver diagnostics: hist64[0..31] = number of wavefronts whose frictionless-stage pivoting loop ran k iterations,
 * hist64[32..63] the same for the friction stage.  Needs DART_CFG_STATS = 1. */
int dart_get_stats(DartStepper* h, uint64_t* hist64, int clear);

/* Debugging aid of the spatial kernel (needs DART_CFG_STATS): per env 160 doubles = {m, ncp, x[40], b[40], hi[40], diagA[40]}
 * of the last LCP solved. */
int dart_debug_dump(DartStepper* h, double* out160);


/* Zero-copy read access to the pinned host staging buffers that dart_step_wait / dart_step fill: (N, obs_dim) float32
 * observations, (N,) float32 rewards, (N,) done and truncated flags.  Valid until the next step / reset on this handle;
 * call dart_step_wait with NULL outputs and read these instead to skip the copy into caller memory (VectorEnv(copy=False),
 * reference gym/vector/sync_vector_env.py:83). */
int dart_host_views(DartStepper* h, const float** obs, const float** reward_f32, const uint8_t** done, const uint8_t** truncated);

/* Where the outputs of the last HOST-buffer step (dart_step / dart_step_async[_to] + dart_step_wait) still sit in HBM: the device block the
 * host copies were made from -- (N, obs_dim) float32, (N) float32, (N) u8, (N) u8 -- valid until the next step / reset on this handle.  For
 * a consumer on the device that follows a host-driven loop (the rollout all-gather of dart_env_amd/distributed.py::gather_rollout: "RCCL
 * only to gather rollouts") -- it need not upload again what the step just produced there. */
int dart_device_outputs(DartStepper* h, const float** d_obs, const float** d_reward_f32, const uint8_t** d_done, const uint8_t** d_truncated);

/* The outputs of a step as ONE caller-owned block -- the zero-staging form of the host-buffer path ("copied back once per
 * batched step", reference gym/vector/vector_env.py:68-92): dart_output_layout gives the block's size and the byte offsets of
 * obs (N, obs_dim) f32 / reward (N) f32 / done (N) u8 / truncated (N) u8 inside it; the LAST round256(8 N) bytes of the block hold
 * the rewards once more as (N) float64 -- the type the reference's API returns -- converted on the device (round 5; offset =
 * total_bytes - ((8 N + 255) & ~255)); dart_register_output page-locks a caller
 * buffer of that size (hipHostRegister; the caller keeps it alive until dart_unregister_output or dart_destroy);
 * dart_step_async_to enqueues H2D actions, kernel, auto-reset and a SINGLE D2H copy straight into the block;
 * dart_step_wait(h, NULL, NULL, NULL, NULL) then only synchronises.  A step's results stay valid for as long as the caller does
 * not hand the same block to another step: dart_env_amd/stepper.py rotates a small pool of blocks and reuses one only when the
 * caller holds no array of it any more, which keeps gym.vector's copy=True contract without a copy.
 *
 * Where the block's memory comes from (round 6): PREFER dart_alloc_output -- total_bytes of memory the driver itself page-locks
 * (hipHostMalloc), known to the handle as a block for dart_step_async_to, and OWNED BY THE CALLER: dart_destroy forgets it, dart_free_output
 * (no handle: arrays a binding handed out may outlive the handle) releases it.  dart_register_output, for a caller that must have the
 * results in memory of its own, locks pageable memory after the fact (hipHostRegister: a "userptr" mapping the driver keeps coherent with
 * the process's page tables) -- on this stack (ROCm 7.2, MI355X) a GPU write into such memory faulted about once in ten runs of this
 * repository's GPU suite ("Memory access fault ... Write access to a read-only page" at an address inside a registered numpy array of a
 * process that had forked; profiles/r06_crash_hunt.txt part 2), never into hipHostMalloc'ed memory; dart_env_amd/stepper.py moved to
 * dart_alloc_output for that reason.  The same caveat holds for dart_register_host_buffer. */
int dart_output_layout(const DartStepper* h, int64_t* total_bytes, int64_t* offsets4);
int dart_alloc_output(DartStepper* h, void** block);
int dart_free_output(void* block);
int dart_register_output(DartStepper* h, void* block);
int dart_unregister_output(DartStepper* h, void* block);   /* (of a dart_alloc_output block: the handle forgets it; it is not freed) */
int dart_step_async_to(DartStepper* h, const float* actions, void* block);

/* Per-env task state that reset_model draws besides (q, dq): the reach target of DartReacher-v1 / DartReacher3d-v1
 * (reference gym/envs/dart/reacher2d.py:53-59, reacher.py:50-55).  values = (N, 4) doubles, slots 0..2 = target x, y, z;
 * mask as in dart_reset.  Call it before dart_reset so that the reset observation sees the new target. */
int dart_set_task_state(DartStepper* h, const uint8_t* mask, const double* values);

/* External body force: `bodynodes[body].add_ext_force(F)` before every world step, the perturbation branch of
 * DartEnv.do_simulation (reference gym/envs/dart/dart_env.py:159-172).  force = (N, 3) world-frame vectors, applied at the
 * body frame origin (pydart2's default offset) in every substep until replaced; NULL switches it off.  Served by the tree
 * kernel's EXTRAS instantiation and by the planar register kernels' EXTRAS instantiation (Hopper, Walker2d, HalfCheetah, Snake);
 * DART_E_UNSUPPORTED on the lean tree kernel (HumanWalker, Walker3d: set card.generic_kernel = 1) and on the cart / arm / chain
 * lane kernels. */
int dart_set_ext_force(DartStepper* h, int body, const double* force);

/* Episode statistics kept on the device (DART_CFG_EPISODE_STATS = 1) -- the batched form of the reference's
 * RecordEpisodeStatistics wrapper (gym/wrappers/record_episode_statistics.py:22-34): every step adds the reward to the
 * env's running return and 1 to its length; an env that reports done latches them into last_return[i] /
 * last_length[i] (the wrapper's info['episode'] 'r' and 'l') and restarts; dart_reset restarts the masked envs.
 * totals3 = {sum of returns, sum of lengths, count} over the episodes finished since the last clear.
 * Any output pointer may be NULL. */
int dart_get_episode_stats(DartStepper* h, double* last_return, int32_t* last_length, double* totals3, int clear_totals);

/* Dynamics quantities of the current state of every env -- pydart2's `skel.M` and `skel.c` (reference
 * gym/envs/dart/walker3d_spd.py:40-55 builds its SPD controller from them): mass_matrix (N, ndofs, ndofs) symmetric,
 * WITHOUT the implicit damping / stiffness terms the integrator adds; coriolis_gravity (N, ndofs) = C(q, dq) dq + g(q).
 * Either pointer may be NULL.  Computed by the generic tree kernel for every model (planar ones included).  For a FreeJoint root
 * (dog.skel) both are in DART's coordinates -- dq[0:6] is the body-frame twist: M = T^T M_int T, c = T^T (c_int + M_int Tdot dq)
 * with T = blockdiag(R, R, I) mapping the twist to the world-frame rates of the kernel's internal root chain (round 4). */
int dart_get_dynamics(DartStepper* h, double* mass_matrix, double* coriolis_gravity);

/* World poses of every body of every env in its current state -- pydart2's `bodynode.T` / `.C` / `.com()` as the task code
 * reads them (reference gym/envs/dart/hopper.py:42,72 `bodynodes[2].com()[1]`; human_walker.py:78-92 `bodynodes[1].com()`,
 * `bodynode('head').com()`, `.to_world()`): rotation (N, nbodies, 3, 3) row-major body-to-world, origin (N, nbodies, 3) of the
 * body frame, com (N, nbodies, 3).  Body order = the card's (the .skel file's).  Any pointer may be NULL (not all). */
int dart_get_body_poses(DartStepper* h, double* rotation, double* origin, double* com);

/* Contacts of the last world step of the last env-step -- pydart2's `world.collision_result.contacts` as the reference
 * reads it (gym/envs/dart/walker2d.py:38-41: contact.force; human_walker.py:97-106: contact.bodynode1 / bodynode2;
 * dart_env.py has no getter of its own).  Needs DART_CFG_CONTACT_REPORT = 1 before the step.
 *   count (N)                      contacts of env i (may exceed max_contacts; only the first max_contacts are returned)
 *   bodies (N, max_contacts, 2)    card body indices {a, b}; b = -1 for the ground skeleton; unused slots -1
 *   point_force (N, max_contacts, 6) world contact point, then the world force the contact applies to body a
 *                                  (= (n l_n + t1 l_1 + t2 l_2) / dt of DART's ContactConstraint; body b receives the opposite)
 * bodies / point_force may be NULL.  Order: ground contacts in shape order (box vertices in face order), then link-link
 * pairs.  Implemented by the tree kernel (its reporting instantiation) and by the planar register kernels (EXTRAS
 * instantiation: one record per contact slot); DART_E_UNSUPPORTED on the cart / arm / chain lane kernels, whose models touch
 * nothing. */
int dart_get_contacts(DartStepper* h, int32_t* count, int32_t* bodies, double* point_force, int32_t max_contacts);

/* pydart2 `skel.constraint_forces()` after the last world step (reference gym/envs/dart/walker3d_spd.py:51 reads it for
 * its SPD controller): the generalized force J^T lambda / dt of all contact, joint-limit and joint-friction rows,
 * (N, ndofs).  Recorded together with the contacts: needs DART_CFG_CONTACT_REPORT = 1 before the step. */
int dart_get_constraint_forces(DartStepper* h, double* constraint_forces);

/* Exact checkpoint of everything that persists between steps: q / dq in the kernel's own precision, TimeLimit and episode
 * counters, the MT19937 bank, per-env task state (reach targets ...), episode statistics -- the batched form of pickling
 * a reference env between steps (set_state / get_state are the reference's only checkpoint hooks,
 * dart_env.py:145-148,211-215; they do not cover np_random or TimeLimit).  dart_snapshot(h, NULL, &n) returns the size;
 * dart_restore accepts a snapshot of a handle with the same card, num_envs, precision and enabled modes, after which the
 * following steps (incl. auto-resets) are bitwise those of the original run. */
int dart_snapshot(DartStepper* h, void* buf, uint64_t* nbytes);
int dart_restore(DartStepper* h, const void* buf, uint64_t nbytes);

/* Wait for everything enqueued on the handle's stream. */
int dart_sync(DartStepper* h);

/* Times `steps` back-to-back dart_step_device launches on the handle's stream with HIP events
 * (used by bench.py for the live roofline figure).  d_actions holds `action_batches` consecutive (N, act_dim)
 * batches that are cycled.  Returns average kernel-to-kernel milliseconds per step in *ms_per_step. */
int dart_time_steps(DartStepper* h, const float* d_actions, int action_batches, float* d_obs, float* d_reward,
                    uint8_t* d_done, uint8_t* d_truncated, int steps, double* ms_per_step);

/* The same stopwatch in two halves, for a caller that brackets its own wall-clock region around the launches (bench.py):
 * dart_timer_mark(h, 0) / (h, 1) enqueue a HIP event on the handle's stream before / after the work (no wait);
 * dart_timer_elapsed waits for the second event and returns the milliseconds between the two. */
int dart_timer_mark(DartStepper* h, int which);
int dart_timer_elapsed(DartStepper* h, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* DART_STEPPER_H */
