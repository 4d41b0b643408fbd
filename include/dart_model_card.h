/* dart_model_card.h -- flat, POD description of one articulated model + task.
 *
 * This is the data half of the C ABI (the function half is dart_stepper.h).
 * It carries what the reference obtains implicitly by handing a .skel path to
 * pydart2 (reference gym/envs/dart/dart_env.py:55,62-67: load world, take the
 * last skeleton, enforce every finite joint limit) plus the per-task constants
 * hard-coded in the reference's env constructors (hopper.py:8-12,
 * walker2d.py:8-12).  Filled by dart_env_amd/skel.py (model compiler).
 *
 * All lengths in metres, angles in radians, row-major matrices, doubles.
 */
#ifndef DART_MODEL_CARD_H
#define DART_MODEL_CARD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DART_CARD_VERSION 2   /* 2: impulse_inertia re-encoded so that a zero-initialised card gets the default */
#define DART_MAX_BODIES 32
#define DART_MAX_DOFS 32
#define DART_MAX_SHAPES 32
#define DART_MAX_ACTIONS 32

/* joint types (DART joint classes named in the .skel `type=` attribute) */
enum {
  DART_JT_WELD = 0,
  DART_JT_PRISMATIC = 1,
  DART_JT_REVOLUTE = 2,
  DART_JT_TRANSLATIONAL = 3,
  DART_JT_EULER_XYZ = 4,
  DART_JT_EULER_ZYX = 5,
  DART_JT_UNIVERSAL = 6,
  DART_JT_FREE = 7
};

/* collision shape types */
enum { DART_SH_CAPSULE = 0, DART_SH_BOX = 1, DART_SH_SPHERE = 2, DART_SH_ELLIPSOID = 3, DART_SH_CYLINDER = 4 };

/* task epilogues (reward / done / observation computed on device) */
enum {
  DART_TASK_NONE = 0,      /* physics only: obs = [q, dq], reward 0, done 0 */
  DART_TASK_HOPPER = 1,    /* reference gym/envs/dart/hopper.py:36-74   */
  DART_TASK_WALKER2D = 2,  /* reference gym/envs/dart/walker2d.py:22-74 */
  DART_TASK_WALKER3D = 3,  /* reference gym/envs/dart/walker3d.py:33-113 */
  DART_TASK_HUMANWALKER = 4, /* reference gym/envs/dart/human_walker.py:60-165 */
  DART_TASK_CARTPOLE = 5,    /* reference gym/envs/dart/cart_pole.py:12-39: obs [q, dq], reward 1, done |q[1]| > angle_max,
                                 action NOT clamped (act_low/high = -/+inf), tau[0] = a[0] * 100 */
  DART_TASK_HALFCHEETAH = 6, /* reference gym/envs/dart/half_cheetah.py:28-93: reward dx/dt + 1 - 0.1 sum a^2 (0 if the
                                 state broke), done adds |q[2]| >= angle_max (1.3), obs q[1:], dq (unclipped) */
  DART_TASK_CARTPOLE_SWINGUP = 7, /* reference gym/envs/dart/cartpole_swingup.py:14-33: obs [q, dq], no clamp, reward
                                 aux_real[0] - |q[1]| - aux_real[1] sum a^2 - aux_real[2] |q[0]|, done |q[1]| > aux_real[3]
                                 or |dq[1]| > aux_real[4] or |q[0]| > aux_real[5] */
  DART_TASK_SNAKE = 9,       /* reference gym/envs/dart/snake_7link.py:35-96: before every world step each body gets the fluid
                                 force -aux_real[3] (v_com . n) n at its frame origin, n = the body's z axis (:37-47); reward
                                 dx/dt + aux_real[0] - aux_real[1] sum a^2 - aux_real[2] |q[2]|, done |q[2]| >= angle_max or a
                                 broken state, obs q[1:], dq */
  DART_TASK_REACHER2D = 10,  /* reference gym/envs/dart/reacher2d.py:18-45 (DartReacher-v1): per-env target (task state
                                 [0..2]); tip = to_world(aux_body[0], aux_real[0..2]); reward -|tip - target| - sum a^2 after
                                 the step, never done; obs cos q, sin q, target x z, dq, tip - target */
  DART_TASK_REACHER3D = 11,  /* reference gym/envs/dart/reacher.py:13-45 (DartReacher3d-v1): reward -|tip - target| -
                                 aux_real[3] sum tau^2 with the tip BEFORE the step, done when that distance <= aux_real[4]
                                 or the state is not finite; obs cos q, sin q, target, dq, tip - target (after) */
  DART_TASK_WALKER3D_SPD = 12, /* reference gym/envs/dart/walker3d_spd.py:40-113 (DartWalker3dSPD-v1): the action is a target pose
                                 (a + 1) / 2 mapped onto each joint's limits; before EVERY world step the torque comes from the
                                 stable-PD law with M, c and the previous step's constraint forces (:40-55, env dt in the law),
                                 clipped to act_scale; reward aux_real[1] dx/dt + 1 - aux_real[2] sum a^2 - aux_real[3] |z|;
                                 done as Walker3d with angle_max 0.54; obs q[1:], clip(dq) */
  DART_TASK_DOG = 13,        /* reference gym/envs/dart/dog.py:28-46 (DartDog-v1): free root joint; reward aux_real[0] dx/dt + 1 -
                                 ctrl_cost sum a^2 on bodynodes[0]'s COM, done on height outside (height_lo, height_hi), side
                                 deviation >= aux_real[1] or a broken state; obs q[1:], clip(dq) in DART's FreeJoint coordinates */
  DART_TASK_DOUBLE_PENDULUM = 8 /* reference gym/envs/dart/inverted_double_pendulum.py:19-53: obs [q0, sin q1..2, cos q1..2,
                                 dq], height = 2 (y(aux_body[1]) - y(aux_body[0]) - aux_real[4]) / aux_real[5], reward
                                 aux_real[0] - (aux_real[1] q0^2 + (height - 2)^2) - (aux_real[2] dq1^2 + aux_real[3] dq2^2),
                                 done height <= 1 */
};

typedef struct DartModelCard {
  int32_t version;       /* DART_CARD_VERSION */
  int32_t struct_bytes;  /* sizeof(DartModelCard) as seen by the caller */
  char name[32];

  /* ---- world ---- */
  double dt;          /* physics time step (dart_env.py:29,55) */
  double gravity[3];
  double ground_y;    /* top face of the immobile ground box; -inf = no floor */
  double friction;    /* Coulomb mu of foot/ground pair (DART default 1.0) */
  double erp;         /* contact error-reduction parameter (DART 0.01) */
  double max_erv;     /* cap of the contact correction velocity (DART ContactConstraint.cpp: DART_MAX_ERV 1e-3) */
  double cfm;         /* diag(A) *= 1 + cfm on joint-limit / joint-friction rows (DART JointLimitConstraint.cpp: DART_CFM 1e-9) */
  double limit_erp;   /* joint-limit correction gain (DART 6: effectively 0) */

  /* ---- bodies (creation order = joint order in file, parents first) ---- */
  int32_t nbodies;
  int32_t ndofs;
  int32_t parent[DART_MAX_BODIES];   /* -1 = world */
  int32_t jtype[DART_MAX_BODIES];
  int32_t dof_offset[DART_MAX_BODIES];
  int32_t ndof[DART_MAX_BODIES];
  double mass[DART_MAX_BODIES];
  double com[DART_MAX_BODIES][3];      /* body frame */
  double inertia[DART_MAX_BODIES][9];  /* about COM, body axes */
  double T_pj[DART_MAX_BODIES][16];    /* joint frame in parent body frame (4x4) */
  double T_cj[DART_MAX_BODIES][16];    /* joint frame in child body frame (4x4) */
  double axes[DART_MAX_BODIES][9];     /* rows: axis k in joint frame */

  /* ---- degrees of freedom ---- */
  double lower[DART_MAX_DOFS];
  double upper[DART_MAX_DOFS];
  int32_t limited[DART_MAX_DOFS];      /* finite limit -> enforced (dart_env.py:64-67) */
  double damping[DART_MAX_DOFS];
  double stiffness[DART_MAX_DOFS];
  double rest[DART_MAX_DOFS];
  double init_pos[DART_MAX_DOFS];
  double init_vel[DART_MAX_DOFS];

  /* ---- collision shapes of the robot (vs. ground plane) ---- */
  int32_t nshapes;
  int32_t shape_body[DART_MAX_SHAPES];
  int32_t shape_type[DART_MAX_SHAPES];
  int32_t shape_collidable[DART_MAX_SHAPES];
  double shape_pose[DART_MAX_SHAPES][16];  /* in body frame */
  double shape_size[DART_MAX_SHAPES][3];   /* capsule: radius,height,0; box: extents */

  /* ---- task constants ---- */
  int32_t task;             /* DART_TASK_* */
  int32_t frame_skip;       /* hopper.py:12 */
  int32_t act_dim;
  int32_t obs_dim;
  int32_t act_dof0;         /* first actuated dof: tau[act_dof0 + k] = clamp(a_k) * scale_k */
  int32_t max_episode_steps;/* TimeLimit (gym/envs/__init__.py:210,269) */
  int32_t height_body;      /* bodynodes[k].com()[1] used for height (hopper.py:42) */
  int32_t penalty_dof;      /* hopper.py:46 `for j in [-2]` resolved to an index; -1 = none */
  double act_scale[DART_MAX_ACTIONS];
  double act_low[DART_MAX_ACTIONS];
  double act_high[DART_MAX_ACTIONS];
  double alive_bonus;       /* 1.0 */
  double ctrl_cost;         /* 1e-3 */
  double limit_penalty;     /* 0.5 * 1.5 per side within penalty_margin (hopper.py:45-58) */
  double penalty_margin;    /* 0.05 */
  double height_lo, height_hi, angle_max;   /* done thresholds (hopper.py:60-62) */
  double state_abs_max;     /* 100 */
  double obs_vel_clip;      /* 10 */
  double reset_noise;       /* 0.005 position noise (hopper.py:78) */
  double reset_noise_vel;   /* velocity noise: 0.005 (hopper.py:79), 0.05 for HumanWalker (human_walker.py:154) */
  /* task-specific extras.  HumanWalker: aux_body = {progress body (bodynodes[1]), head, l-foot, r-foot},
   * aux_real = {target_vel, alive_bonus 2.0, action_pen 0.5, deviation_pen 3.0, height band lo -0.2, hi 1.0,
   *             |q[3]| max 1.3, |q[5]| max 0.4}; angle_max = 2.0 (up/forward), state_abs_max etc. as above;
   * aux_real2 = {side deviation max 0.9}.
   * Walker3d: aux_body = {bodynodes[0] (progress / height / side / up-forward angles), penalty dof a, penalty dof b}
   * (walker3d.py:68 `for j in [-3, -9]` resolved to dof indices), limit_penalty = 0.2 * 1.5 per side,
   * aux_real = {deviation_pen 1e-3}; reward is zeroed when done (walker3d.py:91-92). */
  int32_t aux_body[4];
  double aux_real[8];
  double aux_real2[4];
  /* diag(A) *= 1 + contact_cfm on CONTACT rows (normal + friction): DART's ContactConstraint.cpp carries its own
   * DART_CFM 1e-5 (four orders above the joint-limit value) -- models whose feet are boxes produce up to 4 coplanar,
   * redundant contact points per rigid foot, and this is what keeps their Delassus matrix regular. */
  double contact_cfm;
  /* 1: link-link contacts between the robot's own collision shapes (boxes) are generated for every pair of bodies
   * that are not parent and child -- `robot_skeleton.set_self_collision_check(True)` (walker3d.py:26) with DART's
   * default of skipping adjacent bodies. */
  int32_t self_collision;
  /* 1: always use the generic tree kernel, even if a faster specialised kernel matches the model (needed for features
   * only the generic kernel has: external body forces, link-link contacts, contacts on every shape). */
  int32_t generic_kernel;
  /* Coulomb friction of each dof (<dynamics><friction> in the .skel; only reacher2d.skel has non-zero values): DART's
   * JointCoulombFrictionConstraint -- an LCP row that drives the joint velocity to zero with an impulse bounded by
   * +-joint_friction * dt. */
  double joint_friction[DART_MAX_DOFS];
  /* DART_TASK_WALKER3D_SPD: stable-PD gains per dof (walker3d_spd.py:13-18); act_scale[k] holds the torque limit of
   * actuated dof k (:20-22) for that task. */
  double spd_kp[DART_MAX_DOFS];
  double spd_kd[DART_MAX_DOFS];
  /* Which joint-space inertia the IMPULSE pass runs on.  DART integrates joint damping and
   * springs implicitly in the forward dynamics only: (M + dt D + dt^2 K) qdd = rhs uses the `...Implicit` articulated
   * inertias (BodyNode::updateBiasForce / updateAccelerationFD, GenericJoint::updateInvProjArtInertiaImplicit), while
   * the constraint solver's unit-impulse tests and the final velocity change go through BodyNode::updateBiasImpulse /
   * updateVelocityChangeFD -> GenericJoint::updateVelocityChangeDynamic, which read getArticulatedInertia() /
   * getInvProjArtInertia(): the plain mass matrix.
   *   DART_IMPULSE_MASS (0, the default -- what a memset card gets):  A = J M^-1 J^T,  dq = dq* + M^-1 J^T lambda   (DART 6)
   *   DART_IMPULSE_AUGMENTED (1):          A = J H^-1 J^T,  dq = dq* + H^-1 J^T lambda, H = M + dt D + dt^2 K
   *                                        (what rounds 1-2 of this build used for both passes)
   * dart_create rejects any other value (DART_E_INVALID). */
  int32_t impulse_inertia;
} DartModelCard;

enum { DART_IMPULSE_MASS = 0, DART_IMPULSE_AUGMENTED = 1 };

#ifdef __cplusplus
}
#endif
#endif /* DART_MODEL_CARD_H */
