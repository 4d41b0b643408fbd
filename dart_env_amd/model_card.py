"""ctypes mirror of ``include/dart_model_card.h`` + the per-task constants.

The task constants restate what the reference hard-codes in its env
constructors and step functions:

* DartHopper-v1   -- reference gym/envs/dart/hopper.py:8-12 (bounds +-1, scale 200,
  obs 11, frame_skip 4), :45-58 (reward), :60-62 (done), gym/envs/__init__.py:206-211
  (max_episode_steps 1000)
* DartWalker2d-v1 -- reference gym/envs/dart/walker2d.py:8-12 (scale
  [100,100,20,100,100,20], obs 17), :43-47 (reward), :60-61 (done),
  gym/envs/__init__.py:265-270
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .skel import MAX_BODIES, MAX_DOFS, MAX_SHAPES, ModelCard, parse_skel

MAX_ACTIONS = 32
CARD_VERSION = 2

TASK_NONE, TASK_HOPPER, TASK_WALKER2D, TASK_WALKER3D, TASK_HUMANWALKER, TASK_CARTPOLE, TASK_HALFCHEETAH = 0, 1, 2, 3, 4, 5, 6
TASK_CARTPOLE_SWINGUP, TASK_DOUBLE_PENDULUM, TASK_SNAKE, TASK_REACHER2D, TASK_REACHER3D = 7, 8, 9, 10, 11
TASK_WALKER3D_SPD, TASK_DOG = 12, 13


class DartModelCard(C.Structure):
    _fields_ = [
        ("version", C.c_int32), ("struct_bytes", C.c_int32), ("name", C.c_char * 32),
        ("dt", C.c_double), ("gravity", C.c_double * 3), ("ground_y", C.c_double),
        ("friction", C.c_double), ("erp", C.c_double), ("max_erv", C.c_double), ("cfm", C.c_double),
        ("limit_erp", C.c_double),
        ("nbodies", C.c_int32), ("ndofs", C.c_int32),
        ("parent", C.c_int32 * MAX_BODIES), ("jtype", C.c_int32 * MAX_BODIES),
        ("dof_offset", C.c_int32 * MAX_BODIES), ("ndof", C.c_int32 * MAX_BODIES),
        ("mass", C.c_double * MAX_BODIES), ("com", (C.c_double * 3) * MAX_BODIES),
        ("inertia", (C.c_double * 9) * MAX_BODIES), ("T_pj", (C.c_double * 16) * MAX_BODIES),
        ("T_cj", (C.c_double * 16) * MAX_BODIES), ("axes", (C.c_double * 9) * MAX_BODIES),
        ("lower", C.c_double * MAX_DOFS), ("upper", C.c_double * MAX_DOFS),
        ("limited", C.c_int32 * MAX_DOFS), ("damping", C.c_double * MAX_DOFS),
        ("stiffness", C.c_double * MAX_DOFS), ("rest", C.c_double * MAX_DOFS),
        ("init_pos", C.c_double * MAX_DOFS), ("init_vel", C.c_double * MAX_DOFS),
        ("nshapes", C.c_int32), ("shape_body", C.c_int32 * MAX_SHAPES),
        ("shape_type", C.c_int32 * MAX_SHAPES), ("shape_collidable", C.c_int32 * MAX_SHAPES),
        ("shape_pose", (C.c_double * 16) * MAX_SHAPES), ("shape_size", (C.c_double * 3) * MAX_SHAPES),
        ("task", C.c_int32), ("frame_skip", C.c_int32), ("act_dim", C.c_int32), ("obs_dim", C.c_int32),
        ("act_dof0", C.c_int32), ("max_episode_steps", C.c_int32), ("height_body", C.c_int32),
        ("penalty_dof", C.c_int32),
        ("act_scale", C.c_double * MAX_ACTIONS), ("act_low", C.c_double * MAX_ACTIONS),
        ("act_high", C.c_double * MAX_ACTIONS),
        ("alive_bonus", C.c_double), ("ctrl_cost", C.c_double), ("limit_penalty", C.c_double),
        ("penalty_margin", C.c_double), ("height_lo", C.c_double), ("height_hi", C.c_double),
        ("angle_max", C.c_double), ("state_abs_max", C.c_double), ("obs_vel_clip", C.c_double),
        ("reset_noise", C.c_double), ("reset_noise_vel", C.c_double),
        ("aux_body", C.c_int32 * 4), ("aux_real", C.c_double * 8), ("aux_real2", C.c_double * 4),
        ("contact_cfm", C.c_double), ("self_collision", C.c_int32), ("generic_kernel", C.c_int32),
        ("joint_friction", C.c_double * MAX_DOFS), ("spd_kp", C.c_double * MAX_DOFS), ("spd_kd", C.c_double * MAX_DOFS),
        ("impulse_inertia", C.c_int32),
    ]


@dataclass
class TaskSpec:
    """Per-env constants (see module docstring for the reference lines)."""
    env_id: str
    model: str                      # model-card name under dart_env_amd/models
    task: int
    frame_skip: int
    act_dim: int
    obs_dim: int
    act_dof0: int
    act_scale: List[float]
    max_episode_steps: int
    reward_threshold: Optional[float]
    height_body: int
    penalty_dof: int                # -1 = no joint-limit penalty
    height_lo: float
    height_hi: float
    angle_max: float
    contact_bodies: List[str] = field(default_factory=list)
    alive_bonus: float = 1.0
    ctrl_cost: float = 1e-3
    limit_penalty: float = 0.5
    penalty_margin: float = 0.05
    state_abs_max: float = 100.0
    obs_vel_clip: float = 10.0
    reset_noise: float = 0.005
    reset_noise_vel: float = 0.005
    aux_body_names: List[str] = field(default_factory=list)
    aux_ints: List[int] = field(default_factory=list)   # stored in aux_body[] after the named bodies
    aux_real: List[float] = field(default_factory=list)
    aux_real2: List[float] = field(default_factory=list)
    contact_cfm: Optional[float] = None   # None -> the model's (DART ContactConstraint's 1e-5)
    act_low: float = -1.0
    act_high: float = 1.0
    clamp_actions: bool = True            # False: the env passes a[k] * scale on unclamped (cart_pole.py:16)
    all_bodies_collide: bool = False      # every collision shape of the robot vs. the ground (half_cheetah)
    physics_dt: float = 0.002             # DartEnv.__init__'s dt argument (dart_env.py:29)
    self_collision: bool = False          # robot_skeleton.set_self_collision_check(True) (walker3d.py:26)
    spd_kp: List[float] = field(default_factory=list)   # stable-PD gains per dof (walker3d_spd.py:13-18); kd = kp / 10


HOPPER = TaskSpec(
    env_id="DartHopper-v1", model="hopper", task=TASK_HOPPER, frame_skip=4, act_dim=3, obs_dim=11,
    act_dof0=3, act_scale=[200.0] * 3, max_episode_steps=1000, reward_threshold=3800.0,
    height_body=2, penalty_dof=4, height_lo=0.7, height_hi=1.8, angle_max=0.2,
    contact_bodies=["h_foot"], all_bodies_collide=True)

WALKER2D = TaskSpec(
    env_id="DartWalker2d-v1", model="walker2d", task=TASK_WALKER2D, frame_skip=4, act_dim=6, obs_dim=17,
    act_dof0=3, act_scale=[100.0, 100.0, 20.0, 100.0, 100.0, 20.0], max_episode_steps=1000,
    reward_threshold=None, height_body=2, penalty_dof=-1, height_lo=0.8, height_hi=2.0, angle_max=1.0,
    contact_bodies=["h_foot", "h_foot_left"], all_bodies_collide=True)

# DartHumanWalker-v1 -- reference gym/envs/dart/human_walker.py:16-30 (23 actions, scale*1.5, obs 57+2, frame_skip 15),
# :109-128 (reward, done), :150-165 (reset: velocity noise 0.05), gym/envs/__init__.py:284-288 (300 steps)
HUMANWALKER = TaskSpec(
    env_id="DartHumanWalker-v1", model="humanwalker", task=TASK_HUMANWALKER, frame_skip=15, act_dim=23, obs_dim=59,
    act_dof0=6,
    act_scale=[1.5 * v for v in [120, 120, 120, 100, 60, 60, 120, 120, 120, 100, 60, 60, 100, 100, 100, 80, 80, 80,
                                 50, 80, 80, 80, 50]],
    max_episode_steps=300, reward_threshold=None, height_body=10, penalty_dof=-1, height_lo=-0.2, height_hi=1.0,
    angle_max=2.0, contact_bodies=["l-foot", "r-foot"], alive_bonus=2.0, ctrl_cost=0.5, limit_penalty=0.0,
    reset_noise=0.005, reset_noise_vel=0.05, aux_body_names=["pelvis", "head", "l-foot", "r-foot"],
    aux_real=[1.0, 2.0, 0.5, 3.0, -0.2, 1.0, 1.3, 0.4], aux_real2=[0.9], all_bodies_collide=True)

# DartWalker3d-v1 -- reference gym/envs/dart/walker3d.py:10-15 (15 actions, scale 100 / 150 for the waist / 20 for the
# ankles, obs 41, frame_skip 4), :45-92 (reward, done; quantities of bodynodes[0] = the translational carrier body),
# :68 (limit penalty on q[-3], q[-9] = the two knees), gym/envs/__init__.py:272-276 (1000 steps)
WALKER3D = TaskSpec(
    env_id="DartWalker3d-v1", model="walker3d", task=TASK_WALKER3D, frame_skip=4, act_dim=15, obs_dim=41, act_dof0=6,
    act_scale=[150.0] * 3 + [100.0] * 4 + [20.0] * 2 + [100.0] * 4 + [20.0] * 2,
    max_episode_steps=1000, reward_threshold=None, height_body=0, penalty_dof=-1, height_lo=1.05, height_hi=2.0,
    angle_max=0.84, contact_bodies=["h_foot", "h_foot_left"], alive_bonus=1.0, ctrl_cost=1e-3, limit_penalty=0.2,
    aux_body_names=["h_torso_aux"], aux_ints=[18, 12], aux_real=[1e-3], all_bodies_collide=True,
    self_collision=True)

# DartCartPole-v1 -- reference gym/envs/dart/cart_pole.py:6-39 (dt 0.02, frame_skip 2, obs [q, dq], scale 100, no clamp,
# reward 1, done |q[1]| > 0.2 or non-finite obs, reset noise +-0.01), gym/envs/__init__.py:220-225
CARTPOLE = TaskSpec(
    env_id="DartCartPole-v1", model="cartpole", task=TASK_CARTPOLE, frame_skip=2, act_dim=1, obs_dim=4, act_dof0=0,
    act_scale=[100.0], max_episode_steps=1000, reward_threshold=950.0, height_body=0, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=0.2, alive_bonus=1.0, ctrl_cost=0.0, limit_penalty=0.0,
    state_abs_max=np.inf, obs_vel_clip=np.inf, reset_noise=0.01, reset_noise_vel=0.01, clamp_actions=False,
    physics_dt=0.02)

# DartHalfCheetah-v1 -- reference gym/envs/dart/half_cheetah.py:5-114 (dt 0.01, frame_skip 5, scale [120,90,60,120,60,30],
# obs q[1:], dq (17), reward dx/dt + 1 - 0.1 sum a^2, done on |s[2:]| >= 100 or |q[2]| >= 1.3), __init__.py:213-218
HALFCHEETAH = TaskSpec(
    env_id="DartHalfCheetah-v1", model="halfcheetah", task=TASK_HALFCHEETAH, frame_skip=5, act_dim=6, obs_dim=17,
    act_dof0=3, act_scale=[120.0, 90.0, 60.0, 120.0, 60.0, 30.0], max_episode_steps=1000, reward_threshold=4800.0,
    height_body=2, penalty_dof=-1, height_lo=-np.inf, height_hi=np.inf, angle_max=1.3, alive_bonus=1.0, ctrl_cost=0.1,
    limit_penalty=0.0, obs_vel_clip=np.inf, all_bodies_collide=True, physics_dt=0.01)

# DartCartPoleSwingUp-v1 -- reference gym/envs/dart/cartpole_swingup.py:7-47 (dt 0.01, frame_skip 2, scale 40, no clamp,
# reward 6 - |ang| - 0.01 sum a^2 - 0.01 |x|, done |ang| > 8 pi or |dang| > 25 or |x| > 5; reset: q + U(+-0.1),
# dq + U(+-0.01), then the pole hangs down: q[1] +- pi by a third draw), gym/envs/__init__.py:260-264 (500 steps)
CARTPOLE_SWINGUP = TaskSpec(
    env_id="DartCartPoleSwingUp-v1", model="cartpole_swingup", task=TASK_CARTPOLE_SWINGUP, frame_skip=2, act_dim=1,
    obs_dim=4, act_dof0=0, act_scale=[40.0], max_episode_steps=500, reward_threshold=None, height_body=0, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=np.inf, state_abs_max=np.inf, obs_vel_clip=np.inf, reset_noise=0.1,
    reset_noise_vel=0.01, clamp_actions=False, physics_dt=0.01,
    aux_real=[6.0, 0.01, 0.01, 8 * np.pi, 25.0, 5.0])

# DartDoubleInvertedPendulumEnv-v1 -- reference gym/envs/dart/inverted_double_pendulum.py:9-65 (dt 0.01, frame_skip 2,
# scale 40, no clamp, obs 8; reset: q + U(+-0.1), dq + 0.1 randn), gym/envs/__init__.py:227-231 (1000 steps)
DOUBLE_PENDULUM = TaskSpec(
    env_id="DartDoubleInvertedPendulumEnv-v1", model="double_pendulum", task=TASK_DOUBLE_PENDULUM, frame_skip=2, act_dim=1,
    obs_dim=8, act_dof0=0, act_scale=[40.0], max_episode_steps=1000, reward_threshold=None, height_body=0, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=np.inf, state_abs_max=np.inf, obs_vel_clip=np.inf, reset_noise=0.1,
    reset_noise_vel=0.1, clamp_actions=False, physics_dt=0.01, aux_body_names=["cart", "weight"],
    aux_real=[10.0, 0.01, 1e-3, 5e-3, 0.02, 0.6])

# DartSnake7Link-v1 -- reference gym/envs/dart/snake_7link.py:7-124 (7 capsule links sliding in the x-z plane 1 mm above the
# floor, frame_skip 4, scale 200, fluid forces on every body before every world step, reward dx/dt + 0.1 - 1e-3 sum a^2
# - 0.1 |q[2]|, done |q[2]| >= 1.5, obs q[1:], dq), gym/envs/__init__.py:290-294
SNAKE = TaskSpec(
    env_id="DartSnake7Link-v1", model="snake7link", task=TASK_SNAKE, frame_skip=4, act_dim=6, obs_dim=17, act_dof0=3,
    act_scale=[200.0] * 6, max_episode_steps=1000, reward_threshold=None, height_body=2, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=1.5, obs_vel_clip=np.inf, all_bodies_collide=True,
    aux_real=[0.1, 1e-3, 0.1, 50.0])

# DartReacher-v1 -- reference gym/envs/dart/reacher2d.py:5-66 (2-dof arm about y, dt 0.01 x frame_skip 2, scale 200,
# nothing collides (:11-14), target resampled in reset_model by rejection (:53-57)), gym/envs/__init__.py:233-238 (50 steps)
REACHER2D = TaskSpec(
    env_id="DartReacher-v1", model="reacher2d", task=TASK_REACHER2D, frame_skip=2, act_dim=2, obs_dim=11, act_dof0=0,
    act_scale=[200.0, 200.0], max_episode_steps=50, reward_threshold=-3.75, height_body=0, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=np.inf, state_abs_max=np.inf, obs_vel_clip=np.inf, reset_noise=0.01,
    reset_noise_vel=0.005, physics_dt=0.01, contact_bodies=[], aux_body_names=["link2"], aux_real=[0.0, 0.0, 0.0, 1.0, 0.0])

# DartReacher3d-v1 -- reference gym/envs/dart/reacher.py:5-61 (5-dof arm, dt 0.002 x 4, scale 10, fingertip (0,-0.25,0) on
# bodynodes[2], reward and done use the distance BEFORE the step (:23-35)), gym/envs/__init__.py:240-245 (500 steps)
REACHER3D = TaskSpec(
    env_id="DartReacher3d-v1", model="reacher3d", task=TASK_REACHER3D, frame_skip=4, act_dim=5, obs_dim=21, act_dof0=0,
    act_scale=[10.0] * 5, max_episode_steps=500, reward_threshold=-200.0, height_body=0, penalty_dof=-1,
    height_lo=-np.inf, height_hi=np.inf, angle_max=np.inf, state_abs_max=np.inf, obs_vel_clip=np.inf, reset_noise=0.01,
    reset_noise_vel=0.01, contact_bodies=[], aux_body_names=["link 3"], aux_real=[0.0, -0.25, 0.0, 0.001, 0.1])

# DartWalker3dSPD-v1 -- reference gym/envs/dart/walker3d_spd.py:9-138: the Walker3d model driven by a stable-PD controller
# that runs before every world step; kp per dof as the reference indexes it (300 on the root translation, 30 on dofs 7-8 and 13-14, else 100), kd = kp / 10, torque limits
# 200 / 100 / 20 (:20-22, stored in act_scale), reward 0.45 dx/dt + 1 - 1e-2 sum a^2 - 0.1 |z| (:96-102), angles < 0.54
# kp_diag = [0]*6 + [100]*15; kp_diag[0:3] = 300; kp_diag[7:9] = 30; kp_diag[13:15] = 30  -- indices of the 21-vector (:13-16)
_SPD_KP = [300.0] * 3 + [0.0] * 3 + [100.0] + [30.0] * 2 + [100.0] * 4 + [30.0] * 2 + [100.0] * 6
WALKER3D_SPD = TaskSpec(
    env_id="DartWalker3dSPD-v1", model="walker3d", task=TASK_WALKER3D_SPD, frame_skip=4, act_dim=15, obs_dim=41, act_dof0=6,
    act_scale=[100.0] * 3 + [200.0] * 4 + [20.0] * 2 + [200.0] * 4 + [20.0] * 2,
    max_episode_steps=1000, reward_threshold=None, height_body=0, penalty_dof=-1, height_lo=1.05, height_hi=2.0,
    angle_max=0.54, alive_bonus=1.0, ctrl_cost=1e-2, limit_penalty=0.0, aux_body_names=["h_torso_aux"], aux_ints=[18, 12],
    aux_real=[0.1, 0.45], all_bodies_collide=True, self_collision=True, spd_kp=_SPD_KP)

# DartDog-v1 -- reference gym/envs/dart/dog.py:9-65: quadruped on a FREE root joint (22 dofs), 16 actions x 200 on dofs 6..,
# reward 0.6 dx/dt + 1 - 1e-3 sum a^2 (:35-37), done on height outside (0.7, 1.8) or |z| >= 0.4 (:40-41), obs q[1:], clip(dq)
# in DART's FreeJoint coordinates (rotation vector first); gym/envs/__init__.py:247-251
DOG = TaskSpec(
    env_id="DartDog-v1", model="dog", task=TASK_DOG, frame_skip=4, act_dim=16, obs_dim=43, act_dof0=6, act_scale=[200.0] * 16,
    max_episode_steps=1000, reward_threshold=None, height_body=0, penalty_dof=-1, height_lo=0.7, height_hi=1.8,
    angle_max=np.inf, alive_bonus=1.0, ctrl_cost=1e-3, limit_penalty=0.0, aux_body_names=["main_body"], aux_real=[0.6, 0.4],
    all_bodies_collide=True)

TASKS = {t.env_id: t for t in (HOPPER, WALKER2D, WALKER3D, HUMANWALKER, CARTPOLE, HALFCHEETAH, CARTPOLE_SWINGUP,
                               DOUBLE_PENDULUM, SNAKE, REACHER2D, REACHER3D, WALKER3D_SPD, DOG)}
# reset_model() of these tasks draws more than the two uniform noise vectors.  The device MT19937 bank draws all of it itself
# (csrc/mt19937_kernels.hpp): the swing-up sign, the reach targets and -- since round 4 -- the double pendulum's Gaussian velocities
# (numpy's legacy polar method with a correctly rounded log, csrc/cr_log.hpp: the host stream bit for bit except for one draw in ~1 200,
# where glibc's own log is not correctly rounded, by one ulp).  noise="mt19937-host" still draws everything with numpy on the host.
HOST_RESET_TASKS = ()
MT_ONLY_TASKS = (TASK_CARTPOLE_SWINGUP, TASK_DOUBLE_PENDULUM, TASK_REACHER2D, TASK_REACHER3D)   # no Philox variant
# tasks with per-env state beyond (q, dq) that reset_model draws (the reach target): dart_set_task_state
TASK_STATE_TASKS = (TASK_REACHER2D, TASK_REACHER3D)

_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def load_model(name: str) -> ModelCard:
    """Load a compiled model card shipped with the package (``models/<name>.json``)."""
    with open(os.path.join(_MODELS_DIR, name + ".json")) as f:
        return ModelCard.from_json(f.read())


def build_card(model: ModelCard, task: Optional[TaskSpec] = None) -> DartModelCard:
    """Flatten a :class:`ModelCard` (+ optional task constants) into the C struct."""
    c = DartModelCard()
    c.version = CARD_VERSION
    c.struct_bytes = C.sizeof(DartModelCard)
    c.name = model.name.encode()[:31]
    c.dt = model.dt
    for i in range(3):
        c.gravity[i] = model.gravity[i]
    c.ground_y = model.ground_y
    c.friction, c.erp, c.max_erv, c.cfm, c.limit_erp = (model.friction, model.erp, model.max_erv,
                                                       model.cfm, model.limit_erp)
    c.nbodies, c.ndofs = model.nbodies, model.ndofs
    for i, b in enumerate(model.bodies):
        c.parent[i], c.jtype[i], c.dof_offset[i], c.ndof[i] = b.parent, b.jtype, b.dof_offset, b.ndof
        c.mass[i] = b.mass
        for k in range(3):
            c.com[i][k] = b.com[k]
        for k, v in enumerate(np.asarray(b.inertia).reshape(-1)):
            c.inertia[i][k] = v
        for k, v in enumerate(np.asarray(b.T_pj).reshape(-1)):
            c.T_pj[i][k] = v
        for k, v in enumerate(np.asarray(b.T_cj).reshape(-1)):
            c.T_cj[i][k] = v
        for k, v in enumerate(np.asarray(b.axes).reshape(-1)):
            c.axes[i][k] = v
    for i in range(model.ndofs):
        c.lower[i], c.upper[i] = model.lower[i], model.upper[i]
        c.limited[i] = int(model.limited[i])
        c.damping[i], c.stiffness[i], c.rest[i] = model.damping[i], model.stiffness[i], model.rest[i]
        c.init_pos[i], c.init_vel[i] = model.init_pos[i], model.init_vel[i]
        c.joint_friction[i] = 0.0 if model.joint_friction is None else float(model.joint_friction[i])
    c.contact_cfm = model.contact_cfm
    c.impulse_inertia = int(model.impulse_inertia)
    c.nshapes = len(model.shapes)
    for i, s in enumerate(model.shapes):
        c.shape_body[i], c.shape_type[i], c.shape_collidable[i] = s.body, s.kind, int(s.collidable)
        for k, v in enumerate(np.asarray(s.pose).reshape(-1)):
            c.shape_pose[i][k] = v
        for k in range(3):
            c.shape_size[i][k] = s.size[k]
    if task is None:
        c.task = TASK_NONE
        c.frame_skip = 1
        c.act_dim = model.ndofs
        c.obs_dim = 2 * model.ndofs
        c.act_dof0 = 0
        c.max_episode_steps = 0
        c.height_body, c.penalty_dof = 0, -1
        for k in range(model.ndofs):
            c.act_scale[k], c.act_low[k], c.act_high[k] = 1.0, -np.inf, np.inf
        c.state_abs_max, c.obs_vel_clip = np.inf, np.inf
    else:
        c.task, c.frame_skip, c.act_dim, c.obs_dim = task.task, task.frame_skip, task.act_dim, task.obs_dim
        c.act_dof0, c.max_episode_steps = task.act_dof0, task.max_episode_steps
        c.height_body, c.penalty_dof = task.height_body, task.penalty_dof
        for k in range(task.act_dim):
            c.act_scale[k] = task.act_scale[k]
            c.act_low[k], c.act_high[k] = (task.act_low, task.act_high) if task.clamp_actions else (-np.inf, np.inf)
        c.alive_bonus, c.ctrl_cost = task.alive_bonus, task.ctrl_cost
        c.limit_penalty, c.penalty_margin = task.limit_penalty, task.penalty_margin
        c.height_lo, c.height_hi, c.angle_max = task.height_lo, task.height_hi, task.angle_max
        c.state_abs_max, c.obs_vel_clip, c.reset_noise = task.state_abs_max, task.obs_vel_clip, task.reset_noise
        c.reset_noise_vel = task.reset_noise_vel
        if task.contact_cfm is not None:
            c.contact_cfm = task.contact_cfm
        c.self_collision = int(task.self_collision)
        for k, v in enumerate(task.spd_kp):
            c.spd_kp[k], c.spd_kd[k] = v, v / 10.0
        names = [b.name for b in model.bodies]
        for k, nm in enumerate(task.aux_body_names):
            c.aux_body[k] = names.index(nm)
        for k, v in enumerate(task.aux_ints):
            c.aux_body[len(task.aux_body_names) + k] = v
        for k, v in enumerate(task.aux_real):
            c.aux_real[k] = v
        for k, v in enumerate(task.aux_real2):
            c.aux_real2[k] = v
    return c


def card_for(env_id: str, all_bodies_collide: Optional[bool] = None, generic_kernel: bool = False) -> DartModelCard:
    """The card the batched env for ``env_id`` runs on.

    ``all_bodies_collide``: None = the task's default, which is DART's behaviour for every shipped env: each collision shape
    of the robot is tested against the ground (the reference never restricts it: hopper.py:14-18 / walker2d.py:12-18 only pick
    the detector).  False restricts the contacts to the task's ``contact_bodies`` (the feet; BASELINE.json config[1]'s
    "no contacts beyond foot-ground"), True forces every shape."""
    task = TASKS[env_id]
    model = load_model(task.model)
    every = task.all_bodies_collide if all_bodies_collide is None else bool(all_bodies_collide)
    for s in model.shapes:
        s.collidable = every or model.bodies[s.body].name in task.contact_bodies
    c = build_card(model, task)
    c.generic_kernel = int(generic_kernel)
    return c
