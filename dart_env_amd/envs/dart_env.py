"""Batched counterpart of the reference's ``DartEnv`` base class (reference gym/envs/dart/dart_env.py:25-215).

The reference builds ONE pydart2 world per env object and steps it from Python.  Here one object owns
``num_envs`` worlds that live in HBM behind the C ABI (``include/dart_stepper.h``); everything the reference
computes per step in Python (clamp/scale, reward, done, observation, TimeLimit) happens inside the fused HIP kernel.
What stays on the host is what the reference also keeps there: seeding and reset noise from
``np_random`` (MT19937, reference hopper.py:78-79), spaces, and the auto-reset policy of the vector wrapper.

Reset noise modes
  * ``noise="mt19937"`` (default): bit-exact with the reference -- env i owns the MT19937 stream of
    ``seeding.np_random(seed_i)`` and a reset draws ``uniform(-r, r, ndofs)`` for qpos then for qvel (hopper.py:78-79).
    The generators live in HBM (``dart_seed_mt19937``), so this costs no host loop at any batch size.
  * ``noise="mt19937-host"``: the same streams drawn by numpy on the host and uploaded (cross-check / CPU tests).
  * ``noise="philox"``: counter-based noise generated inside the step kernel (bench.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np

from .. import seeding, spaces
from ..model_card import (DartModelCard, HOST_RESET_TASKS, MT_ONLY_TASKS, TASK_CARTPOLE_SWINGUP, TASK_DOUBLE_PENDULUM, TASK_REACHER2D,
                          TASK_REACHER3D, TASK_STATE_TASKS, TASKS, card_for)
from .. import stepper as _st


class BodyNodeView:
    """One ``bodynode`` of the robot skeleton for all envs: the pose getters the reference's task code calls
    (hopper.py:42,72 ``com()``; human_walker.py:81-92 ``com()``, ``to_world()``)."""

    def __init__(self, skel, index, name):
        self._skel, self.index, self.name = skel, index, name

    def _poses(self):
        R, p, c = self._skel._env._stepper.body_poses()
        return self._skel._u(R[:, self.index]), self._skel._u(p[:, self.index]), self._skel._u(c[:, self.index])

    def com(self):
        """World COM, (num_envs, 3) (pydart2 BodyNode.com() / .C)."""
        return self._poses()[2]

    C = property(com)

    @property
    def T(self):
        """World transform, (num_envs, 4, 4) (pydart2 BodyNode.T / world_transform())."""
        R, p, _ = self._poses()
        T = np.zeros(R.shape[:-2] + (4, 4))
        T[..., :3, :3] = R; T[..., :3, 3] = p; T[..., 3, 3] = 1.0
        return T

    world_transform = lambda self: self.T

    def to_world(self, x=(0.0, 0.0, 0.0)):
        """Body-frame point -> world, (num_envs, 3) (pydart2 BodyNode.to_world, human_walker.py:86-92)."""
        R, p, _ = self._poses()
        return np.einsum("...ij,j->...i", R, np.asarray(x, dtype=np.float64)) + p


class SkeletonView:
    """The ``robot_skeleton`` attributes the reference's task code reads (hopper.py:39-49,69-72; human_walker.py:78-92;
    walker3d_spd.py:44-51), batched: one row per env (un-batched for the single-env facades)."""

    def __init__(self, env):
        self._env = env
        self._names = None

    def _u(self, a):
        return a[0] if getattr(self._env, "_unbatched", False) else a

    @property
    def ndofs(self):
        return self._env.ndofs

    @property
    def q(self):
        return self._u(self._env._stepper.get_state()[0])

    @property
    def dq(self):
        return self._u(self._env._stepper.get_state()[1])

    @property
    def M(self):
        """Mass matrices (num_envs, ndofs, ndofs) -- pydart2 skel.M (reference walker3d_spd.py:44)."""
        return self._u(self._env._stepper.dynamics(True, False)[0])

    @property
    def c(self):
        """Coriolis + gravity forces (num_envs, ndofs) -- pydart2 skel.c (reference walker3d_spd.py:49)."""
        return self._u(self._env._stepper.dynamics(False, True)[1])

    def constraint_forces(self):
        """(num_envs, ndofs) -- pydart2 skel.constraint_forces() (reference walker3d_spd.py:51); enable_contact_report() first."""
        return self._u(self._env._stepper.constraint_forces())

    @property
    def bodynodes(self):
        """BodyNodeView per skeleton body, in the .skel file's order (``robot_skeleton.bodynodes[i]``)."""
        if self._names is None:
            self._names = self._env._body_names()
        return [BodyNodeView(self, i, nm) for i, nm in enumerate(self._names)]

    def bodynode(self, name):
        for b in self.bodynodes:
            if b.name == name:
                return b
        raise KeyError(name)

    @property
    def q_lower(self):
        c = self._env.card
        return np.array([c.lower[i] for i in range(c.ndofs)])

    @property
    def q_upper(self):
        c = self._env.card
        return np.array([c.upper[i] for i in range(c.ndofs)])


class BatchedDartEnv:
    """num_envs Dart worlds stepped in lock-step on one GPU.  No auto-reset here (see DartVectorEnv)."""

    metadata = {"render.modes": []}

    def __init__(self, env_id: str, num_envs: int = 1, device: int = 0, precision: int = 64, noise: str = "mt19937",
                 max_episode_steps: Optional[int] = None, card: Optional[DartModelCard] = None,
                 stepper_factory: Optional[Callable] = None, generic_kernel: bool = False):
        if noise not in ("mt19937", "mt19937-host", "philox"):
            raise ValueError("noise must be 'mt19937', 'mt19937-host' or 'philox'")
        self.env_id = env_id
        self.task = TASKS[env_id]
        self.card = card if card is not None else card_for(env_id, generic_kernel=generic_kernel)
        if max_episode_steps is not None:
            self.card.max_episode_steps = int(max_episode_steps)
        self.num_envs = int(num_envs)
        self.ndofs, self.obs_dim, self.act_dim = self.card.ndofs, self.card.obs_dim, self.card.act_dim
        self.frame_skip = self.card.frame_skip
        self.noise = noise
        factory = stepper_factory or _st.HipStepper  # no CPU fallback: HipStepper raises without lib/GPU
        self._stepper = factory(self.card, self.num_envs, device, precision)
        if noise == "mt19937" and not hasattr(self._stepper, "seed_mt19937"):
            noise = self.noise = "mt19937-host"     # injected test stepper without a device bank
        if self.task.task in MT_ONLY_TASKS and noise == "philox":   # reset_model draws beyond two uniform vectors
            raise ValueError("%s resets from the MT19937 streams only (noise='mt19937')" % env_id)
        if self.task.task in HOST_RESET_TASKS:      # Gaussian draws: numpy on the host
            noise = self.noise = "mt19937-host"
        self.device_noise = noise in ("mt19937", "philox")
        # spaces exactly as DartEnv.__init__ builds them (dart_env.py:85-86, 97-100)
        # control_bounds of the reference env (the card carries +-inf instead when the env does not clamp)
        hi = np.full(self.act_dim, self.task.act_high, dtype=np.float64)
        lo = np.full(self.act_dim, self.task.act_low, dtype=np.float64)
        self.action_space = spaces.Box(lo, hi)
        inf = np.inf * np.ones(self.obs_dim)
        self.observation_space = spaces.Box(-inf, inf)
        self.robot_skeleton = SkeletonView(self)
        self._rngs = [None] * self.num_envs
        self._seeds = [None] * self.num_envs
        self._task_state = np.zeros((self.num_envs, 4))   # per-env state reset_model draws besides (q, dq): reach targets
        self.seed(None)

    # ---- reference API -------------------------------------------------------------------------------------
    @property
    def dt(self):
        return self.card.dt * self.frame_skip  # dart_env.py:154-156

    def seed(self, seeds=None):
        """int s -> env i seeded s+i (sync_vector_env.py:50-58); list -> per env; None -> OS entropy."""
        if seeds is None:
            seeds = [None] * self.num_envs
        elif isinstance(seeds, (int, np.integer)):
            seeds = [int(seeds) + i for i in range(self.num_envs)]
        assert len(seeds) == self.num_envs
        for sd in seeds:
            if sd is not None and not (isinstance(sd, (int, np.integer)) and 0 <= sd):
                raise seeding.SeedError("Seed must be a non-negative integer or omitted, not %r" % (sd,))
        # the seeds actually used (None -> OS entropy, resolved here once), returned like DartEnv.seed's [seed] (dart_env.py:117-119)
        used = [seeding.create_seed(None if sd is None else int(sd)) for sd in seeds]
        self._pending_seeds = list(used)
        self._seeds = list(used)
        self._rngs = [None] * self.num_envs   # RandomStates are built lazily, on an env's first reset
        if self.noise == "philox":
            self._stepper.configure(_st.CFG_SEED, float(used[0] % (1 << 53)))
        elif self.noise == "mt19937":
            # the words numpy's RandomState.seed(list) receives in seeding.np_random (gym/utils/seeding.py:17-18)
            keys, klen = seeding.mt_keys(used)
            self._stepper.seed_mt19937(keys, klen)
        return list(used)

    def _rng(self, i):
        if self._rngs[i] is None:
            self._rngs[i], self._seeds[i] = seeding.np_random(self._pending_seeds[i])
        return self._rngs[i]

    def _draw_noise(self, mask):
        r, rv = self.card.reset_noise, self.card.reset_noise_vel
        qn = np.zeros((self.num_envs, self.ndofs))
        vn = np.zeros((self.num_envs, self.ndofs))
        idx = range(self.num_envs) if mask is None else np.flatnonzero(mask)
        kind = self.task.task
        for i in idx:
            rng = self._rng(i)
            qn[i] = rng.uniform(low=-r, high=r, size=self.ndofs)   # qpos first (hopper.py:78)
            if kind == TASK_DOUBLE_PENDULUM:                       # inverted_double_pendulum.py:50-51
                vn[i] = rng.randn(self.ndofs) * rv
            else:
                vn[i] = rng.uniform(low=-rv, high=rv, size=self.ndofs)   # qvel second (hopper.py:79, human_walker.py:154)
            if kind == TASK_CARTPOLE_SWINGUP:                      # cartpole_swingup.py:41-44: the pole starts hanging
                qn[i, 1] += np.pi if rng.uniform(low=0, high=1, size=1) > 0.5 else -np.pi
            if kind == TASK_REACHER2D:                             # reacher2d.py:53-57: target in the disc of radius 0.2
                while True:
                    tgt = rng.uniform(low=-.2, high=.2, size=3)
                    tgt[1] = 0.0
                    if np.linalg.norm(tgt) < .2:
                        break
                tgt[1] = 0.01
                self._task_state[i, :3] = tgt
            if kind == TASK_REACHER3D:                             # reacher.py:50-52: target in the ball of radius 1.5
                while True:
                    tgt = rng.uniform(low=-1, high=1, size=3)
                    if np.linalg.norm(tgt) < 1.5:
                        break
                self._task_state[i, :3] = tgt
        return qn, vn

    def reset(self, mask=None):
        """reset_model() for the masked envs (all when None); returns the (num_envs, obs_dim) float32 observations."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if self.device_noise:
            return self._stepper.reset(m, None, None)
        qn, vn = self._draw_noise(m)
        if self.task.task in TASK_STATE_TASKS:      # the new targets must be on the device before the reset observation
            self._stepper.set_task_state(m, self._task_state)
        return self._stepper.reset(m, qn, vn)

    def step(self, actions):
        """-> obs (N,obs) f32, reward (N,) f64, done (N,) bool, truncated (N,) bool   (no auto-reset)"""
        return self._stepper.step(actions)

    def step_async(self, actions, staged=False):
        if staged:
            self._stepper.step_async(actions, staged=True)
        else:
            self._stepper.step_async(actions)

    def step_wait(self, copy=True):
        return self._stepper.step_wait() if copy else self._stepper.step_wait(copy=False)

    def set_ext_force(self, body, forces):
        """``bodynodes[body].add_ext_force(F_i)`` before every world step of env i from now on -- the perturbation branch
        of DartEnv.do_simulation (dart_env.py:159-172), with the host choosing the forces.  ``forces``: (num_envs, 3)
        world-frame vectors or None to stop.  Needs ``generic_kernel=True`` for the planar models."""
        self._stepper.set_ext_force(body, forces)

    def enable_contact_report(self, on=True):
        """Record ``world.collision_result.contacts`` of every env-step's last world step (walker2d.py:38-41,
        human_walker.py:97-106).  Planar models need ``generic_kernel=True``."""
        self._stepper.configure(_st.CFG_CONTACT_REPORT, 1 if on else 0)

    def contacts(self):
        """-> count (N,), bodies (N, K, 2) {bodynode1, bodynode2 as skeleton body indices; -1 = ground}, point (N, K, 3),
        force (N, K, 3) on the first body -- the fields the reference reads from pydart2 Contact objects."""
        return self._stepper.contacts()

    def set_state(self, qpos, qvel):
        qpos = np.asarray(qpos, dtype=np.float64).reshape(self.num_envs, self.ndofs)
        qvel = np.asarray(qvel, dtype=np.float64).reshape(self.num_envs, self.ndofs)
        self._stepper.set_state(qpos, qvel)

    def state_vector(self):
        q, dq = self._stepper.get_state()
        return np.concatenate([q, dq], axis=1)  # dart_env.py:211-215, one row per env

    def snapshot(self):
        """Exact checkpoint of the device side (state, TimeLimit / episode counters, MT19937 bank, task state) -- with the
        default device noise the following steps and auto-resets are bitwise those of the original run after restore()."""
        return self._stepper.snapshot()

    def restore(self, snap):
        self._stepper.restore(snap)

    def _body_names(self):
        from ..model_card import load_model
        return [b.name for b in load_model(self.task.model).bodies]

    def close(self):
        if self._stepper is not None:
            self._stepper.close()
            self._stepper = None


class DartEnv:
    """The reference's ``DartEnv`` base class for user-defined tasks, batched (reference gym/envs/dart/dart_env.py:25-215):
    same constructor arguments, ``do_simulation(tau, n_frames)``, ``set_state``, ``state_vector``, ``dt``,
    ``robot_skeleton``, ``seed`` / ``np_random``, ``reset()`` -> ``reset_model()`` -- a subclass written the way the reference's
    env classes are (hopper.py etc.) runs ``num_envs`` worlds at once with (num_envs, ...) arrays in place of vectors.

    ``model_paths``: a ``.skel`` file (or a list whose last entry is one; absolute, or relative to the working directory --
    the reference resolves relative names against its own assets directory, dart_env.py:38-44).  The robot is
    ``skeletons[-1]`` (dart_env.py:62), its finite joint limits are enforced (dart_env.py:64-67), the world runs at ``dt``
    (dart_env.py:55).  The world lives in HBM behind the C ABI as a physics-only card: action = generalized forces.
    Only ``obs_type='parameter'`` / ``action_type='continuous'`` (no rendering, SURVEY.md section 8 scope)."""

    _unbatched = False

    def __init__(self, model_paths, frame_skip, observation_size, action_bounds, dt=0.002, obs_type="parameter",
                 action_type="continuous", visualize=False, disableViewer=True, screen_width=80, screen_height=45,
                 num_envs=1, device=0, precision=64, collidable_bodies=None, stepper_factory=None, generic_kernel=False):
        """generic_kernel=True forces the tree kernel; by default the library first offers the model to the lane-per-env register
        kernels (a .skel whose tree matches one of their compiled topologies runs ~two orders of magnitude faster there:
        DART_Q_LANE_KERNEL tells which one serves it) and falls back to the tree kernel for every other shape."""
        import os
        from ..skel import parse_skel
        from ..model_card import build_card
        if obs_type != "parameter" or action_type != "continuous":
            raise NotImplementedError("image observations / discrete actions need the renderer (out of scope)")
        if isinstance(model_paths, str):
            model_paths = [model_paths]
        path = model_paths[-1]
        if not os.path.exists(path):
            raise IOError("File %s does not exist" % path)      # dart_env.py:43-44
        self.model = parse_skel(path, dt=dt, collidable_bodies=collidable_bodies)
        self.card = build_card(self.model, None)
        self.card.generic_kernel = int(bool(generic_kernel))
        self.card.frame_skip = int(frame_skip)
        self.num_envs, self.frame_skip = int(num_envs), int(frame_skip)
        self.ndofs = self.card.ndofs
        self._stepper = (stepper_factory or _st.HipStepper)(self.card, self.num_envs, device, precision)
        self.robot_skeleton = SkeletonView(self)
        self._obs_type, self.obs_dim, self.act_dim = obs_type, int(observation_size), len(action_bounds[0])
        self.action_space = spaces.Box(np.asarray(action_bounds[1], dtype=np.float64), np.asarray(action_bounds[0], dtype=np.float64))
        self.observation_space = spaces.Box(-np.inf * np.ones(self.obs_dim), np.inf * np.ones(self.obs_dim))   # dart_env.py:97-100
        self.metadata = {"render.modes": []}
        self._dev = None
        self.seed()

    def _body_names(self):
        return [b.name for b in self.model.bodies]

    def seed(self, seed=None):
        """int s -> env i seeded s + i (sync_vector_env.py:50-58); each env owns ``seeding.np_random(seed_i)`` (dart_env.py:117-119)."""
        seeds = [None] * self.num_envs if seed is None else ([int(seed) + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed))
        pairs = [seeding.np_random(sd) for sd in seeds]
        self.np_randoms = [p[0] for p in pairs]
        self.np_random = self.np_randoms[0]
        return [p[1] for p in pairs]

    def uniform(self, low, high, size):
        """(num_envs, size): ``self.np_random.uniform(low, high, size)`` of every env's own stream, in env order."""
        return np.stack([r.uniform(low=low, high=high, size=size) for r in self.np_randoms])

    @property
    def dt(self):
        return self.card.dt * self.frame_skip      # dart_env.py:154-156

    def do_simulation(self, tau, n_frames):
        """``n_frames`` world steps with the generalized forces ``tau`` (num_envs, ndofs) re-applied before each
        (dart_env.py:158-175).  ``n_frames`` must be a multiple of the constructor's ``frame_skip`` (one launch each)."""
        k, r = divmod(int(n_frames), self.frame_skip)
        if r or k < 1:
            raise ValueError("n_frames must be a positive multiple of frame_skip=%d" % self.frame_skip)
        if hasattr(tau, "data_ptr") and getattr(tau, "is_cuda", False):
            # torch tensor resident in HBM: launch on torch's current stream, no host round trip; returns the new state
            # [q, dq] as a (num_envs, 2 ndofs) float32 tensor (valid until the next call) for a task written in torch
            import torch
            t = tau.to(dtype=torch.float32).reshape(self.num_envs, self.ndofs).contiguous()
            if self._dev is None:
                self._dev = (torch.empty((self.num_envs, 2 * self.ndofs), dtype=torch.float32, device=t.device),
                             torch.empty(self.num_envs, dtype=torch.float32, device=t.device),
                             torch.empty(self.num_envs, dtype=torch.uint8, device=t.device),
                             torch.empty(self.num_envs, dtype=torch.uint8, device=t.device))
            st_, rw, dn, tr = self._dev
            cur = torch.cuda.current_stream(t.device)
            stream = cur.cuda_stream
            if stream == 0:
                # torch's legacy default stream has no handle to pass (NULL means "the stepper's own stream" in the ABI):
                # order the two streams by hand -- tau must be complete before, the state visible to torch after
                cur.synchronize()
            for _ in range(k):
                self._stepper.step_device(t.data_ptr(), st_.data_ptr(), rw.data_ptr(), dn.data_ptr(), tr.data_ptr(), stream)
            if stream == 0:
                self._stepper.sync()
            return st_
        tau = np.ascontiguousarray(np.asarray(tau, dtype=np.float32).reshape(self.num_envs, self.ndofs))
        for _ in range(k):
            self._stepper.step(tau)

    def set_state(self, qpos, qvel):
        qpos = np.asarray(qpos, dtype=np.float64); qvel = np.asarray(qvel, dtype=np.float64)
        assert qpos.shape == (self.num_envs, self.ndofs) and qvel.shape == (self.num_envs, self.ndofs)   # dart_env.py:146
        self._stepper.set_state(qpos, qvel)

    def set_state_vector(self, state):
        state = np.asarray(state, dtype=np.float64).reshape(self.num_envs, 2 * self.ndofs)
        self.set_state(state[:, :self.ndofs], state[:, self.ndofs:])     # dart_env.py:150-152

    def state_vector(self):
        q, dq = self._stepper.get_state()
        return np.concatenate([q, dq], axis=1)     # dart_env.py:211-215

    @property
    def init_qpos(self):
        return np.tile(np.array([self.card.init_pos[i] for i in range(self.ndofs)]), (self.num_envs, 1))

    @property
    def init_qvel(self):
        return np.tile(np.array([self.card.init_vel[i] for i in range(self.ndofs)]), (self.num_envs, 1))

    def enable_contact_report(self, on=True):
        self._stepper.configure(_st.CFG_CONTACT_REPORT, 1 if on else 0)

    def contacts(self):
        return self._stepper.contacts()

    def reset_model(self):
        """Reset the robot degrees of freedom (qpos and qvel); implement this in each subclass (dart_env.py:124-129)."""
        raise NotImplementedError

    def reset(self):
        self._stepper.reset(None, None, None, want_obs=False)      # world.reset(): q, dq back to the initial values
        return self.reset_model()                                   # dart_env.py:140-143

    def step(self, a):
        raise NotImplementedError

    def close(self):
        if self._stepper is not None:
            self._stepper.close()
            self._stepper = None
