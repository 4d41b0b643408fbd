#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (gym fork at /root/reference) in THIS container.

The reference's task logic (DartHopperEnv / DartWalker2dEnv .step/_get_obs/reset_model, DartEnv, TimeLimit,
SyncVectorEnv, seeding.np_random, Box.sample) is imported unmodified.  Its physics backend pydart2/DART is not
installed anywhere, so `pydart2` (and the OpenGL/GLUT imports of static_window.py) are replaced in sys.modules by the
small stub below whose World.step() is driven by this repo's fp64 oracle.  The fixtures therefore pin, bit-for-bit in
fp64, everything on the hot path EXCEPT DART's arithmetic (which nothing in the reference pins either):
seeding -> reset noise, action stream, clamp/scale, reward, done, observation, TimeLimit, auto-reset order.

Run:  python tests/golden/make_golden.py      (needs /root/reference; the GPU box never runs this)
Only data (inputs + expected outputs) is written; no reference source text is stored.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("DART_REFERENCE", "/root/reference")

from dart_env_amd.model_card import build_card  # noqa: E402
from dart_env_amd.skel import parse_skel  # noqa: E402
from tests.oracle_lib import OracleWorld  # noqa: E402


# ------------------------------------------------------------------ stub pydart2 driven by the oracle
class Anything:
    """Permissive dummy for viewer / GL objects."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return Anything()

    def __call__(self, *a, **k):
        return Anything()

    def __getitem__(self, i):
        return 0.0

    def __setitem__(self, i, v):
        pass

    def __mul__(self, o):
        return 0.0

    __rmul__ = __mul__

    def __neg__(self):
        return 0.0

    def __sub__(self, o):
        return 0

    def __or__(self, o):
        return 0

    __ror__ = __or__

    def __iter__(self):
        return iter(())

    def __add__(self, o):
        return 0

    def __len__(self):
        return 1   # walker3d.py:28 loops over range(1, len(ground.bodynodes)): the shipped ground has ONE body


class SkelVector(np.ndarray):
    """pydart2 returns an ndarray subclass whose tuple keys are fancy indices (reference hopper.py:41 `q[0,2]`)."""

    def __new__(cls, a):
        return np.asarray(a, dtype=np.float64).view(cls)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            key = list(key)
        return np.asarray(self).__getitem__(key)


class StubDof:
    pass


class StubJoint:
    def __init__(self, limited):
        self._limited = limited
        self.dofs = [StubDof() for _ in limited]
        self.enforced = False

    def has_position_limit(self, d):
        return bool(self._limited[d])

    def set_position_limit_enforced(self, flag):
        self.enforced = flag


class StubBody:
    def __init__(self, world, index):
        self.world, self.index = world, index

    def com(self):
        return self.world.oracle.body_com(self.index)

    C = property(lambda self: self.com())

    def com_spatial_velocity(self):
        return self.world.oracle.body_com_spatial_velocity(self.index)

    def add_ext_force(self, f):
        self.world.oracle.add_body_force(self.index, np.asarray(f, dtype=np.float64))

    def set_collidable(self, flag):
        pass   # reacher2d.py:13-14: the card for that env marks no shape collidable

    def set_friction_coeff(self, mu):
        pass   # snake_7link.py:29-31: the snake never touches the floor (no vertical dof, 1 mm clearance)

    def local_com(self):
        return np.array(self.world.model.bodies[self.index].com, dtype=np.float64)

    def to_world(self, p=(0.0, 0.0, 0.0)):
        T = self.world.oracle.body_pose(self.index)
        return T[:3, :3] @ np.asarray(p, dtype=np.float64) + T[:3, 3]


class StubSkeleton:
    def __init__(self, world, model):
        self.world = world
        self.ndofs = model.ndofs
        self.q_lower = np.array(model.lower)
        self.q_upper = np.array(model.upper)
        self.joints = [StubJoint([model.limited[b.dof_offset + k] for k in range(b.ndof)]) for b in model.bodies]
        self.bodynodes = [StubBody(world, i) for i in range(model.nbodies)]
        self._by_name = {b.name: self.bodynodes[i] for i, b in enumerate(model.bodies)}
        self.name_to_body = self._by_name

    def bodynode(self, name):
        return self._by_name[name]

    q = property(lambda self: SkelVector(self.world.oracle.get_state()[0]))
    dq = property(lambda self: SkelVector(self.world.oracle.get_state()[1]))

    M = property(lambda self: self.world.oracle.mass_matrix())
    c = property(lambda self: self.world.oracle.bias())

    def constraint_forces(self):
        return self.world.oracle.constraint_forces()

    def set_positions(self, q):
        self.world.oracle.set_state(np.asarray(q, dtype=np.float64), self.world.oracle.get_state()[1])

    def set_velocities(self, dq):
        self.world.oracle.set_state(self.world.oracle.get_state()[0], np.asarray(dq, dtype=np.float64))

    def set_forces(self, tau):
        self.world.oracle.set_forces(np.asarray(tau, dtype=np.float64))

    def com(self):
        return np.zeros(3)

    def set_self_collision_check(self, flag):
        assert bool(flag) == bool(self.world.oracle.card.self_collision)   # the card was built for this env's choice


class StubContact:
    def __init__(self, ground, body, force):
        self.skel_id1, self.skel_id2 = 0, 1
        self.bodynode1, self.bodynode2 = ground, body
        self.force = force


class StubCollisionResult:
    def __init__(self):
        self.contacts = []


class StubWorld:
    n_steps = 0

    spd = False   # walker3d_waist.skel serves two env ids; the generator flips this before making the SPD one

    def __init__(self, dt, skel_path=None):
        name = os.path.basename(skel_path)
        contact = {"hopper_capsule.skel": None, "walker2d.skel": None,   # every capsule collides, as in DART (the default cards)
                   "kima_human_edited.skel": None, "walker3d_waist.skel": None,
                   "cartpole.skel": None, "half_cheetah.skel": None, "cartpole_swingup.skel": None,
                   "inverted_double_pendulum.skel": None, "snake_7link.skel": None, "reacher2d.skel": [],
                   "reacher.skel": [], "dog.skel": None}[name]   # None: every collision shape
        model = parse_skel(skel_path, dt=dt, collidable_bodies=contact)
        self.model = model
        self.dt = dt
        card = build_card(model, None)
        from dart_env_amd.model_card import TASKS
        spec = {"hopper_capsule.skel": "DartHopper-v1", "walker2d.skel": "DartWalker2d-v1",
                "kima_human_edited.skel": "DartHumanWalker-v1", "walker3d_waist.skel": "DartWalker3d-v1",
                "cartpole.skel": "DartCartPole-v1", "half_cheetah.skel": "DartHalfCheetah-v1",
                "cartpole_swingup.skel": "DartCartPoleSwingUp-v1",
                "inverted_double_pendulum.skel": "DartDoubleInvertedPendulumEnv-v1",
                "snake_7link.skel": "DartSnake7Link-v1", "reacher2d.skel": "DartReacher-v1",
                "reacher.skel": "DartReacher3d-v1", "dog.skel": "DartDog-v1"}[name]
        if StubWorld.spd and name == "walker3d_waist.skel":
            spec = "DartWalker3dSPD-v1"
        if TASKS[spec].contact_cfm is not None:   # same contact regularisation as the shipped task card
            card.contact_cfm = TASKS[spec].contact_cfm
        card.self_collision = int(TASKS[spec].self_collision)   # what the env's set_self_collision_check() call will ask for
        self.oracle = OracleWorld(card)
        # ground (and target) skeletons are permissive dummies; the robot is the last one (dart_env.py:62)
        n_other = {"reacher2d.skel": 2}.get(name, 1)
        self.skeletons = [Anything() for _ in range(n_other)] + [StubSkeleton(self, model)]
        self.collision_result = StubCollisionResult()

    def step(self):
        StubWorld.n_steps += 1
        self.oracle.step()
        # pydart2 refreshes collision_result after every world step
        robot = self.skeletons[-1]
        self.collision_result.contacts = [
            StubContact(self.skeletons[0], robot.bodynodes[int(c[0])], np.array([c[6], c[5], c[7]]) / self.dt)
            for c in self.oracle.last_contacts()]

    def reset(self):
        self.oracle.reset()

    def set_collision_detector(self, k):
        self.detector = k


def install_stubs():
    pyd = types.ModuleType("pydart2")
    pyd.init = lambda *a, **k: None
    pyd.World = StubWorld
    sys.modules["pydart2"] = pyd
    for name in ["pydart2.gui", "pydart2.gui.trackball", "pydart2.gui.opengl", "pydart2.gui.opengl.scene",
                 "pydart2.gui.glut", "pydart2.gui.glut.window", "OpenGL", "OpenGL.GL", "OpenGL.GLU", "OpenGL.GLUT"]:
        m = types.ModuleType(name)
        m.__getattr__ = lambda attr: Anything()
        m.__all__ = []
        sys.modules[name] = m
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            setattr(sys.modules[parent], leaf, m)
    sys.modules["pydart2.gui.trackball"].Trackball = Anything
    sys.modules["pydart2.gui.opengl.scene"].OpenGLScene = Anything
    sys.modules["pydart2.gui.glut.window"].GLUTWindow = type("GLUTWindow", (), {
        "__init__": lambda self, *a, **k: setattr(self, "scene", Anything()),
        "__getattr__": lambda self, name: Anything(),
        "run": lambda self, *a, **k: None, "close": lambda self: None})
    sys.modules["pydart2.gui.glut.window"].__all__ = ["GLUTWindow"]
    sys.path.insert(0, REF)


# ------------------------------------------------------------------ fixture generation
def rollout_single(gym, env_id, seed, steps, act_scale=1.0, limit=None):
    if limit is None:
        env = gym.make(env_id)
        if "Human" in env_id:
            assert env.unwrapped.robot_skeleton.ndofs == 29
    else:  # the reference's TimeLimit wrapper with a short horizon, so truncation is actually exercised
        from gym.wrappers import TimeLimit
        from gym.envs.dart import DartHopperEnv, DartWalker2dEnv
        env = TimeLimit({"DartHopper-v1": DartHopperEnv, "DartWalker2d-v1": DartWalker2dEnv}[env_id](),
                        max_episode_steps=limit)
    env.seed(seed)
    env.action_space.seed(seed + 1000)
    rec = dict(obs0=None, actions=[], obs=[], reward=[], done=[], truncated=[], q=[], dq=[], reset_obs=[])
    rec["obs0"] = env.reset()
    for t in range(steps):
        a32 = (env.action_space.sample() * act_scale).astype(np.float32)
        a = a32.astype(np.float64)  # float64 view of the float32 sample: numpy-version independent
        ob, r, d, info = env.step(a)
        rec["actions"].append(a32); rec["obs"].append(ob); rec["reward"].append(r); rec["done"].append(d)
        rec["truncated"].append(bool(info.get("TimeLimit.truncated", False)))
        if "broke_sim" in info:
            rec.setdefault("broke_sim", []).append(bool(info["broke_sim"]))
        sv = env.unwrapped.state_vector()
        n = len(sv) // 2
        rec["q"].append(sv[:n]); rec["dq"].append(sv[n:])
        if d:
            rec["reset_obs"].append(env.reset())
        else:
            rec["reset_obs"].append(np.full_like(ob, np.nan))
    return {k: np.asarray(v) for k, v in rec.items()}


def rollout_vector(gym, env_id, n_envs, seed, steps, act_scale=1.0):
    from gym.vector import SyncVectorEnv
    venv = SyncVectorEnv([lambda: gym.make(env_id) for _ in range(n_envs)])
    venv.seed(seed)
    venv.action_space.seed(seed + 77)
    rec = dict(obs0=venv.reset(), actions=[], obs=[], reward=[], done=[], truncated=[])
    for t in range(steps):
        a = (np.stack(venv.action_space.sample()) * act_scale).astype(np.float32)
        ob, r, d, infos = venv.step(list(a.astype(np.float64)))
        rec["actions"].append(a); rec["obs"].append(ob); rec["reward"].append(r); rec["done"].append(d)
        rec["truncated"].append([bool(i.get("TimeLimit.truncated", False)) for i in infos])
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["obs_dtype"] = np.array(str(out["obs"].dtype)); out["reward_dtype"] = np.array(str(out["reward"].dtype))
    out["done_dtype"] = np.array(str(out["done"].dtype))
    return out


def main():
    install_stubs()
    import gym
    from gym.utils import seeding
    from gym import spaces
    out = os.path.dirname(os.path.abspath(__file__))

    # (1) seeding -> reset-noise draws, (2) Box action stream
    seed_fix = {}
    for s in range(8):
        for n in (6, 9, 21, 29):
            rng, _ = seeding.np_random(s)
            seed_fix["noise_s%d_n%d" % (s, n)] = np.stack([rng.uniform(-.005, .005, n), rng.uniform(-.005, .005, n)])
        seed_fix["hash_%d" % s] = np.array(seeding.hash_seed(s), dtype=np.uint64)
    for k in range(4):
        b = spaces.Box(-np.ones(3, dtype=np.float32), np.ones(3, dtype=np.float32))
        b.seed(k)
        seed_fix["box3_seed%d" % k] = np.stack([b.sample() for _ in range(1000)])
    np.savez_compressed(os.path.join(out, "seeding.npz"), **seed_fix)

    # (3) single-env task logic incl. episode ends and (4) TimeLimit truncation at 1000 (tiny actions keep it alive)
    for env_id, tag in (("DartHopper-v1", "hopper"), ("DartWalker2d-v1", "walker2d")):
        np.savez_compressed(os.path.join(out, "%s_single_seed0.npz" % tag), **rollout_single(gym, env_id, 0, 300))
        np.savez_compressed(os.path.join(out, "%s_single_seed5_small.npz" % tag),
                            **rollout_single(gym, env_id, 5, 1100, act_scale=0.02))
        np.savez_compressed(os.path.join(out, "%s_single_seed2_limit20.npz" % tag),
                            **rollout_single(gym, env_id, 2, 100, act_scale=0.02, limit=20))
        # (5) SyncVectorEnv semantics: seed fan-out s+i, auto-reset with post-reset obs, dtypes
        np.savez_compressed(os.path.join(out, "%s_vector4_seed3.npz" % tag), **rollout_vector(gym, env_id, 4, 3, 120))
    # (6) DartHumanWalker-v1 (29 dof, box feet, springs, contact flags in the observation), 300-step TimeLimit
    np.savez_compressed(os.path.join(out, "humanwalker_single_seed0.npz"), **rollout_single(gym, "DartHumanWalker-v1", 0, 1000))
    np.savez_compressed(os.path.join(out, "humanwalker_single_seed4_small.npz"),
                        **rollout_single(gym, "DartHumanWalker-v1", 4, 150, act_scale=0.05))
    # (7) DartWalker3d-v1 (21 dof, box links; the stub world ignores set_self_collision_check, like the HIP kernel)
    np.savez_compressed(os.path.join(out, "walker3d_single_seed0.npz"), **rollout_single(gym, "DartWalker3d-v1", 0, 1000))
    np.savez_compressed(os.path.join(out, "walker3d_single_seed6_small.npz"),
                        **rollout_single(gym, "DartWalker3d-v1", 6, 400, act_scale=0.05))
    np.savez_compressed(os.path.join(out, "walker3d_vector4_seed3.npz"), **rollout_vector(gym, "DartWalker3d-v1", 4, 3, 120))
    # (8) DartCartPole-v1 (no clamp, obs [q, dq], dt 0.02) and DartHalfCheetah-v1 (all capsules collide, dt 0.01)
    for env_id, tag in (("DartCartPole-v1", "cartpole"), ("DartHalfCheetah-v1", "halfcheetah")):
        np.savez_compressed(os.path.join(out, "%s_single_seed0.npz" % tag), **rollout_single(gym, env_id, 0, 300))
        np.savez_compressed(os.path.join(out, "%s_single_seed1_big.npz" % tag),
                            **rollout_single(gym, env_id, 1, 200, act_scale=1.5))   # beyond +-1: clamp / no clamp
        np.savez_compressed(os.path.join(out, "%s_vector4_seed3.npz" % tag), **rollout_vector(gym, env_id, 4, 3, 120))
    # (9) cart-pole swing-up (third reset draw: +-pi) and double inverted pendulum (Gaussian velocity noise, sin/cos obs)
    for env_id, tag in (("DartCartPoleSwingUp-v1", "swingup"), ("DartDoubleInvertedPendulumEnv-v1", "doublependulum")):
        np.savez_compressed(os.path.join(out, "%s_single_seed0.npz" % tag), **rollout_single(gym, env_id, 0, 300, act_scale=1.5))
        np.savez_compressed(os.path.join(out, "%s_vector4_seed3.npz" % tag), **rollout_vector(gym, env_id, 4, 3, 150, act_scale=1.5))
    # (10) DartSnake7Link-v1: the reference's own fluid-force loop (com_spatial_velocity / add_ext_force per body and substep)
    np.savez_compressed(os.path.join(out, "snake_single_seed0.npz"), **rollout_single(gym, "DartSnake7Link-v1", 0, 300))
    np.savez_compressed(os.path.join(out, "snake_vector4_seed3.npz"), **rollout_vector(gym, "DartSnake7Link-v1", 4, 3, 120))
    # (12) DartReacher-v1 (2-D): Coulomb joint friction rows, target in the x-z disc, never done (TimeLimit 50)
    np.savez_compressed(os.path.join(out, "reacher2d_single_seed0.npz"), **rollout_single(gym, "DartReacher-v1", 0, 260))
    np.savez_compressed(os.path.join(out, "reacher2d_vector4_seed3.npz"), **rollout_vector(gym, "DartReacher-v1", 4, 3, 120))
    # (11) DartReacher3d-v1 (target resampled by rejection, reward / done from the pre-step fingertip distance)
    np.savez_compressed(os.path.join(out, "reacher3d_single_seed0.npz"), **rollout_single(gym, "DartReacher3d-v1", 0, 700))
    np.savez_compressed(os.path.join(out, "reacher3d_vector4_seed3.npz"), **rollout_vector(gym, "DartReacher3d-v1", 4, 3, 120))
    # (13) DartWalker3dSPD-v1: the reference's _spd (numpy inverse of M + Kd dt, skel.c, constraint_forces()) per substep
    StubWorld.spd = True
    np.savez_compressed(os.path.join(out, "walker3dspd_single_seed0.npz"), **rollout_single(gym, "DartWalker3dSPD-v1", 0, 400))
    StubWorld.spd = False
    # (14) DartDog-v1: free root joint; the reference's task code reads q / dq in DART's FreeJoint coordinates
    np.savez_compressed(os.path.join(out, "dog_single_seed0.npz"), **rollout_single(gym, "DartDog-v1", 0, 300))
    np.savez_compressed(os.path.join(out, "dog_vector4_seed3.npz"), **rollout_vector(gym, "DartDog-v1", 4, 3, 100))
    print("world.step() calls issued by the reference code:", StubWorld.n_steps)


if __name__ == "__main__":
    main()
