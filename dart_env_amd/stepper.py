"""ctypes binding of ``libdart_stepper.so`` (C ABI: ``include/dart_stepper.h``).

This is the only way the Python env layer reaches the physics: there is no
Python or CPU implementation behind it.  If the shared library is missing, or
no HIP device is present, construction raises -- loudly, never a fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from typing import Optional

import numpy as np

from .model_card import DartModelCard

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libdart_stepper.so"
LIB_PATH = os.environ.get("DART_STEPPER_LIB", os.path.join(_HERE, LIB_NAME))

# error codes / keys (include/dart_stepper.h)
DART_OK, E_INVALID, E_NO_DEVICE, E_UNSUPPORTED, E_HIP, E_PENDING, E_NOT_PENDING = 0, -1, -2, -3, -4, -5, -6
Q_NUM_ENVS, Q_NDOFS, Q_OBS_DIM, Q_ACT_DIM, Q_FRAME_SKIP, Q_PRECISION, Q_DEVICE, Q_LCP_SLOTS, Q_STATIC_KERNEL, Q_MAX_CONTACTS, Q_LDS_BYTES, Q_LANE_KERNEL = range(12)
(CFG_SOLVER, CFG_ITERS_STAGE1, CFG_ITERS_STAGE2, CFG_AUTORESET, CFG_SEED, CFG_ENV_OFFSET, CFG_BLOCK_THREADS, CFG_STATS,
 CFG_EPISODE_STATS, CFG_CONTACT_REPORT, CFG_DEBUG_FORCE_FALLBACK, CFG_LAUNCH_ORDER, _CFG_RETIRED_12, CFG_HOST_DMA) = range(14)
SOLVER_BPP, SOLVER_PGS = 0, 1

EXPORTS = [
    "dart_last_error", "dart_create", "dart_destroy", "dart_query", "dart_configure", "dart_reset",
    "dart_set_state", "dart_get_state", "dart_step", "dart_step_async", "dart_step_wait",
    "dart_step_device", "dart_reset_device", "dart_sync", "dart_time_steps", "dart_get_counters", "dart_get_stats", "dart_debug_dump", "dart_seed_mt19937", "dart_get_dynamics", "dart_get_episode_stats", "dart_set_ext_force", "dart_set_task_state", "dart_host_views", "dart_get_contacts", "dart_get_constraint_forces", "dart_get_body_poses", "dart_snapshot", "dart_restore", "dart_timer_mark", "dart_timer_elapsed",
    "dart_output_layout", "dart_alloc_output", "dart_free_output", "dart_register_output", "dart_unregister_output", "dart_step_async_to",
    "dart_register_host_buffer", "dart_unregister_host_buffer", "dart_device_outputs",
]


class StepperError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dart_stepper error %d: %s" % (code, msg))
        self.code = code


class AlreadyPendingCallError(StepperError):
    """Mirrors gym.error.AlreadyPendingCallError (reference gym/error.py:143-150)."""


class NoAsyncCallError(StepperError):
    """Mirrors gym.error.NoAsyncCallError (reference gym/error.py:152-159)."""


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and declare prototypes.  Raises OSError when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError("%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    # PyTorch-ROCm wheels bundle their own HIP / HSA runtime under torch/lib.  Whichever runtime opens the GPU first owns it
    # for the process, and a second HSA runtime then reports "No HIP GPUs are available": if torch is installed, let it
    # load its runtime first (this library then shares the one HSA instance, and torch tensors' device pointers can be
    # handed to dart_step_device).  No torch -> nothing to coordinate with.
    try:
        import torch  # noqa: F401
    except Exception:   # pragma: no cover
        pass
    L = C.CDLL(p)
    vp, i64, dp, fp, u8 = C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    L.dart_last_error.restype = C.c_char_p
    L.dart_last_error.argtypes = [vp]
    L.dart_create.argtypes = [C.POINTER(DartModelCard), i64, C.c_int, C.c_int, C.POINTER(vp)]
    L.dart_destroy.argtypes = [vp]
    L.dart_query.argtypes = [vp, C.c_int, C.POINTER(i64)]
    L.dart_configure.argtypes = [vp, C.c_int, C.c_double]
    L.dart_reset.argtypes = [vp, u8, dp, dp, fp]
    L.dart_set_state.argtypes = [vp, dp, dp]
    L.dart_get_state.argtypes = [vp, dp, dp]
    L.dart_step.argtypes = [vp, fp, fp, dp, u8, u8]
    L.dart_step_async.argtypes = [vp, fp]
    L.dart_step_wait.argtypes = [vp, fp, dp, u8, u8]
    L.dart_step_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.dart_reset_device.argtypes = [vp, vp, vp, vp]
    L.dart_sync.argtypes = [vp]
    L.dart_seed_mt19937.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    L.dart_get_dynamics.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dart_get_contacts.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int32]
    L.dart_get_constraint_forces.argtypes = [vp, C.POINTER(C.c_double)]
    L.dart_snapshot.argtypes = [vp, C.c_void_p, C.POINTER(C.c_uint64)]
    L.dart_restore.argtypes = [vp, C.c_void_p, C.c_uint64]
    L.dart_get_body_poses.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dart_set_ext_force.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.dart_set_task_state.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_double)]
    L.dart_output_layout.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dart_alloc_output.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.dart_free_output.argtypes = [C.c_void_p]
    L.dart_register_output.argtypes = [vp, C.c_void_p]
    L.dart_unregister_output.argtypes = [vp, C.c_void_p]
    L.dart_step_async_to.argtypes = [vp, C.c_void_p, C.c_void_p]     # (const float* actions as an address: see _addr)
    L.dart_register_host_buffer.argtypes = [vp, C.c_void_p, C.c_uint64]
    L.dart_unregister_host_buffer.argtypes = [vp, C.c_void_p]
    L.dart_device_outputs.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.dart_host_views.argtypes = [vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint8))]
    L.dart_get_episode_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int]
    L.dart_debug_dump.argtypes = [vp, dp]
    L.dart_get_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.dart_get_counters.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    L.dart_time_steps.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, dp]
    L.dart_timer_mark.argtypes = [vp, C.c_int]
    L.dart_timer_elapsed.argtypes = [vp, dp]
    for name in EXPORTS:
        if name != "dart_last_error":
            getattr(L, name).restype = C.c_int
    if path is None:
        _lib = L
    return L


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _addr(a):
    """address of a numpy array's first byte (for a void* / float* argument): `a.ctypes.data_as(...)` builds a helper object and a typed
    pointer per call -- 4 us -- which is real money on a 150 us step (round 6: 20 us of Python per host step, measured with a no-op library)"""
    return a.__array_interface__["data"][0]


class _Lease:
    """What the arrays of one step are views of: exposes the step's output block through the array interface and gives the block back to
    the pool when the last array derived from it is garbage (numpy keeps this object alive as their base).  Round 6: replaces a ctypes
    array + weakref.finalize per step (3.8 us) by one slotted object (1.6 us); same ownership rule (see HipStepper._free_block)."""
    __slots__ = ("__array_interface__", "ent")

    def __init__(self, ai, ent):
        self.__array_interface__ = ai
        self.ent = ent
        ent[1] = True

    def __del__(self):
        try:
            self.ent[1] = False
        except Exception:      # interpreter shutdown
            pass


class _PinnedBlock:
    """Owner of one output block allocated by dart_alloc_output (driver page-locked memory; round 6): numpy arrays made from it keep it
    alive as their base, and the memory goes back (dart_free_output, which needs no handle) when the last of them is garbage -- the lifetime
    rule numpy's own memory had while the blocks were numpy arrays locked with hipHostRegister: results stay readable after close()."""
    __slots__ = ("__array_interface__", "_free", "_addr")

    def __init__(self, lib, addr, nbytes):
        self._free, self._addr = lib.dart_free_output, addr
        self.__array_interface__ = {"data": (addr, False), "shape": (int(nbytes),), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._free(self._addr)
        except Exception:      # interpreter shutdown
            pass


class HipStepper:
    """N batched worlds resident on one MI355X (one handle per GPU)."""

    def __init__(self, card: DartModelCard, num_envs: int, device: int = 0, precision: int = 64):
        """precision: 64 (default) = the kernels' fp64 instantiation, the mode that meets the north star's tolerance (RMS state
        divergence < 1e-4 over 1 000 env-steps; measured ~1e-13).  32 = the fast mode: 15-40 % quicker, but it does NOT meet that
        tolerance (untrimmed RMS q / dq over 1 000 steps: Hopper 4e-6 / 2e-4, Walker2d 1e-3 / 4e-2 -- a contact or limit event taken
        a 2 ms substep early decorrelates that env for the rest of its episode)."""
        self.L = load_library()
        self.card = card
        self.h = C.c_void_p()
        rc = self.L.dart_create(C.byref(card), int(num_envs), int(device), int(precision), C.byref(self.h))
        if rc != DART_OK:
            msg = self.L.dart_last_error(None).decode()
            self.h = None
            raise StepperError(rc, msg)
        self.num_envs = int(num_envs)
        self.ndofs, self.obs_dim, self.act_dim = card.ndofs, card.obs_dim, card.act_dim
        self.device, self.precision = device, precision

    # -- plumbing --
    def _check(self, rc):
        if rc == DART_OK:
            return
        msg = self.L.dart_last_error(self.h).decode()
        if rc == E_PENDING:
            raise AlreadyPendingCallError(rc, msg)
        if rc == E_NOT_PENDING:
            raise NoAsyncCallError(rc, msg)
        raise StepperError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.L.dart_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query(self, what: int) -> int:
        out = C.c_int64(0)
        self._check(self.L.dart_query(self.h, what, C.byref(out)))
        return out.value

    def configure(self, key: int, value: float):
        self._check(self.L.dart_configure(self.h, key, float(value)))

    def seed_mt19937(self, keys, key_len):
        """keys (N,2) uint32, key_len (N,) int32: the words of seeding.hash_seed(seed_i) (see include/dart_stepper.h)."""
        k = np.ascontiguousarray(keys, dtype=np.uint32).reshape(self.num_envs, 2)
        n = np.ascontiguousarray(key_len, dtype=np.int32).reshape(self.num_envs)
        self._check(self.L.dart_seed_mt19937(self.h, _ptr(k, C.c_uint32), _ptr(n, C.c_int32)))

    def set_task_state(self, mask, values):
        """(N, <=4) per-env task state (reach target) for the masked envs; call before reset()."""
        v = np.zeros((self.num_envs, 4), dtype=np.float64)
        vv = np.asarray(values, dtype=np.float64).reshape(self.num_envs, -1)
        v[:, :vv.shape[1]] = vv
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.L.dart_set_task_state(self.h, _ptr(m, C.c_uint8) if m is not None else None, _ptr(v, C.c_double)))

    def set_ext_force(self, body, force):
        """(N, 3) world-frame forces on body `body` in every substep from now on (None = off); generic kernel only."""
        if force is None:
            self._check(self.L.dart_set_ext_force(self.h, 0, None))
        else:
            f = np.ascontiguousarray(force, dtype=np.float64).reshape(self.num_envs, 3)
            self._check(self.L.dart_set_ext_force(self.h, int(body), _ptr(f, C.c_double)))

    def episode_stats(self, per_env=True, clear_totals=False):
        """-> (last_return (N,), last_length (N,), totals [sum_return, sum_length, count]); needs CFG_EPISODE_STATS."""
        r = np.empty(self.num_envs, dtype=np.float64) if per_env else None
        l = np.empty(self.num_envs, dtype=np.int32) if per_env else None
        tot = np.zeros(3, dtype=np.float64)
        self._check(self.L.dart_get_episode_stats(self.h, _ptr(r, C.c_double) if per_env else None,
                                                  _ptr(l, C.c_int32) if per_env else None, _ptr(tot, C.c_double), int(clear_totals)))
        return r, l, tot

    def contacts(self, max_contacts=None):
        """Contacts of the last world step of the last env-step (pydart2 world.collision_result.contacts); needs
        CFG_CONTACT_REPORT.  -> count (N,), bodies (N, K, 2) int32 {a, b; b = -1 ground}, point (N, K, 3), force on a (N, K, 3)."""
        k = self.query(Q_MAX_CONTACTS) if max_contacts is None else int(max_contacts)
        n = self.num_envs
        cnt = np.empty(n, dtype=np.int32); bod = np.empty((n, k, 2), dtype=np.int32); pf = np.empty((n, k, 6), dtype=np.float64)
        self._check(self.L.dart_get_contacts(self.h, _ptr(cnt, C.c_int32), _ptr(bod, C.c_int32), _ptr(pf, C.c_double), k))
        return cnt, bod, pf[:, :, :3], pf[:, :, 3:]

    def snapshot(self) -> np.ndarray:
        """Exact checkpoint (state, counters, MT19937 bank, task state, episode statistics) as a uint8 array."""
        nb = C.c_uint64(0)
        self._check(self.L.dart_snapshot(self.h, None, C.byref(nb)))
        buf = np.empty(nb.value, dtype=np.uint8)
        self._check(self.L.dart_snapshot(self.h, buf.ctypes.data_as(C.c_void_p), C.byref(nb)))
        return buf

    def restore(self, buf: np.ndarray):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._check(self.L.dart_restore(self.h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def body_poses(self):
        """-> rotation (N, nbodies, 3, 3), origin (N, nbodies, 3), com (N, nbodies, 3): pydart2 bodynode.T / .com() of every body."""
        n, nb = self.num_envs, self.card.nbodies
        R = np.empty((n, nb, 3, 3), dtype=np.float64); p = np.empty((n, nb, 3), dtype=np.float64); c = np.empty((n, nb, 3), dtype=np.float64)
        self._check(self.L.dart_get_body_poses(self.h, _ptr(R, C.c_double), _ptr(p, C.c_double), _ptr(c, C.c_double)))
        return R, p, c

    def constraint_forces(self):
        """pydart2 skel.constraint_forces() of the last world step, (N, ndofs); needs CFG_CONTACT_REPORT."""
        cf = np.empty((self.num_envs, self.ndofs), dtype=np.float64)
        self._check(self.L.dart_get_constraint_forces(self.h, _ptr(cf, C.c_double)))
        return cf

    def dynamics(self, mass=True, bias=True):
        """-> (M (N, n, n), c (N, n)): pydart2's skel.M and skel.c for every env (None for the one not requested)."""
        n, d = self.num_envs, self.ndofs
        M = np.empty((n, d, d), dtype=np.float64) if mass else None
        c = np.empty((n, d), dtype=np.float64) if bias else None
        self._check(self.L.dart_get_dynamics(self.h, _ptr(M, C.c_double) if mass else None, _ptr(c, C.c_double) if bias else None))
        return M, c

    # -- state --
    def reset(self, mask=None, qpos_noise=None, qvel_noise=None, want_obs=True):
        n = self.num_envs
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        qn = None if qpos_noise is None else np.ascontiguousarray(qpos_noise, dtype=np.float64).reshape(n, self.ndofs)
        vn = None if qvel_noise is None else np.ascontiguousarray(qvel_noise, dtype=np.float64).reshape(n, self.ndofs)
        obs = np.empty((n, self.obs_dim), dtype=np.float32) if want_obs else None
        self._check(self.L.dart_reset(self.h, _ptr(m, C.c_uint8), _ptr(qn, C.c_double), _ptr(vn, C.c_double),
                                      _ptr(obs, C.c_float)))
        return obs

    def set_state(self, q, dq):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(self.num_envs, self.ndofs)
        dq = np.ascontiguousarray(dq, dtype=np.float64).reshape(self.num_envs, self.ndofs)
        self._check(self.L.dart_set_state(self.h, _ptr(q, C.c_double), _ptr(dq, C.c_double)))

    def get_state(self):
        q = np.empty((self.num_envs, self.ndofs), dtype=np.float64)
        dq = np.empty_like(q)
        self._check(self.L.dart_get_state(self.h, _ptr(q, C.c_double), _ptr(dq, C.c_double)))
        return q, dq

    # -- stepping (host buffers) --
    _POOL_SETS = 4

    # ---- output blocks: one page-locked, caller-visible buffer per step in flight (include/dart_stepper.h, dart_step_async_to) ----
    def _layout(self):
        if getattr(self, "_out_layout", None) is None:
            tot = C.c_int64(0); off = (C.c_int64 * 4)()
            self._check(self.L.dart_output_layout(self.h, C.byref(tot), off))
            self._out_layout = (int(tot.value), [int(x) for x in off])
        return self._out_layout

    def _free_block(self):
        """An output block no caller array refers to any more, or a new one (None when the pool is exhausted or pooling is
        off).  Layout of a block: [obs | reward f32 | done | truncated] exactly as the device block, then (N) float64 rewards -- the
        type gym.vector returns -- written by the same copy kernel (round 5; rounds 3-4 converted them on the host after every step).
        Ownership is explicit (round 4; it used to be inferred from sys.getrefcount): the arrays a step returns are views of a LEASE --
        a small object exposing the block through the array interface, made for that one step (_Lease; rounds 4-5: a ctypes array with a
        weakref finalizer) -- and numpy keeps the lease alive as the base of every view derived from them; when the last such view is gone
        the lease's destructor returns the block to the pool.  copy=True semantics
        of sync_vector_env.py:83 without a copy (and without the page faults of fresh multi-MB arrays every step), independent of
        who else holds references to the block itself (debuggers, profilers, a test's own bookkeeping)."""
        if not self.__dict__.get("output_pool", True):     # (attribute set by tools/bench_host_path.py for its A/B: round 2's staging path)
            return None
        pool = self.__dict__.setdefault("_blocks", [])      # [block array, leased?, its address (c_void_p), its array interface]
        for ent in pool:
            if not ent[1]:
                return ent[0]
        if len(pool) >= self._POOL_SETS:
            return None
        total, _ = self._layout()
        # memory the driver page-locks (dart_alloc_output), not a numpy array locked after the fact (dart_register_output: round 6 saw GPU
        # writes into such arrays fault, rarely -- "write access to a read-only page"; include/dart_stepper.h, profiles/r06_crash_hunt.txt)
        p = C.c_void_p()
        if self.L.dart_alloc_output(self.h, C.byref(p)) != DART_OK or not p.value:
            return None
        blk = np.asarray(_PinnedBlock(self.L, p.value, total))
        pool.append([blk, False, C.c_void_p(p.value), dict(blk.__array_interface__)])
        return blk

    def _lease(self, blk):
        """-> a uint8 array over `blk` whose views keep the block out of the pool until all of them are garbage"""
        ent = next(e for e in self._blocks if e[0] is blk)
        return np.asarray(_Lease(ent[3], ent))

    def register_host_buffer(self, arr):
        """page-lock a caller-owned numpy array for direct DMA (dart_register_host_buffer): `dart_step` arguments inside it skip the
        staging copies.  The caller keeps `arr` alive until unregister_host_buffer / close.  Actions are consumed before the call that
        takes them returns (the blocking `step_into` reads a registered array in place, `step_async` copies at call time), so the array
        may be refilled right after either call."""
        self._check(self.L.dart_register_host_buffer(self.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        self.__dict__.setdefault("_host_bufs", []).append(arr)

    def unregister_host_buffer(self, arr):
        self._check(self.L.dart_unregister_host_buffer(self.h, arr.ctypes.data_as(C.c_void_p)))
        self._host_bufs = [a for a in self.__dict__.get("_host_bufs", []) if a is not arr]

    def step_into(self, actions, obs, reward, done, truncated=None):
        """`dart_step` with caller-owned arrays, as a C caller binds it: actions (N, act) f32 in, obs (N, obs) f32 / reward (N) f64 /
        done (N) u8 / truncated (N) u8 out.  Arrays inside registered buffers are filled by DMA (include/dart_stepper.h)."""
        self._check(self.L.dart_step(self.h, _ptr(actions, C.c_float), _ptr(obs, C.c_float), _ptr(reward, C.c_double), _ptr(done, C.c_uint8),
                                     _ptr(truncated, C.c_uint8) if truncated is not None else None))

    def _block_views(self, blk):
        n = self.num_envs
        total, off = self._layout()
        blk = self._lease(blk)          # every array below is a view of this step's lease (see _free_block)
        obs = blk[off[0]:off[0] + 4 * n * self.obs_dim].view(np.float32).reshape(n, self.obs_dim)
        done = blk[off[2]:off[2] + n].view(np.bool_)        # the kernels write exactly 0 / 1
        trunc = blk[off[3]:off[3] + n].view(np.bool_)
        r64 = total - ((8 * n + 255) & ~255)     # the block's tail: float64 rewards, converted by the copy kernel (include/dart_stepper.h)
        rew = blk[r64:r64 + 8 * n].view(np.float64)
        return obs, rew, done, trunc

    def _outs(self):
        """Fresh output arrays (the staging path: library pinned buffer -> these)."""
        n = self.num_envs
        return (np.empty((n, self.obs_dim), dtype=np.float32), np.empty(n, dtype=np.float64),
                np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_async(self, actions, staged=False):
        """staged=True: the outputs go through the library's own pinned staging buffer (what step_wait(copy=False) returns views of).
        `actions` is copied before this returns: the caller may overwrite it while the step is in flight."""
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, self.act_dim)
        blk = None if staged else self._free_block()
        if blk is None:
            self._check(self.L.dart_step_async(self.h, _ptr(a, C.c_float)))
        else:
            ent = next(e for e in self._blocks if e[0] is blk)
            self._check(self.L.dart_step_async_to(self.h, _addr(a), ent[2]))
        self._pending_block = blk

    def _views(self):
        if getattr(self, "_host_views", None) is None:
            po, pr = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
            pd, pt = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)()
            self._check(self.L.dart_host_views(self.h, C.byref(po), C.byref(pr), C.byref(pd), C.byref(pt)))
            n = self.num_envs
            self._host_views = (np.ctypeslib.as_array(po, shape=(n, self.obs_dim)), np.ctypeslib.as_array(pr, shape=(n,)),
                                np.ctypeslib.as_array(pd, shape=(n,)), np.ctypeslib.as_array(pt, shape=(n,)))
        return self._host_views

    def device_outputs(self):
        """-> device addresses (ints) of the last host-buffer step's obs (N, obs_dim) f32 / reward (N) f32 / done (N) u8 / truncated (N) u8,
        still resident in HBM until the next step or reset (dart_device_outputs)."""
        p = [C.c_void_p() for _ in range(4)]
        self._check(self.L.dart_device_outputs(self.h, *[C.byref(x) for x in p]))
        return tuple(int(x.value) for x in p)

    def step_wait(self, copy=True):
        """copy=True: arrays the caller owns (views of a page-locked block of this step; see _free_block).  copy=False: views of the
        library's staging buffer, valid until the next call (only when the step went through it)."""
        blk, self._pending_block = getattr(self, "_pending_block", None), None
        if blk is not None:
            self._check(self.L.dart_step_wait(self.h, None, None, None, None))
            return self._block_views(blk)
        if not copy:
            self._check(self.L.dart_step_wait(self.h, None, None, None, None))
            o, r, d, t = self._views()
            return o, r.astype(np.float64), d.astype(np.bool_), t.astype(np.bool_)
        obs, rew, done, trunc = self._outs()
        self._check(self.L.dart_step_wait(self.h, _ptr(obs, C.c_float), _ptr(rew, C.c_double),
                                          _ptr(done, C.c_uint8), _ptr(trunc, C.c_uint8)))
        return obs, rew, done.view(np.bool_), trunc.view(np.bool_)     # the kernels write exactly 0 / 1

    # -- stepping (device pointers: ints / torch data_ptr()) --
    def step_device(self, d_actions, d_obs=0, d_reward=0, d_done=0, d_truncated=0, stream=0):
        self._check(self.L.dart_step_device(self.h, d_actions, d_obs or None, d_reward or None, d_done or None,
                                            d_truncated or None, stream or None))

    def reset_device(self, d_mask=0, d_obs=0, stream=0):
        self._check(self.L.dart_reset_device(self.h, d_mask or None, d_obs or None, stream or None))

    def counters(self):
        el = np.empty(self.num_envs, dtype=np.int32)
        ep = np.empty(self.num_envs, dtype=np.uint32)
        self._check(self.L.dart_get_counters(self.h, _ptr(el, C.c_int32), _ptr(ep, C.c_uint32)))
        return el, ep

    def solver_stats(self, clear=True):
        h = np.zeros(64, dtype=np.uint64)
        self._check(self.L.dart_get_stats(self.h, _ptr(h, C.c_uint64), int(clear)))
        return h[:32], h[32:]

    def debug_dump(self):
        out = np.zeros((self.num_envs, 160))
        self._check(self.L.dart_debug_dump(self.h, _ptr(out, C.c_double)))
        return out

    def sync(self):
        self._check(self.L.dart_sync(self.h))

    def timer_mark(self, which: int):
        """enqueue HIP event `which` (0 start, 1 stop) on the handle's stream"""
        self._check(self.L.dart_timer_mark(self.h, int(which)))

    def timer_elapsed(self) -> float:
        ms = C.c_double(0)
        self._check(self.L.dart_timer_elapsed(self.h, C.byref(ms)))
        return ms.value

    def time_steps(self, d_actions, action_batches, steps, d_obs=0, d_reward=0, d_done=0, d_truncated=0) -> float:
        ms = C.c_double(0)
        self._check(self.L.dart_time_steps(self.h, d_actions, int(action_batches), d_obs or None, d_reward or None,
                                           d_done or None, d_truncated or None, int(steps), C.byref(ms)))
        return ms.value
