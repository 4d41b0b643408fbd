"""ctypes binding of tests/kernel_emu/libdart_planar_emu.so -- TEST INFRASTRUCTURE ONLY.

The planar device kernels (dart_env_amd/csrc/planar_kernel.hpp) compiled for the host by g++ against a stand-in
<hip/hip_runtime.h>, one lane at a time.  It exists so that the kernels' arithmetic can be held against the fp64 oracle in
the GPU-less development container (`pytest -m "not gpu"`); nothing under dart_env_amd/ loads it and the product library has
no CPU path.  `EmuStepper` mirrors the subset of `dart_env_amd.stepper.HipStepper` the parity helpers use.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from dart_env_amd.model_card import DartModelCard
from dart_env_amd import stepper as st

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_emu")
_LIB = os.environ.get("DART_EMU_LIB", os.path.join(_DIR, "libdart_planar_emu.so"))   # the override serves bitwise A/B checks of kernel edits
_TREE_LIB = os.path.join(_DIR, "libdart_spatial_emu.so")   # the tree kernel (one env per wavefront) on the fiber runtime of fake_wave_include/
_libs = {}


_WAVE_LIB = os.path.join(_DIR, "libdart_planar_wave_emu.so")   # the lane kernels as whole 64-lane waves on the same fiber runtime


def lib(tree=False, waves=False):
    path = _TREE_LIB if tree else (_WAVE_LIB if waves else _LIB)
    _lib = _libs.get(path)
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _DIR] + (["libdart_spatial_emu.so"] if tree else ["libdart_planar_wave_emu.so"] if waves else []))
        L = C.CDLL(path)
        vp, dp, fp, u8 = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.emu_create.restype = vp
        L.emu_create.argtypes = [C.POINTER(DartModelCard), C.c_int64, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.emu_destroy.argtypes = [vp]
        L.emu_is_static.argtypes = [vp]
        L.emu_slots.argtypes = [vp]
        L.emu_lds_bytes.argtypes = [vp]; L.emu_lds_bytes.restype = C.c_longlong
        L.emu_set_solver.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.emu_enable_stats.argtypes = [vp, C.c_int]
        L.emu_force_slow.argtypes = [vp, C.c_int]
        L.emu_set_task_state.argtypes = [vp, C.POINTER(C.c_uint8), dp]
        L.emu_set_ext_force.argtypes = [vp, C.c_int, dp]
        L.emu_contact_report.argtypes = [vp, C.c_int]
        L.emu_max_contacts.argtypes = [vp]
        L.emu_get_contacts.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, C.c_int]
        L.emu_get_constraint_forces.argtypes = [vp, dp]
        L.emu_get_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.emu_reset.argtypes = [vp, u8, dp, dp, fp, C.c_uint64, C.c_uint64, C.c_int]
        L.emu_step.argtypes = [vp, fp, fp, fp, u8, u8, C.c_int, C.c_uint64, C.c_uint64]
        L.emu_state.argtypes = [vp, dp, dp, C.c_int]
        L.emu_counters.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        _lib = _libs[path] = L
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def poison_static_lds(waves=True, tree=False):
    """Register every static __shared__ array of an emulator library (function-local statics of the host build: `slow_lds` of the lane
    kernels) with its launcher, so that DART_EMU_POISON_LDS -- read once per process by the launcher -- fills them, like the dynamic LDS
    block, with a byte pattern before every workgroup.  Addresses = load base + symbol value from the library's symbol table."""
    L = lib(tree=tree, waves=waves)
    path = _TREE_LIB if tree else (_WAVE_LIB if waves else _LIB)
    syms = {}
    for line in subprocess.check_output(["nm", "-S", "--defined-only", path], text=True).splitlines():
        p = line.split()
        if len(p) == 4:
            syms[p[3]] = (int(p[0], 16), int(p[1], 16))
    base = C.cast(L.emu_create, C.c_void_p).value - syms["emu_create"][0]
    L.emu_register_static_lds.argtypes = [C.c_void_p, C.c_ulonglong]
    n = 0
    for name, (off, size) in syms.items():
        if name.endswith("slow_lds"):
            L.emu_register_static_lds(C.c_void_p(base + off), size); n += 1
    return n


class EmuStepper:
    def __init__(self, card, num_envs, precision=64, allow_static=True, tree=False, waves=False):
        """tree=True: the tree kernel (csrc/spatial_*.hpp) instead of the lane kernels -- any card.  waves=True: the lane kernels run as
        whole 64-lane wavefronts (fibers): wave votes, the wave-served fallback and the hand-off behave as on the device"""
        self.L = lib(tree, waves)
        self.card, self.n, self.precision = card, int(num_envs), precision
        why = C.create_string_buffer(512)
        self.h = self.L.emu_create(C.byref(card), self.n, precision, int(allow_static), why, 512)
        if not self.h:
            raise RuntimeError("no planar kernel for this card: " + why.value.decode())
        self.autoreset, self.seed, self.env_offset = 0, 0, 0
        self.nd, self.na, self.no = card.ndofs, card.act_dim, card.obs_dim

    def close(self):
        if getattr(self, "h", None):
            self.L.emu_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def is_static(self):
        return bool(self.L.emu_is_static(self.h))

    @property
    def lds_bytes(self):
        """DART_Q_LDS_BYTES of the product library for this card: the step kernel's LDS block per workgroup"""
        return int(self.L.emu_lds_bytes(self.h))

    def configure(self, key, value):
        if key == st.CFG_AUTORESET: self.autoreset = int(value != 0)
        elif key == st.CFG_SEED: self.seed = int(value)
        elif key == st.CFG_ENV_OFFSET: self.env_offset = int(value)
        elif key in (st.CFG_SOLVER, st.CFG_ITERS_STAGE1, st.CFG_ITERS_STAGE2):   # 0 sweeps / iterations = the implementation's default
            sv = getattr(self, "_solver_cfg", [0, 0, 0])
            sv[(st.CFG_SOLVER, st.CFG_ITERS_STAGE1, st.CFG_ITERS_STAGE2).index(key)] = int(value)
            self._solver_cfg = sv
            self.L.emu_set_solver(self.h, sv[0], sv[1], sv[2])
        elif key == st.CFG_STATS: self.L.emu_enable_stats(self.h, int(value != 0))
        else: raise ValueError(key)

    def set_ext_force(self, body, force):
        f = None if force is None else np.ascontiguousarray(force, dtype=np.float64)
        rc = self.L.emu_set_ext_force(self.h, int(body), _p(f, C.c_double))
        if rc != 0:
            raise RuntimeError("set_ext_force rc=%d" % rc)

    def enable_contact_report(self, on=True):
        assert self.L.emu_contact_report(self.h, int(on)) == 0

    def contacts(self):
        K = self.L.emu_max_contacts(self.h)
        cnt = np.zeros(self.n, dtype=np.int32); bod = np.zeros((self.n, K, 2), dtype=np.int32); pf = np.zeros((self.n, K, 6))
        assert self.L.emu_get_contacts(self.h, _p(cnt, C.c_int32), _p(bod, C.c_int32), _p(pf, C.c_double), K) == 0
        return cnt, bod, pf[:, :, :3], pf[:, :, 3:]

    def constraint_forces(self):
        cf = np.zeros((self.n, self.nd))
        assert self.L.emu_get_constraint_forces(self.h, _p(cf, C.c_double)) == 0
        return cf

    def set_task_state(self, mask, values):
        """(N, <=4) per-env task state (reach target) for the masked envs; call before reset() -- as HipStepper.set_task_state"""
        v = np.zeros((self.n, 4), dtype=np.float64)
        vals = np.asarray(values, dtype=np.float64).reshape(self.n, -1)
        v[:, :vals.shape[1]] = vals
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = self.L.emu_set_task_state(self.h, _p(m, C.c_uint8) if m is not None else None, _p(v, C.c_double))
        if rc != 0:
            raise RuntimeError("set_task_state rc=%d" % rc)

    def force_slow(self, on=True):
        """every env with a contact goes through the single-lane fallback solver (validates it against the oracle)"""
        self.L.emu_force_slow(self.h, int(on))

    def solver_stats(self):
        h = np.zeros(64, dtype=np.uint64)
        self.L.emu_get_stats(self.h, _p(h, C.c_uint64))
        return h[:32], h[32:]

    def reset(self, mask=None, qpos_noise=None, qvel_noise=None, want_obs=True):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        qn = None if qpos_noise is None else np.ascontiguousarray(qpos_noise, dtype=np.float64)
        vn = None if qvel_noise is None else np.ascontiguousarray(qvel_noise, dtype=np.float64)
        obs = np.zeros((self.n, self.no), dtype=np.float32) if want_obs else None
        self.L.emu_reset(self.h, _p(m, C.c_uint8), _p(qn, C.c_double), _p(vn, C.c_double), _p(obs, C.c_float), self.seed, self.env_offset, 0)
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        obs = np.zeros((self.n, self.no), dtype=np.float32); rew = np.zeros(self.n, dtype=np.float32)
        done = np.zeros(self.n, dtype=np.uint8); trunc = np.zeros(self.n, dtype=np.uint8)
        self.L.emu_step(self.h, _p(a, C.c_float), _p(obs, C.c_float), _p(rew, C.c_float), _p(done, C.c_uint8), _p(trunc, C.c_uint8),
                        self.autoreset, self.seed, self.env_offset)
        return obs, rew.astype(np.float64), done.astype(bool), trunc.astype(bool)

    def get_state(self):
        q = np.zeros((self.n, self.nd)); dq = np.zeros((self.n, self.nd))
        self.L.emu_state(self.h, _p(q, C.c_double), _p(dq, C.c_double), 0)
        return q, dq

    def set_state(self, q, dq):
        q = np.ascontiguousarray(q, dtype=np.float64); dq = np.ascontiguousarray(dq, dtype=np.float64)
        self.L.emu_state(self.h, _p(q, C.c_double), _p(dq, C.c_double), 1)

    def counters(self):
        el = np.zeros(self.n, dtype=np.int32); ep = np.zeros(self.n, dtype=np.uint32)
        self.L.emu_counters(self.h, _p(el, C.c_int32), _p(ep, C.c_uint32))
        return el, ep
