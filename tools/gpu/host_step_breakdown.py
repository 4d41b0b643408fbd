#!/usr/bin/env python3
"""Where the ~210 us of a host-buffer step go (DartHopper-v1 x 65 536, fp64): incremental variants in one process, 300 steps each after
warm-up.  python tools/gpu/host_step_breakdown.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for

n = 65536
card = card_for("DartHopper-v1")
ring = np.random.RandomState(0).uniform(-1, 1, (16, n, 3)).astype(np.float32)   # a ring of action batches (round 5; one constant batch triples the kernel time)
a = ring[0]
d_ring = torch.from_numpy(ring).cuda()
_cnt = [0]


def nxt():
    _cnt[0] += 1
    return _cnt[0] % 16


class _A:      # (the device pointer of the next action batch)
    def data_ptr(self):
        return d_ring[nxt()].data_ptr()


d_a = _A()
d_obs = torch.empty((n, 11), device="cuda"); d_rew = torch.empty(n, device="cuda"); d_done = torch.empty(n, dtype=torch.uint8, device="cuda"); d_tr = torch.empty(n, dtype=torch.uint8, device="cuda")


def mk(autoreset=1, mt=False):
    s = st.HipStepper(card, n, precision=64)
    if mt:
        from dart_env_amd import seeding
        k, l = seeding.mt_keys(list(range(n))); s.seed_mt19937(k, l)
    s.configure(st.CFG_AUTORESET, autoreset)
    s.reset(None, None, None, want_obs=False)
    return s


def bench(name, f, k=300):
    for _ in range(400):
        f()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    print("%-78s %.1f us/step" % (name, (time.perf_counter() - t0) / k * 1e6), flush=True)


s1 = mk(); bench("device step (philox reset in-kernel) + dart_sync every step", lambda: (s1.step_device(d_a.data_ptr(), d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_tr.data_ptr()), s1.sync()))
s2 = mk(mt=True); bench("device step (MT19937 bank: step + draw + reset kernels) + dart_sync", lambda: (s2.step_device(d_a.data_ptr(), d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_tr.data_ptr()), s2.sync()))
s3 = mk(); bench("host step, philox, block path (memcpy actions, kernel reads them, copy kernel out)", lambda: s3.step(ring[nxt()]))
s4 = mk(mt=True); bench("host step, MT19937 bank, block path", lambda: s4.step(ring[nxt()]))
def into(arrs):
    np.copyto(arrs[0], ring[nxt()])
    return arrs


reg = [a.copy(), np.zeros((n, 11), np.float32), np.zeros(n, np.float64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)]
s5 = mk()
for x in reg:
    s5.register_host_buffer(x)
bench("host step, philox, registered caller arrays (no host memcpy at all)", lambda: s5.step_into(*into(reg)))
bench("  ... of which obs only (reward / done / truncated NULL)", lambda: s5.L.dart_step(s5.h, into(reg)[0].ctypes.data_as(C.POINTER(C.c_float)), reg[1].ctypes.data_as(C.POINTER(C.c_float)), None, None, None))
bench("  ... done flags only", lambda: s5.L.dart_step(s5.h, into(reg)[0].ctypes.data_as(C.POINTER(C.c_float)), None, None, reg[3].ctypes.data_as(C.POINTER(C.c_uint8)), None))
t0 = time.perf_counter()
for _ in range(300):
    np.copyto(reg[0], a)
print("host memcpy of the actions alone (0.8 MB): %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
