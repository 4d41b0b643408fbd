#!/usr/bin/env python3
"""The host-buffer boundary as a C caller binds it, A/B in ONE process on one box (PCIe-inclusive; never bench.py's `value`):

  staging     dart_step(actions, obs, reward, done, trunc) with plain caller arrays: memcpy into the library's pinned block, H2D, kernel,
              one packed D2H, memcpy / float64 conversion into the caller's arrays
  registered  the same call, same arrays, after dart_register_host_buffer on each: DMA straight from / into the caller's memory,
              float64 rewards made on the device
  block       dart_step_async_to + dart_step_wait into ONE registered output block (what DartVectorEnv uses)
  python      DartVectorEnv.step (numpy in / out, device MT19937 resets): the drop-in surface

Every variant runs in blocks of 100 steps, round-robin, after 600 warm-up steps (a fresh process runs its first few hundred steps at
half speed -- tools/gpu/host_loop_probe.py -- which is what round 3's 522 us for `dart_step` was).
    python tools/gpu/host_path_c.py [envs]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dart_env_amd  # noqa: E402
from dart_env_amd import stepper as st  # noqa: E402
from dart_env_amd.model_card import card_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
card = card_for("DartHopper-v1")
# a RING of action batches (round 5): one batch applied at every step -- what rounds 3-4 timed -- keeps the hoppers on the floor, where the
# step kernel takes 96 us instead of 32 (profiles/r05_host_path.txt); the caller of a real rollout writes new actions every step
ring = np.random.RandomState(0).uniform(-1, 1, (16, n, 3)).astype(np.float32)
a = ring[0]
_cnt = [0]


def nxt():
    _cnt[0] += 1
    return ring[_cnt[0] % 16]


def into(arrs):      # the caller's own (registered) action array gets this step's actions: a 0.8 MB host memcpy, part of what is timed
    np.copyto(arrs[0], nxt())
    return arrs


def caller_arrays():
    return (a.copy(), np.zeros((n, card.obs_dim), np.float32), np.zeros(n, np.float64), np.zeros(n, np.uint8), np.zeros(n, np.uint8))


def make(dma=None):
    """dma: DART_CFG_HOST_DMA of this handle: None = default (3: kernel reads actions from page-locked memory, copy kernel for the
    outputs), 'copy' = 0, hipMemcpyAsync both ways (rounds 1-3), 'zc_actions' = 1 / 'd2h_kernel' = 2: one leg each"""
    s = st.HipStepper(card, n, precision=64)
    if dma:
        s.configure(st.CFG_HOST_DMA, {"copy": 0, "zc_actions": 1, "d2h_kernel": 2}[dma])
    s.configure(st.CFG_AUTORESET, 1)
    s.reset(None, None, None, want_obs=False)
    return s


s_stage, s_reg, s_blk = make(), make(), make()
s_blk_copy, s_blk_zc, s_blk_dk, s_reg_copy = make("copy"), make("zc_actions"), make("d2h_kernel"), make("copy")

arr_reg_copy = caller_arrays()
for x in arr_reg_copy:
    s_reg_copy.register_host_buffer(x)
arr_stage, arr_reg = caller_arrays(), caller_arrays()
for x in arr_reg:
    s_reg.register_host_buffer(x)
venv = dart_env_amd.vector.make("DartHopper-v1", n)
venv.seed(0); venv.reset()

variants = {
    "staging    dart_step, plain caller arrays": lambda: s_stage.step_into(*into(arr_stage)),
    "registered dart_step, dart_register_host_buffer'd caller arrays": lambda: s_reg.step_into(*into(arr_reg)),
    "block      dart_step_async_to + dart_step_wait (HipStepper.step)": lambda: s_blk.step(nxt()),
    "python     DartVectorEnv.step, device MT19937 resets": lambda: venv.step(nxt()),
    "block      ... DART_CFG_HOST_DMA=copy (hipMemcpyAsync H2D + D2H, rounds 1-3)": lambda: s_blk_copy.step(nxt()),
    "block      ... DART_CFG_HOST_DMA=zc_actions (kernel reads pinned actions, D2H by hipMemcpyAsync)": lambda: s_blk_zc.step(nxt()),
    "block      ... DART_CFG_HOST_DMA=d2h_kernel (H2D by hipMemcpyAsync, outputs by copy kernel)": lambda: s_blk_dk.step(nxt()),
    "registered ... DART_CFG_HOST_DMA=copy": lambda: s_reg_copy.step_into(*into(arr_reg_copy)),
}
for f in variants.values():
    _cnt[0] = 0                       # every variant sees the same action sequence (the equality check below relies on it)
    for _ in range(600):
        f()
# the registered path must give the staging path's numbers: same seeds, same actions, same step count so far
same = all(np.array_equal(x, y) for x, y in zip(arr_stage[1:], arr_reg[1:]))
print("registered == staging outputs after 600 steps (obs, float64 rewards, done, truncated):", same)
tot = {k: [] for k in variants}
for rnd in range(4):
    for k, f in variants.items():
        t0 = time.perf_counter()
        for _ in range(100):
            f()
        tot[k].append((time.perf_counter() - t0) / 100 * 1e6)
for k, v in tot.items():
    best, med = min(v), sorted(v)[len(v) // 2]
    print("%-70s median %.1f us/step (min %.1f, rounds %s)  %.3e env-steps/s" % (k, med, best, " ".join("%.0f" % x for x in v), n / (med * 1e-6)))
assert same
