#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 5: the reference-exact MT19937 auto-reset in the step kernel's epilogue against the two launches behind it
(mt_draw_kernel + masked reset kernel; DART_CFG_HOST_DMA bit 3 keeps that path).  DartHopper-v1 x 65 536 fp64, a ring of 16 action batches:
  device-side   dart_step_device + dart_sync per step (HBM-resident actions and outputs)
  host surface  DartVectorEnv.step (numpy in, numpy out, gym.vector types) = bench.py's host_surface.copy_true
plus the pinned D2H ceiling of this box for one step's output block (hipMemcpyAsync from HBM into page-locked memory), which is what the
floor of a synchronous host step is made of.   python tools/gpu/mt_fused_ab.py [env-id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import dart_env_amd.vector as V
from dart_env_amd import seeding, stepper as st
from dart_env_amd.model_card import card_for

env_id = sys.argv[1] if len(sys.argv) > 1 else "DartHopper-v1"
n = 65536
card = card_for(env_id)
dev = torch.device("cuda", 0)
ring_t = (torch.rand((16, n, card.act_dim), device=dev) * 2 - 1).contiguous()
ring = ring_t.cpu().numpy()
obs = torch.empty((n, card.obs_dim), device=dev); rew = torch.empty(n, device=dev)
done = torch.empty(n, dtype=torch.uint8, device=dev); trunc = torch.empty(n, dtype=torch.uint8, device=dev)
keys, klen = seeding.mt_keys(list(range(n)))
for rep in range(2):
    for name, dma in (("fused (epilogue)", 3), ("split (two launches)", 3 | 8)):
        g = st.HipStepper(card, n, precision=64)
        g.seed_mt19937(keys, klen)
        g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_HOST_DMA, dma)
        g.reset(None, None, None, want_obs=False)
        stride = n * card.act_dim * 4
        for i in range(400):
            g.step_device(ring_t.data_ptr() + (i % 16) * stride, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), trunc.data_ptr())
        g.sync()
        K = 1000
        t0 = time.perf_counter()
        for i in range(K):
            g.step_device(ring_t.data_ptr() + (i % 16) * stride, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), trunc.data_ptr())
            g.sync()
        dt_sync = (time.perf_counter() - t0) / K * 1e6
        ms = g.time_steps(ring_t.data_ptr(), 16, 1000, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), trunc.data_ptr())
        g.close()
        print("%-22s device-side: %.1f us per step + sync, %.2f us per step back to back (HIP events)" % (name, dt_sync, ms * 1e3), flush=True)
    for name, dma in (("fused (epilogue)", 3), ("split (two launches)", 3 | 8)):
        venv = V.make(env_id, n, device=0, precision=64, copy=True)
        venv.seed(0)
        venv.env._stepper.configure(st.CFG_HOST_DMA, dma)
        venv.reset()
        for i in range(300):
            venv.step(ring[i % 16])
        K = 500
        t0 = time.perf_counter()
        for i in range(K):
            venv.step(ring[i % 16])
        dt = (time.perf_counter() - t0) / K * 1e6
        venv.close()
        print("%-22s DartVectorEnv.step: %.1f us per step (%.3e env-steps/s)" % (name, dt, n / dt * 1e6), flush=True)
# pinned D2H ceiling for one step's output block
nbytes = n * (card.obs_dim * 4 + 4 + 1 + 1)
src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
for _ in range(20):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200 * 1e6
print("pinned D2H of one step's output block (%d bytes): %.1f us per copy + sync = %.1f GB/s" % (nbytes, dt, nbytes / dt / 1e3))
