#!/usr/bin/env python3
"""Where the 75 us between a 32 us step kernel and a 107 us `dart_step_device + dart_sync` go.

One process, DartHopper-v1 fp64, HBM-resident inputs and outputs (no PCIe payload at all), patterns timed with perf_counter over 400 calls:
  sync_only          dart_sync on an idle stream                       -> cost of the call itself
  step+sync  N=64    one wavefront: the kernel's own latency (~30 us, a lone wave) + the submit -> complete round trip
  step+sync  N=65536 the drop-in pattern
  2step+sync N=65536 two launches, one wait: (this - step+sync) = the kernel's duration when the device does not idle in between
  8step+sync N=65536 eight launches, one wait: / 8 -> back-to-back rate
Run it bare and under `rocprofv3 --kernel-trace --stats` (kernel durations inside the step-sync-step pattern), and with the HSA / ROCclr wait
knobs of the environment (tools/gpu/host_latency.sh): HSA_ENABLE_INTERRUPT=0 (signals polled, no interrupt), ROC_ACTIVE_WAIT_TIMEOUT=<us>.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for

card = card_for("DartHopper-v1")
tag = os.environ.get("PROBE_TAG", "default")


def mk(n):
    s = st.HipStepper(card, n, precision=64)
    s.configure(st.CFG_AUTORESET, 1)
    s.reset(None, None, None, want_obs=False)
    a = (torch.rand((16, n, 3), device="cuda") * 2 - 1).contiguous()     # a ring of 16 action batches (PROBE_CONST=1: one batch, what rounds 3-4 timed)
    bufs = (torch.empty((n, 11), device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"),
            torch.empty(n, dtype=torch.uint8, device="cuda"))
    class Ptr:
        k = 0
        def __iter__(self):
            Ptr.k += 1
            off = 0 if os.environ.get("PROBE_CONST") == "1" else (Ptr.k % 16) * n * 3 * 4
            return iter((a.data_ptr() + off,) + tuple(b.data_ptr() for b in bufs))
    return s, Ptr(), (a, bufs)


def bench(name, f, k=400, warm=300):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    us = (time.perf_counter() - t0) / k * 1e6
    print("[%s] %-28s %8.1f us" % (tag, name, us), flush=True)
    return us


big, pb, keep1 = mk(65536)
small, ps, keep2 = mk(64)
torch.cuda.synchronize()
r = {}
if os.environ.get("PROBE_ONLY") == "s1":      # (under rocprofv3: only the drop-in pattern, so that every gap in the trace is a host round trip)
    bench("step+sync N=65536", lambda: (big.step_device(*pb), big.sync()), k=1000)
    sys.exit(0)
r["sync_only"] = bench("sync_only", lambda: big.sync())
r["s64"] = bench("step+sync N=64", lambda: (small.step_device(*ps), small.sync()))
r["s1"] = bench("step+sync N=65536", lambda: (big.step_device(*pb), big.sync()))
r["s2"] = bench("2step+sync N=65536", lambda: (big.step_device(*pb), big.step_device(*pb), big.sync()))
r["s8"] = bench("8step+sync N=65536", lambda: ([big.step_device(*pb) for _ in range(8)], big.sync()), k=200)
kern = r["s2"] - r["s1"]
print("[%s] kernel when the device stays busy (2step - 1step): %.1f us; back-to-back rate (8step / 8): %.1f us; fixed submit->complete "
      "round trip of one step (1step - kernel): %.1f us; of a one-wave launch: %.1f us" % (tag, kern, r["s8"] / 8, r["s1"] - kern, r["s64"] - 30.0), flush=True)
