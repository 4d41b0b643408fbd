#!/usr/bin/env python3
"""Kernel time per batched step as a function of how long the batch has been running (round 6, VERDICT r5 item 3: DartWalker2d-v1 was quoted
at 100.6 / 101.9 / 113.5 us depending on which window of the rollout was timed).  All envs of a fresh batch start their first episode
together; a wave takes as long as its slowest lane, so the kernel time moves until the episodes have de-synchronised.  Prints the HIP-event
time per step over consecutive windows of W steps from reset on.   python tools/gpu/kernel_time_windows.py [env-id ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import HipBenchEnv, default_envs
W = int(os.environ.get("W", "100")); NW = int(os.environ.get("NW", "30")); RING = int(os.environ.get("RING", "16"))
# RING: distinct action batches resident in HBM, cycled through (bench.py: 16).  The workload is "random actions": a SHORT ring is a periodic
# forcing -- DartWalker2d-v1 fp64 measured 113-124 us per step with a ring of 8, 104 with 16 (profiles/r06_walker2d_windows.txt)
for env_id in (sys.argv[1:] or ["DartWalker2d-v1", "DartHopper-v1"]):
    for prec in (64, 32):
        b = HipBenchEnv(env_id, default_envs(env_id), 0, prec, 0, ring=RING)
        b.reset(); b.run(3); b.sync()
        ms = [b.timed_steps(W) for _ in range(NW)]
        print("%s f%d, ring %d, windows of %d steps from reset: %s us" % (env_id, prec, RING, W, " ".join("%.1f" % (m * 1e3) for m in ms)), flush=True)
        print("   first window %.1f, windows 2-5 %.1f, last ten windows %.1f us; done fraction at the end %.4f" %
              (ms[0] * 1e3, sum(ms[1:5]) / 4 * 1e3, sum(ms[-10:]) / 10 * 1e3, b.done_fraction()))
        b.close()
