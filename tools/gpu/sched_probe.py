"""Does dispatching the expensive envs first shorten a tree-kernel launch?  (DART_CFG_LAUNCH_ORDER: workgroups are dispatched in the order of
the per-env durations recorded at the previous step, longest first.)  Alternates blocks of ten launches in index order and in that
order and reports the mean kernel time of each.      python tools/gpu/sched_probe.py [env-id] [precision] [envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench      # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "DartHumanWalker-v1"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
b = bench.HipBenchEnv(env_id, n, 0, prec, 0, ring=16, ring_seed=1234, all_bodies_collide=None, configure=[])
st = b.st
b.reset(); b.run(40); b.sync()
t = {0: [], 1: []}
step = 40
for block in range(8):
    on = block % 2
    b.env.configure(st.CFG_LAUNCH_ORDER, on)
    b.run(2, step); b.sync(); step += 2            # (the first launches after switching it on have no durations to go by yet)
    for i in range(10):
        b.mark(0); b.run(1, step); b.mark(1); b.sync()
        t[on].append(b.elapsed_ms()); step += 1
for on, name in ((0, "index order"), (1, "longest first (previous step's durations)")):
    v = np.array(t[on])
    print("%s fp%d x%d  %-45s mean %.3f ms  (min %.3f, max %.3f)" % (env_id, prec, n, name, v.mean(), v.min(), v.max()))
