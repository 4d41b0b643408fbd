#!/usr/bin/env python3
"""Half cheetah x 65 536: kernel time (HIP events, bench.time_config) per wave-vote setting K, for the library DART_STEPPER_LIB names (default:
the in-tree one).  K = 0: the big register tier for every lane beyond the small one; K = 64: every such lane to the wave solvers
(wave_constraints4: four envs per pass; wave_constraints beyond 16 rows).  python tools/gpu/cheetah_coop4_probe.py [tag] [K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dart_env_amd import stepper as st

tag = sys.argv[1] if len(sys.argv) > 1 else "base"
ks = [int(a) for a in sys.argv[2:]] or [0, 3, 64]
for prec in (64, 32):
    for k in ks:
        cfg = [(st.CFG_WAVE_VOTE, k)] if k >= 0 else []
        ms = [bench.time_config("DartHalfCheetah-v1", 65536, 0, prec, 100, 300, configure=cfg)[0] for _ in range(2)]
        print("%-8s DartHalfCheetah-v1 f%d  K=%-3d  %.4f / %.4f ms" % (tag, prec, k, ms[0], ms[1]), flush=True)
