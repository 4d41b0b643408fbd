#!/usr/bin/env python3
"""Worst cases of the half cheetah's wave solvers (round 5): 65 536 envs dropped onto the floor at a given root height, zero actions, no auto-reset --
every env rests on 3 ... 8 capsules, so every lane of every wave goes through wave_constraints4 (<= 16 rows: 16 passes per world step) or the
one-at-a-time wave_constraints (> 16 rows).  Prints ms per batched step, the share of finite states and the touching-capsule histogram of a sample
(oracle-free: counts come from the contact report).  python tools/gpu/cheetah_floor_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for

card = card_for("DartHalfCheetah-v1")
n, nd, na = 65536, card.ndofs, card.act_dim
for prec in (64, 32):
    for height, pitch in ((-0.25, 0.0), (-0.45, 0.0), (-0.45, 1.4)):
        s = st.HipStepper(card, n, precision=prec)
        s.configure(st.CFG_AUTORESET, 0)
        s.configure(st.CFG_CONTACT_REPORT, 1)
        rng = np.random.RandomState(1)
        q = rng.uniform(-0.05, 0.05, (n, nd)); dq = np.zeros((n, nd))
        q[:, 1] += height; q[:, 2] += pitch
        s.set_state(q, dq)
        a = np.zeros((n, na), np.float32)
        for _ in range(3):
            s.step(a)
        s.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            s.step(a)
        s.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        cnt = s.contacts()[0][:4096]
        qq, _ = s.get_state()
        print("f%d root height %+.2f pitch %.1f: %.2f ms per host-synchronous step; finite %.4f; touching capsules (4 096 envs): %s" %
              (prec, height, pitch, ms, np.isfinite(qq).all(axis=1).mean(), np.bincount(cnt, minlength=9).tolist()), flush=True)
        s.close()
