"""Where the construction time of a 65 536-env vector env goes (GPU box diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dart_env_amd
from dart_env_amd import stepper as st, seeding
from dart_env_amd.model_card import card_for
n = 65536
t = time.perf_counter(); s = st.HipStepper(card_for("DartHopper-v1"), n); print("HipStepper create  %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); used = [seeding.create_seed(i) for i in range(n)]; keys, klen = seeding.mt_keys(used); print("host keys          %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); s.seed_mt19937(keys, klen); print("dart_seed_mt19937  %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); s.seed_mt19937(keys, klen); print("dart_seed_mt19937 again %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); s.reset(None, None, None); print("dart_reset         %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); v = dart_env_amd.vector.make("DartHopper-v1", n); print("vector.make        %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); v.seed(0); print("venv.seed(0)       %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); v.reset(); print("venv.reset()       %.3f s" % (time.perf_counter() - t))
