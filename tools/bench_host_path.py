#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (dart_step with numpy arrays, as the gym.vector surface uses it).
Not the headline number (bench.py's `value` is HBM-resident by contract); recorded in DESIGN.md."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "ab":
    # A/B of the host path in two fresh processes on the same box: round 2's (four D2H copies per step, fresh output arrays every
    # step) against the current one (one packed D2H, output arrays reused once the caller has dropped them)
    for tag, env in (("legacy (DART_CFG_HOST_DMA=4: four copies; DART_NO_OUT_POOL=1)", {"BENCH_HOST_DMA": "4", "DART_NO_OUT_POOL": "1"}), ("current", {})):
        print("==", tag, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[2:], env=dict(os.environ, **env), check=True)
    sys.exit(0)
import numpy as np
import dart_env_amd
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
card = card_for("DartHopper-v1")
s = st.HipStepper(card, n)
s.configure(st.CFG_AUTORESET, 1)
s.output_pool = os.environ.get("DART_NO_OUT_POOL") != "1"
if os.environ.get("BENCH_HOST_DMA"):          # (this tool's own switch: the library reads no environment variables)
    s.configure(st.CFG_HOST_DMA, int(os.environ["BENCH_HOST_DMA"]))
s.reset(None, None, None)
a = np.random.RandomState(0).uniform(-1, 1, (n, 3)).astype(np.float32)
for _ in range(20):
    s.step(a)
t0 = time.perf_counter(); K = 200
for _ in range(K):
    s.step(a)
dt = time.perf_counter() - t0
print("host-buffer dart_step (H2D actions + kernel + D2H obs/reward/done, philox auto-reset): %.1f us/step, %.3e env-steps/s"
      % (dt / K * 1e6, n * K / dt))
venv = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox")
venv.reset()
for _ in range(5):
    venv.step(a)
t0 = time.perf_counter(); K = 100
for _ in range(K):
    venv.step(a)
dt = time.perf_counter() - t0
print("DartVectorEnv.step (python surface, philox): %.1f us/step, %.3e env-steps/s" % (dt / K * 1e6, n * K / dt))
venvz = dart_env_amd.vector.make("DartHopper-v1", n, noise="philox", copy=False)   # observations = views of the pinned buffer
venvz.reset()
for _ in range(5):
    venvz.step(a)
t0 = time.perf_counter(); K = 100
for _ in range(K):
    venvz.step(a)
dt = time.perf_counter() - t0
print("DartVectorEnv.step (python surface, philox, copy=False): %.1f us/step, %.3e env-steps/s" % (dt / K * 1e6, n * K / dt))
t0 = time.perf_counter()
venv2 = dart_env_amd.vector.make("DartHopper-v1", n)   # default: reference-exact MT19937 reset noise, bank in HBM
venv2.seed(0); venv2.reset()
print("seed(0) + reset() of %d MT19937 envs: %.2f s" % (n, time.perf_counter() - t0))
for _ in range(5):
    venv2.step(a)
t0 = time.perf_counter(); K = 100
for _ in range(K):
    venv2.step(a)
dt = time.perf_counter() - t0
print("DartVectorEnv.step (python surface, device mt19937): %.1f us/step, %.3e env-steps/s" % (dt / K * 1e6, n * K / dt))
venv3 = dart_env_amd.vector.make("DartHopper-v1", 4096, noise="mt19937-host")   # numpy draws on the host
venv3.seed(0); venv3.reset()
a2 = a[:4096]
for _ in range(3):
    venv3.step(a2)
t0 = time.perf_counter(); K = 20
for _ in range(K):
    venv3.step(a2)
dt = time.perf_counter() - t0
print("DartVectorEnv.step (mt19937-host noise, 4096 envs): %.1f us/step, %.3e env-steps/s" % (dt / K * 1e6, 4096 * K / dt))
