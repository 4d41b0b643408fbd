"""Why does a loop that keeps the previous step's arrays alive run slower than one that drops them?  (host surface of DartVectorEnv)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dart_env_amd
if len(sys.argv) > 1 and sys.argv[1] == "torch":      # bench.py's process has torch's HIP context alive next to the stepper's
    import torch
    torch.cuda.set_device(0); x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
    print("torch context up, threads", torch.get_num_threads())
n = 65536
a = np.random.RandomState(0).uniform(-1, 1, (n, 3)).astype(np.float32)
for mode in ("discard", "bind", "discard"):
    venv = dart_env_amd.vector.make("DartHopper-v1", n)
    venv.seed(0); venv.reset()
    for _ in range(10):
        venv.step(a)
    t0 = time.perf_counter(); K = 200
    if mode == "discard":
        for _ in range(K):
            venv.step(a)
    else:
        for _ in range(K):
            obs, rew, done, info = venv.step(a)
    dt = time.perf_counter() - t0
    st = venv.env._stepper
    print(mode, "%.1f us/step" % (dt / K * 1e6), "blocks", len(st.__dict__.get("_blocks", [])))
    # where does the time go in one step
    t = {}
    for _ in range(50):
        t0 = time.perf_counter(); venv.step_async(a); t1 = time.perf_counter(); r = venv.step_wait(); t2 = time.perf_counter()
        t["async"] = t.get("async", 0) + t1 - t0; t["wait"] = t.get("wait", 0) + t2 - t1
        if mode == "bind":
            keep = r
    print("   ", {k: "%.1f us" % (v / 50 * 1e6) for k, v in t.items()})
    venv.close()
