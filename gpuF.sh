cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x -k "all_capsule" -s 2>&1 | tail -5
python - <<'PY'
import torch; torch.cuda.init()
import numpy as np, time, dart_env_amd
v = dart_env_amd.vector.make("DartHopper-v1", 65536, noise="philox", all_bodies_collide=True)
v.reset(); a = np.random.RandomState(0).uniform(-1,1,(8, 65536, 3)).astype(np.float32)
s = v.env._stepper
d_a = torch.from_numpy(a).cuda()
for i in range(100): s.step_device(d_a[i % 8].data_ptr())
s.sync(); t0=time.perf_counter()
for i in range(200): s.step_device(d_a[i % 8].data_ptr())
s.sync(); dt=(time.perf_counter()-t0)/200
print("Hopper all-capsule (planar kernel, 11 rows): %.1f us/step, %.3e env-steps/s, LCP slots %d" % (dt*1e6, 65536/dt, s.query(7)))
PY
