cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch; torch.cuda.init()
import numpy as np, time, dart_env_amd
for env_id, kw in [("DartWalker2d-v1", True), ("DartHalfCheetah-v1", False), ("DartHopper-v1", True)]:
    v = dart_env_amd.vector.make(env_id, 65536, noise="philox", all_bodies_collide=kw)
    v.reset(); a = np.random.RandomState(0).uniform(-1,1,(8, 65536, v.env.act_dim)).astype(np.float32)
    s = v.env._stepper
    d_a = torch.from_numpy(a).cuda()
    for w in (0, 1):
        for i in range(100): s.step_device(d_a[i % 8].data_ptr())
        s.sync(); t0=time.perf_counter()
        for i in range(50): s.step_device(d_a[i % 8].data_ptr())
        s.sync(); dt=(time.perf_counter()-t0)/50
        print(env_id, "spatial kernel: %.3f ms/step, %.3e env-steps/s" % (dt*1e3, 65536/dt))
    v.close()
PY
