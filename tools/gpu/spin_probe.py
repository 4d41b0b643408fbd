"""What has to sit between the device spin-up and bench.py's timed window (steps 5..25 of a batch's episodes)?  us per batched step of that
window with the scratch batch released before it / kept alive / the spin-up on the measured batch itself / none / 20 ms of idle in between.
    python tools/gpu/spin_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
env_id = "DartHopper-v1"; n = 65536
def fresh(seed): return bench.HipBenchEnv(env_id, n, 0, 64, 0, ring=16, ring_seed=seed, all_bodies_collide=None, configure=[])
def window(b):
    b.reset(); b.mark(0); b.run(5); b.sync()
    b.mark(0); b.run(20, 5); b.mark(1); b.sync()
    return b.elapsed_ms() / 20 * 1e3
res = {}
for rep in range(3):
    # (a) scratch spin, close, then measure
    b = fresh(1234); sb = fresh(99); sb.reset(); sb.run(2000); sb.sync(); sb.close(); res.setdefault("a close-before", []).append(window(b)); b.close()
    # (b) scratch spin, keep alive
    b = fresh(1234); sb = fresh(99); sb.reset(); sb.run(2000); sb.sync(); res.setdefault("b keep-alive", []).append(window(b)); sb.close(); b.close()
    # (d) spin on the measured batch itself, then reset
    b = fresh(1234); b.reset(); b.run(2000); b.sync(); res.setdefault("d same-batch", []).append(window(b)); b.close()
    # (e) no spin
    b = fresh(1234); res.setdefault("e none", []).append(window(b)); b.close()
    # (f) spin, then sleep 20 ms idle, then measure
    b = fresh(1234); sb = fresh(99); sb.reset(); sb.run(2000); sb.sync(); time.sleep(0.02); res.setdefault("f spin+20ms idle", []).append(window(b)); sb.close(); b.close()
for k, v in res.items(): print(k, ["%.2f" % x for x in v])
