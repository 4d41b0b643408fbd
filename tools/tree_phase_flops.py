"""Where do the tree kernel's flops go?  The counting-scalar build of the tree kernel on the fiber runtime (tests/kernel_emu/emu_tree_flops.cpp)
with the kernel's own phase stopwatch (SP_TICK) reading the running flop total: lane-flops per env-step by phase of the world step.
    python tools/tree_phase_flops.py [path to libdart_tree_flops.so]"""
import ctypes as C, numpy as np, sys
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dart_env_amd.model_card import DartModelCard, card_for
if len(sys.argv) < 2:
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "kernel_emu"), "libdart_tree_flops.so"])
L = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "kernel_emu", "libdart_tree_flops.so"))
L.flops_create.restype = C.c_void_p
L.flops_create.argtypes = [C.POINTER(DartModelCard), C.c_int64, C.c_int, C.c_char_p, C.c_int]
L.flops_destroy.argtypes = [C.c_void_p]
L.flops_enable_stats.argtypes = [C.c_void_p, C.c_int]; L.flops_get_stats.argtypes = [C.c_void_p, C.c_void_p]
L.flops_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
L.flops_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64, C.c_uint64]
names = ["kinematics + link dynamics", "mass-matrix rows", "factorisations + forward dynamics", "contact detection / rows", "Jacobian rows", "W = L^-1 J^T", "A = W W^T", "pivoting, frictionless stage", "pivoting, friction stage", "velocity update"]
for env_id, n, warm, steps in (("DartHumanWalker-v1", 8, 15, 20), ("DartWalker3d-v1", 8, 15, 20)):
    card = card_for(env_id)
    why = C.create_string_buffer(256)
    h = L.flops_create(C.byref(card), n, 1, why, 256)
    obs = np.zeros((n, card.obs_dim), np.float32); rew = np.zeros(n, np.float32); done = np.zeros(n, np.uint8); trunc = np.zeros(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.flops_reset(h, p(obs), 0, 0)
    rng = np.random.RandomState(0)
    for t in range(warm + steps):
        if t == warm: L.flops_enable_stats(h, 1)
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        L.flops_step(h, p(a), p(obs), p(rew), p(done), p(trunc), 0, 0)
    st = np.zeros(64, np.uint64); L.flops_get_stats(h, p(st))
    ph = st[40:50].astype(float); tot = ph.sum()
    print(env_id, "flops per env-step inside the world steps: %.0f" % (tot / (n * steps)))
    for nm, v in zip(names, ph): print("   %-36s %5.1f %%   %9.0f per env-step" % (nm, 100 * v / tot, v / (n * steps)))
    L.flops_destroy(h)
