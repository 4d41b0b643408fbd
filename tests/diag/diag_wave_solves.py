"""Pivoting solves per LANE and per 64-lane WAVE, by stage and world step of the env-step (research tool, host build of the lane kernels).

A lane kernel's wave iterates the pivoting loop until its slowest lane has converged: what a start-set rule is worth is decided by the
maximum over 64 lanes, not by the lane mean.  Uses the tracing build of tests/diag/emu_trace.cpp (command in that file):
    TRLIB=/tmp/libdart_planar_emu_trace.so python tests/diag/diag_wave_solves.py DartWalker2d-v1
Round 5: this is the measurement behind WARM_FRICTION = false for the Walker2d topologies (planar_kernel.hpp, constraint_phase)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
os.environ["DART_EMU_LIB"]=""+os.environ.get("TRLIB","/tmp/libdart_planar_emu_trace.so")+""
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
env_id = sys.argv[1] if len(sys.argv) > 1 else "DartWalker2d-v1"
card = card_for(env_id)
n = int(os.environ.get("N","1024"))
g = EmuStepper(card, n, precision=64)
L = g.L
L.emu_trace_size.restype = C.c_int64
g.configure(st.CFG_AUTORESET, 1); g.configure(st.CFG_SEED, 0)
g.reset()
rng = np.random.RandomState(0)
for t in range(60):
    g.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32))
L.emu_trace_clear()
T=60
for t in range(T):
    g.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32))
k = L.emu_trace_size()
rec = np.zeros((k, 10), dtype=np.uint32)
L.emu_trace_get10(rec.ctypes.data_as(C.POINTER(C.c_uint32)))
print("records", k, "per env-step per env", k/(T*n))
its = rec[:,7].astype(int)+1
fs = card.frame_skip
for zb in (1,0):
    for f in range(fs):
        m = (rec[:,1]==zb)&(rec[:,9]==f)
        r = its[m]
        print("stage", 1 if zb else 2, "substep", f, "runs", m.sum(), "mean solves/lane %.3f" % r.mean(), "share>1: %.3f" % (r>1).mean(), "share>2: %.3f"%(r>2).mean())
# wave-level: group by (launch index inferred from order, wave = env//64, substep, stage): max over lanes
# records are ordered by launch (env loop inside) -> build launch id by detecting env index decreasing
env = rec[:,8].astype(int)
launch = np.cumsum(np.r_[0, (np.diff(env) < 0).astype(int)])
key = ((launch*64 + env//64)*fs + rec[:,9].astype(int))*2 + rec[:,1].astype(int)
order = np.argsort(key, kind='stable')
ks = key[order]; iv = its[order]
starts = np.r_[0, np.nonzero(np.diff(ks))[0]+1]
wmax = np.maximum.reduceat(iv, starts)
wkey = ks[starts]
for zb in (1,0):
    for f in range(fs):
        sel = ((wkey%2)==zb) & (((wkey//2)%fs)==f)
        print("WAVE stage", 1 if zb else 2, "substep", f, "waves", sel.sum(), "mean wave solves %.3f" % wmax[sel].mean())
print("WAVE total solves per env-step: %.2f ; lane mean total per env-step: %.2f" % (wmax.sum()/ (len(np.unique(launch))*(n//64)), its.sum()/(T*n)))
