#!/usr/bin/env python3
"""Exact floating-point operation counts of the lane kernels, per env-step and per model -> profiles/flops_per_env_step.json.

The planar / cart / arm / 3-D-chain kernels are one template on their scalar type; tests/kernel_emu/emu_flops.cpp instantiates
that same device source on the host with a COUNTING scalar (every +, -, *, fma, /, sqrt, v_rcp, v_rsq on a `Real` increments a
counter) and this script drives it with the workload bench.py times: random actions U[-1, 1), on-device (Philox) auto-reset,
after a warm-up that de-synchronises the episodes.  flops = add + mul + 2 fma + div + sqrt + rcp + rsq, per LANE = per env.

What the number is, and is not:
  * it is the arithmetic one environment's lane performs (pivoting loops end on the lane's own convergence);
  * the GPU executes more: a wavefront iterates until its slowest lane is done, masked LDL^T rows are computed and discarded --
    bench.py's `roofline.valu.frac` = (these flops x env-steps/s) / vector peak is therefore the USEFUL fraction of the VALU peak,
    next to `roofline.valu_issue`, the measured issue statistics of everything the wave executes;
  * the counting type takes the fp64 code paths (sincos polynomial, two Newton steps per reciprocal); the fp32 instantiation
    does ~3 % less per env-step (shorter polynomials, one Newton step) -- both lines use this count.
The tree kernel (HumanWalker, Walker3d, Dog) is wave-cooperative code: it is counted on the fiber runtime (emu_tree_flops.cpp), summed
over the lanes of an env's wavefront -- LANE-REPLICATED work included (a row per lane means every lane carries every column update of a
factorisation); `useful_factorisation_flops_per_world_step` is the arithmetic a sequential sparse Cholesky of the same matrices
needs, from the factor's pattern, beside it.  The PMC pass gives the lane-instructions the hardware executed
(tools/update_pmc_traffic.py: SQ_THREAD_CYCLES_VALU lane-instructions x 2 flops as an upper bound): counted flops <= that bound.
Every entry is stamped with the hash of the kernel sources it was counted on (tools/source_hash.py); bench.py ignores stale entries.

Run here (CPU only):  python tools/count_flops.py
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dart_env_amd.model_card import DartModelCard, card_for  # noqa: E402
from tools.source_hash import family_hash  # noqa: E402
from tools.gen_tree_patterns import factor_pattern  # noqa: E402


def useful_factorisation_flops(card):
    """Flops of ONE sequential sparse Cholesky of the model's mass matrix (leaves-first order: no fill-in) from the factor's pattern:
    per column j with c_j entries below the diagonal -- 1 rsqrt, c_j scalings, c_j (c_j + 1) / 2 multiply-adds (2 flops each)."""
    rows = factor_pattern(card)
    n = len(rows)
    flops = 0
    for j in range(n):
        c = sum(1 for i in range(j + 1, n) if (rows[i] >> j) & 1)
        flops += 1 + c + c * (c + 1)
    return flops, sum(bin(r).count("1") for r in rows)

EMU_DIR = os.path.join(ROOT, "tests", "kernel_emu")
FIELDS = ["add", "mul", "fma", "div", "sqrt", "rcp", "rsq", "cmp", "minmax", "abs", "neg", "cvt"]
ENVS = ["DartHopper-v1", "DartWalker2d-v1", "DartHalfCheetah-v1", "DartSnake7Link-v1", "DartCartPole-v1",
        "DartDoubleInvertedPendulumEnv-v1"]


# env id, envs, warm-up, counted steps (COUNT_TREE_ENVS / COUNT_TREE_STEPS override the sample: the fiber runtime takes ~0.3 s per
# HumanWalker env-step, so 64 envs x 100 steps is half an hour on one core)
_TN, _TS = int(os.environ.get("COUNT_TREE_ENVS", "64")), int(os.environ.get("COUNT_TREE_STEPS", "100"))
TREE_ENVS = [("DartHumanWalker-v1", _TN, 20, _TS), ("DartWalker3d-v1", max(8, _TN // 4), 20, 30), ("DartDog-v1", max(8, _TN // 4), 20, 30)]


def load(lib="libdart_planar_flops.so"):
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, lib])
    L = C.CDLL(os.path.join(EMU_DIR, lib))
    L.flops_create.restype = C.c_void_p
    L.flops_create.argtypes = [C.POINTER(DartModelCard), C.c_int64, C.c_int, C.c_char_p, C.c_int]
    L.flops_destroy.argtypes = [C.c_void_p]
    L.flops_is_static.argtypes = [C.c_void_p]
    L.flops_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.flops_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64, C.c_uint64]
    L.flops_read.argtypes = [C.c_void_p]
    return L


def count(L, env_id, n=256, warm=60, steps=200, seed=1234, **card_kw):
    card = card_for(env_id, **card_kw)
    why = C.create_string_buffer(256)
    h = L.flops_create(C.byref(card), n, 1, why, 256)
    if not h:
        raise RuntimeError("%s: %s" % (env_id, why.value.decode()))
    rng = np.random.RandomState(seed)
    obs = np.zeros((n, card.obs_dim), np.float32); rew = np.zeros(n, np.float32)
    done = np.zeros(n, np.uint8); trunc = np.zeros(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.flops_reset(h, p(obs), 0, 0)
    resets = 0
    for t in range(warm + steps):
        if t == warm:
            L.flops_clear()
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        L.flops_step(h, p(a), p(obs), p(rew), p(done), p(trunc), 0, 0)
        if t >= warm:
            resets += int(done.sum())
    raw = (C.c_ulonglong * 12)()
    L.flops_read(raw)
    static = bool(L.flops_is_static(h))
    L.flops_destroy(h)
    per = {k: raw[i] / float(n * steps) for i, k in enumerate(FIELDS)}
    flops = per["add"] + per["mul"] + 2 * per["fma"] + per["div"] + per["sqrt"] + per["rcp"] + per["rsq"]
    return {"flops_per_env_step": flops, "ops_per_env_step": per, "frame_skip": int(card.frame_skip),
            "flops_per_world_step": flops / card.frame_skip, "compile_time_model": static,
            "sample": "%d envs x %d env-steps after %d warm-up steps, random actions U[-1,1), Philox auto-reset (%.3f resets per env-step)"
                      % (n, steps, warm, resets / float(n * steps)),
            "other_valu_ops_per_env_step": per["cmp"] + per["minmax"] + per["abs"] + per["neg"] + per["cvt"]}


def main():
    L = load()
    out = {"_method": "tests/kernel_emu/emu_flops.cpp: the lane kernels' own source instantiated with a counting scalar; "
                      "flops = add + mul + 2 fma + div + sqrt + rcp + rsq per lane (= per env), fp64 code paths; tools/count_flops.py"}
    hp, hs = family_hash("planar"), family_hash("spatial")
    for env_id in ENVS:
        r = count(L, env_id)
        r["kernel_family"], r["source_hash"] = "planar", hp
        out[env_id] = r
        print("%-36s %9.0f flops / env-step (%6.0f / world step), + %6.0f other VALU ops; %s" %
              (env_id, r["flops_per_env_step"], r["flops_per_world_step"], r["other_valu_ops_per_env_step"], r["sample"]))
    r = count(L, "DartHopper-v1", all_bodies_collide=False)
    r["kernel_family"], r["source_hash"] = "planar", hp
    out["DartHopper-v1/feet_only"] = r
    print("%-36s %9.0f flops / env-step" % ("DartHopper-v1 feet only", r["flops_per_env_step"]))
    # the tree kernel (one env per wavefront): the same counting scalar on the fiber runtime (tests/kernel_emu/emu_tree_flops.cpp); the
    # count is the sum over the 64 lanes of what each lane executes for its env -- redundant per-lane work (every lane evaluating a
    # wave-uniform scalar) included, idle lanes not
    LT = load("libdart_tree_flops.so")
    for env_id, n, warm, steps in TREE_ENVS:
        r = count(LT, env_id, n=n, warm=warm, steps=steps)
        r["kernel"] = "tree kernel: summed over the 64 lanes of the env's wavefront"
        r["kernel_family"], r["source_hash"] = "spatial", hs
        card = card_for(env_id)
        uf, nnz = useful_factorisation_flops(card)
        # two factorisations per world step (M + E for the forward dynamics, M for the impulse pass)
        r["useful_factorisation_flops_per_world_step"] = 2 * uf
        r["factor_offdiagonal_nonzeros"] = nnz
        out[env_id] = r
        print("%-36s %9.0f flops / env-step (%6.0f / world step), + %6.0f other VALU ops; %s" %
              (env_id, r["flops_per_env_step"], r["flops_per_world_step"], r["other_valu_ops_per_env_step"], r["sample"]))
    path = os.path.join(ROOT, "profiles", "flops_per_env_step.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
