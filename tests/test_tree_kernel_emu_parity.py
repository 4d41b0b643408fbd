"""The TREE kernel (dart_env_amd/csrc/spatial_*.hpp + wave_blcp.hpp: one env per 64-lane wavefront, lanes cooperating through LDS,
barriers, ballots, shuffles, v_readlane and DPP) held against the fp64 oracle WITHOUT a GPU: the device source is compiled by g++
against tests/kernel_emu/fake_wave_include/hip/hip_runtime.h, where every lane of a workgroup is a fiber and every cross-lane operation
a rendezvous with the hardware's semantics.  This is the kernel of BASELINE config 4 (DartHumanWalker-v1) and of every model the lane
kernels do not take; its GPU parity tests are tests/test_gpu_spatial.py and tests/test_gpu_long_parity.py."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for
from tests.batch_oracle import OracleBatch
from tests.emu_lib import EmuStepper

CASES = [   # env id, card options, envs, env-steps: every instantiation of the step kernel the library launches
    ("DartHumanWalker-v1", {}, 3, 14),                       # big model, compile-time factor pattern, register LCP solver
    ("DartHumanWalker-v1", {"generic_kernel": True}, 2, 6),  # the same model on the general (EXTRAS) instantiation
    ("DartWalker3d-v1", {}, 3, 12),                          # link-link box contacts (PAIRS), register solver
    ("DartWalker3dSPD-v1", {}, 2, 8),                        # stable-PD controller: second factorisation, carried constraint forces
    ("DartDog-v1", {}, 3, 12),                               # free root joint, LDS solver
    ("DartHopper-v1", {"generic_kernel": True}, 4, 20),      # small models on the tree kernel
    ("DartWalker2d-v1", {"generic_kernel": True}, 3, 12),
    ("DartHalfCheetah-v1", {"generic_kernel": True}, 3, 12), # welds, joint springs
    ("DartSnake7Link-v1", {"generic_kernel": True}, 3, 12),  # fluid forces
    ("DartCartPole-v1", {"generic_kernel": True}, 4, 30),
    ("DartReacher-v1", {"generic_kernel": True}, 3, 12),
    ("DartReacher3d-v1", {"generic_kernel": True}, 3, 8),
]


@pytest.mark.parametrize("env_id,kw,n,T", CASES, ids=[c[0] + ("/generic" if c[1] else "") for c in CASES])
def test_tree_kernel_matches_oracle_on_the_host(env_id, kw, n, T):
    card = card_for(env_id, **kw)
    g = EmuStepper(card, n, precision=64, tree=True)
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(1)
    qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
    og = g.reset(None, qn, vn); ora.reset(None, qn, vn)
    assert np.abs(og - ora.obs()).max() < 1e-6
    resets = 0
    for t in range(T):
        a = (rng.uniform(-1, 1, (n, card.act_dim)) * (2.5 if t % 5 == 4 else 1.0)).astype(np.float32)     # some actions beyond the clamp
        o, r, d, tr = g.step(a)
        oo, ro, do, to = ora.step(a)
        qg, dqg = g.get_state(); qo, dqo = ora.state()
        assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-7, (t, np.abs(qg - qo).max(), np.abs(dqg - dqo).max())
        assert np.array_equal(d.astype(bool), np.asarray(do, bool)) and np.abs(o - oo).max() < 2e-5 and np.abs(r - ro).max() < 1e-4, t
        if np.any(do):
            resets += int(np.sum(do))
            qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
            g.reset(np.asarray(do, np.uint8), qn, vn, want_obs=False); ora.reset(np.asarray(do, bool), qn, vn)
    g.close()


def test_tree_kernel_fp32_and_contact_report_on_the_host():
    """the fp32 instantiation stays near the oracle step by step (re-synchronised every step), and the reporting instantiation
    (dart_get_contacts) returns the oracle's contacts: bodies, points, forces"""
    card = card_for("DartHumanWalker-v1")
    n = 2
    g32 = EmuStepper(card, n, precision=32, tree=True)
    g64 = EmuStepper(card, n, precision=64, tree=True)
    g64.enable_contact_report(True)
    ora = OracleBatch(card, n)
    rng = np.random.RandomState(2)
    qn = rng.uniform(-.005, .005, (n, card.ndofs)); vn = rng.uniform(-.005, .005, (n, card.ndofs))
    for g in (g32, g64):
        g.reset(None, qn, vn)
    ora.reset(None, qn, vn)
    seen = 0
    for t in range(8):
        a = rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)
        g32.step(a); g64.step(a); ora.step(a)
        qo, dqo = ora.state()
        q32, dq32 = g32.get_state()
        assert np.abs(q32 - qo).max() < 2e-3 and np.percentile(np.abs(dq32 - dqo), 90) < 5e-2, t
        g32.set_state(qo, dqo)
        cnt, bod, pt, fc = g64.contacts()
        for i, w in enumerate(ora.worlds):
            rep = w.contact_report()
            assert cnt[i] == len(rep)
            if len(rep):
                seen += 1
                assert np.array_equal(bod[i, :len(rep)], rep[:, :2].astype(np.int32))
                assert np.allclose(pt[i, :len(rep)], rep[:, 2:5], atol=1e-9) and np.allclose(fc[i, :len(rep)], rep[:, 5:8], atol=1e-5, rtol=1e-6)
    assert seen >= 8
    g32.close(); g64.close()


def test_lds_block_of_the_tree_kernel_decides_the_workgroups_per_cu():
    """Round 4: the tree kernel runs one env per workgroup and a CU holds floor(160 KB / LDS block) of them whatever the registers allow --
    HumanWalker's fp64 time went 9.7 -> 9.0 -> 6.95 ms as its block went 27 640 -> 26 928 -> 22 776 B (5 -> 6 -> 7 workgroups per CU;
    DESIGN.md section 4.2).  The thresholds are pinned here so that a field added to the block cannot silently cost a workgroup per CU:
    this is DART_Q_LDS_BYTES of the product library (same host code), read through the emulator build."""
    cu = 160 * 1024
    sizes = {}
    for env_id, prec in (("DartHumanWalker-v1", 64), ("DartHumanWalker-v1", 32), ("DartWalker3d-v1", 64), ("DartDog-v1", 64)):
        e = EmuStepper(card_for(env_id), 1, precision=prec, tree=True)
        sizes[(env_id, prec)] = e.lds_bytes
        e.close()
    assert cu // sizes[("DartHumanWalker-v1", 64)] >= 8, sizes       # round 5: the factor's skyline storage -> 19 952 B = two waves on every SIMD
    assert cu // sizes[("DartHumanWalker-v1", 32)] >= 12, sizes      # 3 waves per SIMD x 4 SIMDs
    assert cu // sizes[("DartWalker3d-v1", 64)] >= 3, sizes
    assert cu // sizes[("DartDog-v1", 64)] >= 6, sizes
    # the pose / dynamics split of the link records: the block no longer grows with 37 Reals per link
    assert sizes[("DartHumanWalker-v1", 64)] <= 20416 and sizes[("DartHumanWalker-v1", 32)] <= 11904, sizes


_POISON_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
outs = []
for env_id, prec in (("DartHumanWalker-v1", 64), ("DartHumanWalker-v1", 32), ("DartWalker3d-v1", 64), ("DartDog-v1", 64)):
    card = card_for(env_id); n = 2
    g = EmuStepper(card, n, precision=prec, tree=True)
    g.reset()
    rng = np.random.RandomState(3)
    for t in range(4):
        ob, r, d, tr = g.step(rng.uniform(-1, 1, (n, card.act_dim)).astype(np.float32)); outs += [np.asarray(ob, np.float64).ravel(), np.asarray(r, np.float64).ravel()]
    outs += [np.asarray(x, np.float64).ravel() for x in g.get_state()]
    g.close()
np.save(sys.argv[2], np.concatenate(outs))
"""


def test_tree_kernel_results_do_not_depend_on_what_lds_held_before(tmp_path):
    """Round 4 re-laid the tree kernel's LDS block (arrays dropped for the lean kernels, link records split, the dynamics records and the
    solver's matrix living in blocks that other arrays own at other times).  On the device a workgroup starts with whatever the previous one
    left in LDS: the emulator fills the block with a byte pattern before every workgroup (DART_EMU_POISON_LDS, fake_wave_include) -- NaNs /
    -1 with 0xff, plausible finite numbers with 0x3f -- and the results must be bit for bit those of a zeroed block."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "poison.py"
    script.write_text(_POISON_SCRIPT)
    got = {}
    for tag, val in (("zero", "0x00"), ("nan", "0xff"), ("finite", "0x3f")):
        out = tmp_path / (tag + ".npy")
        env = dict(os.environ, DART_EMU_POISON_LDS=val)
        subprocess.check_call([sys.executable, str(script), root, str(out)], env=env, timeout=900)
        got[tag] = np.load(out)
    assert np.all(np.isfinite(got["zero"]))
    assert np.array_equal(got["zero"], got["nan"]) and np.array_equal(got["zero"], got["finite"])


def test_lds_reservation_covers_what_the_kernels_carve():
    """sp_lds_bytes (what the host reserves per workgroup) against sp_carve (what the kernels lay out in it) over a sweep of model dimensions,
    both precisions, with and without the register-LCP trim: the carve must end inside the reservation -- and not far inside (a reservation
    that grows unnoticed costs workgroups per CU, the round-4 lever)."""
    import ctypes as C
    from tests import emu_lib
    L = emu_lib.lib(tree=True)
    for f in (L.emu_lds_reserved, L.emu_lds_carved):
        f.argtypes = [C.c_int] * 7; f.restype = C.c_longlong
    sky = L.emu_pattern_hreals()          # the HumanWalker pattern's skyline H block (round 5)
    assert 0 < sky < 576
    rng = np.random.RandomState(0)
    cases = [(33, 29, 36, 12), (22, 22, 36, 12), (21, 21, 64, 20), (4, 6, 36, 12), (1, 1, 36, 12), (64, 32, 64, 20)]
    cases += [(int(nl), int(min(nl, rng.randint(1, 33))), int(rng.choice([36, 64])), int(rng.choice([12, 20]))) for nl in rng.randint(1, 65, 60)]
    for nl, n, maxm, maxcp in cases:
        for rb in (4, 8):
            for reg in (0, 1):
                for hreals in ((0, sky) if (nl, n) == (33, 29) else (0,)):
                    res, car = L.emu_lds_reserved(nl, n, rb, maxm, maxcp, reg, hreals), L.emu_lds_carved(nl, n, rb, maxm, maxcp, reg, hreals)
                    assert car <= res, (nl, n, maxm, maxcp, rb, reg, hreals, car, res)
                    assert res - car <= 64 + 8 * rb, (nl, n, maxm, maxcp, rb, reg, hreals, car, res)
