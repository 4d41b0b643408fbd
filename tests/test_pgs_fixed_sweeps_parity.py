"""Fixed-sweep PGS (`north_star`'s solver; DART_CFG_SOLVER = 1) of the DEVICE CODE against the oracle's PGS at the same sweep count --
without a GPU: the lane kernels' host build and the tree kernel on the fiber runtime (tests/kernel_emu).  The GPU run of the same
protocol is tests/test_gpu_pgs_parity.py.  Protocol and rationale: tests/pgs_protocol.py."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
from tests.pgs_protocol import exact_vs_pgs_gap, pgs_rollout

LANE = lambda card, n: EmuStepper(card, n, precision=64)
TREE = lambda card, n: EmuStepper(card, n, precision=64, tree=True)


@pytest.mark.parametrize("K", [30, 80])
@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_lane_kernel_pgs_equals_oracle_pgs_at_the_same_sweep_count(env_id, K):
    card = card_for(env_id)
    r = pgs_rollout(LANE, card, 64, 20, K)
    assert r["done_mismatches"] == 0
    assert r["dq"][0] < 1e-9 and max(r["dq"]) < 1e-7 and max(r["q"]) < 1e-9, (r["dq"][0], max(r["dq"]), max(r["q"]))


def test_one_world_step_of_lane_kernel_pgs():
    """frame_skip = 1: a single world step per launch, K = 30"""
    card = card_for("DartHopper-v1")
    card.frame_skip = 1
    r = pgs_rollout(LANE, card, 64, 40, 30)
    assert r["done_mismatches"] == 0 and max(r["dq"]) < 1e-9, max(r["dq"])


CASES = [("DartHumanWalker-v1", {}, 2, 5, 30),                       # prefix row order (normals / limits / tangents), register-LCP model
         ("DartHumanWalker-v1", {}, 2, 3, 80),
         ("DartWalker3d-v1", {}, 2, 4, 30),                          # link-link contacts: interleaved {n, t1, t2} rows
         ("DartHopper-v1", {"generic_kernel": True}, 4, 12, 30),     # planar model on the tree kernel (z tangents pinned)
         ("DartReacher-v1", {"generic_kernel": True}, 3, 10, 30)]    # Coulomb joint-friction rows


@pytest.mark.parametrize("env_id,kw,n,T,K", CASES, ids=["%s-K%d%s" % (c[0], c[4], "/generic" if c[1] else "") for c in CASES])
def test_tree_kernel_pgs_equals_oracle_pgs_at_the_same_sweep_count(env_id, kw, n, T, K):
    """The tree kernel stores its LCP rows in prefix order (normals, limits, joint friction, tangents) but sweeps them in the oracle's
    order ({n, t1, t2} per contact first) -- round 4; before, its K-sweep iterate differed from the oracle's by 1e-4 after one env-step."""
    card = card_for(env_id, **kw)
    r = pgs_rollout(TREE, card, n, T, K)
    assert r["done_mismatches"] == 0
    assert r["dq"][0] < 1e-9 and max(r["dq"]) < 1e-7 and max(r["q"]) < 1e-9, (r["dq"][0], max(r["dq"]), max(r["q"]))


def test_thirty_sweeps_are_far_from_the_exact_solve():
    """the parity checks above compare ITERATES: 30 sweeps leave a velocity error of >= 1e-4 against the exact solve on the oracle itself,
    five orders of magnitude above the kernel-vs-oracle tolerance"""
    assert exact_vs_pgs_gap(card_for("DartHopper-v1"), 32, 20, 30) > 1e-4
