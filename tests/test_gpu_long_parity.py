"""North-star parity gate: RMS state divergence vs the CPU oracle < 1e-4 over 1 000 env-steps -- UNTRIMMED.

Protocol = SURVEY.md 8(d) / tests/parity_protocol.py: >= 4 096 envs, random actions, same Philox reset streams, resets follow the
oracle's done flags (every env is compared inside the same episode), RMS over ALL envs x dofs of the pre-reset state
difference after 1, 10, 100 and 1 000 env-steps; nothing is trimmed, no percentile.  The mode that claims the bound is the
product default, `precision=64` (the kernels' fp64 instantiation; DESIGN.md section 6 explains why fp32 arithmetic cannot: one
contact / limit event taken a substep early anywhere in 4 M env-steps already costs more than 1e-4).  The fp32 fast mode is held
to what it does deliver by tests/test_gpu_parity.py.

Reference call sites: hopper.py:60-62 (done), dart_env.py:170-175 (do_simulation), walker2d.py:22-65, human_walker.py:60-165.
"""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for
from tests.parity_protocol import parity_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,ne", [("DartHopper-v1", 4096), ("DartWalker2d-v1", 4096), ("DartHumanWalker-v1", 4096)])
def test_fp64_untrimmed_rms_below_1e_4_over_1000_steps(env_id, ne):
    stats, ref, _ = parity_check(env_id, 64, ne, 1000, 0)
    print(env_id, {k: (v["q"], v["dq"]) for k, v in stats["by_step"].items()}, "episodes", stats["episodes"],
          "oracle %.1fs on %d threads, gpu %.1fs" % (ref["seconds"], ref["threads"], stats["stepper_seconds"]))
    assert stats["envs"] >= 4096 and stats["env_steps"] >= 1000 and set(stats["by_step"]) == {"1", "10", "100", "1000"}
    assert stats["done_flag_mismatches"] == 0
    assert all(v["envs_non_finite"] == 0 for v in stats["by_step"].values())
    # envs the reference's own validity bound calls broken are left out of the RMS only where BOTH sides say so
    assert all(v["envs_broken_one_side"] == 0 for v in stats["by_step"].values())
    assert max(v["envs_broken_sim"] for v in stats["by_step"].values()) <= 0.01 * ne
    assert stats["q"] < 1e-4 and stats["dq"] < 1e-4, (stats["q"], stats["dq"])
    # and with a wide margin: the fp64 kernels follow the oracle to rounding error
    assert stats["q"] < 1e-7 and stats["dq"] < 1e-6, (stats["q"], stats["dq"])


@pytest.mark.parametrize("env_id,ne,steps", [("DartHopper-v1", 1024, 300), ("DartWalker2d-v1", 1024, 300), ("DartHumanWalker-v1", 256, 100),
                                             ("DartReacher-v1", 1024, 100), ("DartReacher3d-v1", 256, 100), ("DartCartPole-v1", 1024, 100)])
def test_the_other_setting_of_the_impulse_inertia_knob_is_served_too(env_id, ne, steps):
    """card.impulse_inertia = 1 = DART_IMPULSE_AUGMENTED (A3: the impulse pass on M + dt D + dt^2 K, what rounds 1-2 of this build assumed) on every kernel
    family -- planar register kernel (runtime-parameter variant: the baked kernels carry DART 6's setting), tree kernel, arm, 3-D
    chain and cart kernels -- against the oracle with the same setting, same untrimmed protocol."""
    stats, ref, _ = parity_check(env_id, 64, ne, steps, 0, impulse_inertia=1)
    print(env_id, "knob 0", {k: (v["q"], v["dq"]) for k, v in stats["by_step"].items()}, "episodes", stats["episodes"])
    assert stats["done_flag_mismatches"] == 0
    assert stats["q"] < 1e-7 and stats["dq"] < 1e-6, (stats["q"], stats["dq"])


@pytest.mark.parametrize("env_id,steps", [("DartHalfCheetah-v1", 100), ("DartSnake7Link-v1", 1000), ("DartCartPole-v1", 1000),
                                          ("DartDoubleInvertedPendulumEnv-v1", 1000), ("DartReacher-v1", 1000)])
def test_lane_kernels_added_in_round_2_meet_the_same_bound(env_id, steps):
    """The same untrimmed protocol for the tasks that moved from the tree kernel to lane-per-env kernels in round 2.  The half
    cheetah is compared over 100 env-steps: its episodes only end with the 1 000-step TimeLimit, and a cheetah thrashing under
    random torques is chaotic -- two fp64 evaluation orders of the same equations drift apart by ~10^4 per 90 env-steps (measured:
    RMS q 3e-17 after 1 step, 2e-16 after 10, 3e-12 after 100, O(0.1) after 1 000; two copies of the ORACLE started 1e-15 apart do
    the same: 1e-11 after 100 env-steps, 7e-7 after 200, 0.09 after 400); the configs of the north star reset long before that."""
    stats, ref, _ = parity_check(env_id, 64, 4096, steps, 0)
    print(env_id, {k: (v["q"], v["dq"]) for k, v in stats["by_step"].items()}, "episodes", stats["episodes"],
          "oracle %.1fs on %d threads, gpu %.1fs" % (ref["seconds"], ref["threads"], stats["stepper_seconds"]))
    assert stats["envs"] >= 4096 and stats["env_steps"] >= steps and stats["done_flag_mismatches"] == 0
    assert all(v["envs_non_finite"] == 0 for v in stats["by_step"].values())
    assert stats["q"] < 1e-7 and stats["dq"] < 1e-6, (stats["q"], stats["dq"])


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_fp32_fast_mode_divergence_is_a_few_flipped_events(env_id):
    """What fp32 delivers under the same protocol: almost every env within 1e-4 of the oracle at every checkpoint; the
    untrimmed RMS is owned by the few envs whose contact / limit switched a substep early (counted here)."""
    stats, _, _ = parity_check(env_id, 32, 4096, 1000, 0)
    worst = max(v["envs_beyond_1e-4"] for v in stats["by_step"].values())
    print(env_id, "fp32 untrimmed rms q %.2e dq %.2e, envs beyond 1e-4 (worst checkpoint) %d / 4096, done mismatches %d / %d"
          % (stats["q"], stats["dq"], worst, stats["done_flag_mismatches"], 4096 * 1000))
    assert worst <= 0.02 * 4096
    assert stats["done_flag_mismatches"] <= 2e-3 * 4096 * 1000
