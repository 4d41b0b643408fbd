"""The planar DEVICE kernels, compiled for the host (tests/emu_lib.py), against the fp64 oracle -- runs without a GPU.

This is the arithmetic of dart_env_amd/csrc/planar_kernel.hpp itself (world-aligned planar CRBA, LDL^T, pivoting LCP, task
epilogue, Philox auto-reset), lane by lane; what it cannot show is anything wave-level (votes, occupancy) or the fp32 hardware
approximations -- the `-m gpu` tests hold the real build to the same oracle."""
import numpy as np
import pytest

from dart_env_amd.model_card import card_for
from tests.emu_lib import EmuStepper
from tests.parity_protocol import make_reference, run_host_api


@pytest.mark.parametrize("env_id", ["DartHopper-v1", "DartWalker2d-v1"])
def test_fp64_kernel_code_meets_the_north_star_bound_on_cpu(env_id):
    card = card_for(env_id)
    acts, ref = make_reference(card, 128, 300)
    s = run_host_api(EmuStepper(card, 128, precision=64), acts, ref)
    assert s["done_flag_mismatches"] == 0 and s["episodes"] > 100
    assert s["q"] < 1e-9 and s["dq"] < 1e-8, (s["q"], s["dq"])          # bound to beat: 1e-4


def test_generic_and_baked_kernels_agree():
    """runtime-parameter kernel (any card of the topology) = compile-time specialisation, bit for bit in fp64"""
    card = card_for("DartWalker2d-v1")
    acts, ref = make_reference(card, 64, 60)
    a = EmuStepper(card, 64, precision=64, allow_static=True); b = EmuStepper(card, 64, precision=64, allow_static=False)
    assert a.is_static and not b.is_static
    run_host_api(a, acts, ref); run_host_api(b, acts, ref)
    qa, dqa = a.get_state(); qb, dqb = b.get_state()
    assert np.abs(qa - qb).max() < 1e-12 and np.abs(dqa - dqb).max() < 1e-10
