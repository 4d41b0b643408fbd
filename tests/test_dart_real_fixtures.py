"""Parity against the REAL DART stack, whenever fixtures captured with tools/capture_dart_golden.py are committed under
tests/golden/dart_real/ (SURVEY.md 8(c) last row).  None exist in this build -- pydart2 / DART are available nowhere in
the build environment -- so these tests skip and the oracle stays "parity unpinned"; with the files present they pin
the oracle (CPU) and the HIP kernels (GPU) to DART itself, step by step from DART's own states."""
import glob
import os

import numpy as np
import pytest

from dart_env_amd.model_card import card_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "dart_real", "*.npz")))
needs_real = pytest.mark.skipif(not REAL, reason="no fixtures captured from real pydart2/DART (tools/capture_dart_golden.py)")
REF = os.environ.get("DART_REFERENCE", "/root/reference")

# one-step tolerances from DART's own state: the restatement must reproduce DART's step, not merely resemble it
TOL_Q, TOL_DQ = 1e-9, 1e-6


def _one_step_errors(step_fn, d):
    """Feed DART's state before every step, compare the state after it with DART's (episode ends skipped)."""
    q_prev, dq_prev = d["q0"], d["dq0"]
    eq = edq = 0.0
    for t in range(len(d["actions"])):
        q1, dq1 = step_fn(q_prev, dq_prev, d["actions"][t])
        eq = max(eq, np.abs(q1 - d["q"][t]).max()); edq = max(edq, np.abs(dq1 - d["dq"][t]).max())
        if d["done"][t]:
            break                       # the state after a reset comes from np_random, covered by the seeding tests
        q_prev, dq_prev = d["q"][t], d["dq"][t]
    return eq, edq


def oracle_one_step_errors(path):
    from tests.oracle_lib import OracleWorld
    d = np.load(path)
    w = OracleWorld(card_for(str(d["env_id"])))

    def step(q, dq, a):
        w.set_state(q, dq)
        w.env_step(a.astype(np.float64))
        return w.get_state()
    return _one_step_errors(step, d)


@needs_real
@pytest.mark.parametrize("path", REAL, ids=[os.path.basename(p) for p in REAL])
def test_oracle_one_step_matches_real_dart(path):
    eq, edq = oracle_one_step_errors(path)
    assert eq < TOL_Q and edq < TOL_DQ, (eq, edq)


def _capture(tmp, knob, envs, steps=40):
    import subprocess
    import sys
    out = os.path.join(str(tmp), knob or "default")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "golden", "capture_with_stub.py"), "--out", out, "--steps", str(steps), "--envs"] + envs
    if knob:
        cmd += ["--knob", knob]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    files = sorted(glob.glob(os.path.join(out, "*.npz")))
    assert len(files) == 2 * len(envs)          # per env and seed: full-scale and small actions
    return files


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gym", "envs", "dart")), reason="needs the reference's Python (build container only)")
def test_capture_pipeline_runs_end_to_end_and_its_check_can_fail(tmp_path):
    """VERDICT r4 item 8.  tools/capture_dart_golden.py -- the script that pins the oracle the day a pydart2 box exists -- is run HERE,
    unchanged, against the stub pydart2 of make_golden.py (reference Python over this repo's oracle).  (i) Its fixtures carry every field
    this file's checks read and pass them trivially (the captured "DART" IS the oracle); (ii) with one Appendix-C knob flipped in the
    captured world -- A3: the impulse pass's inertia, A9: the ContactConstraint constants -- the SAME check fails.  So a real capture
    that disagrees with the restatement cannot pass unnoticed, and the only step left for real parity is running one script."""
    envs = ["DartHopper-v1", "DartWalker2d-v1"]
    for path in _capture(tmp_path, "", envs):
        d = np.load(path)
        for key in ("env_id", "q0", "dq0", "actions", "q", "dq", "done", "obs", "reward", "truncated", "reset_obs", "ncontacts", "pydart2_version"):
            assert key in d.files, (path, key)
        assert d["actions"].dtype == np.float32 and len(d["actions"]) == 40 and (d["ncontacts"] >= 0).all()
        eq, edq = oracle_one_step_errors(path)
        assert eq < TOL_Q and edq < TOL_DQ, (path, eq, edq)
        assert eq == 0.0 and edq == 0.0               # same arithmetic on both sides: bitwise, not merely within tolerance
    for knob in ("A3", "A9"):
        worst = [oracle_one_step_errors(p) for p in _capture(tmp_path, knob, envs)]
        assert any(eq >= TOL_Q or edq >= TOL_DQ for eq, edq in worst), (knob, worst)      # the check notices a world that is not the oracle's


@needs_real
@pytest.mark.gpu
@pytest.mark.parametrize("path", REAL, ids=[os.path.basename(p) for p in REAL])
def test_kernel_one_step_matches_real_dart(path):
    from dart_env_amd.stepper import HipStepper
    d = np.load(path)
    s = HipStepper(card_for(str(d["env_id"])), 1, precision=64)

    def step(q, dq, a):
        s.set_state(q[None], dq[None])
        s.step(a[None].astype(np.float32))
        qq, dd = s.get_state()
        return qq[0], dd[0]
    eq, edq = _one_step_errors(step, d)
    s.close()
    assert eq < TOL_Q and edq < TOL_DQ, (eq, edq)
