"""Round 5: a step kernel's results must not depend on what the vector / accumulator registers, the private segment
or LDS held before the launch.  One build of the half cheetah's kernel did -- the toolchain had placed five VGPR -> AGPR spill copies ahead of
the EXEC restore of a join block (tools/exec_prologue_lint.py; root cause and evidence in DESIGN.md section 4.1 / profiles/r05_first_launch.txt),
so lanes that had skipped a divergent region reloaded accumulator registers nobody had written for them: whatever the previous kernel left
there.  That shows as "first launch differs from the later ones" only by accident; the direct test is to CONTROL the leftovers.

Each case runs in a fresh process (tools/gpu/first_launch_probe.py): four identical rollouts at 65 536 envs, with every register / the
scratch backing / all LDS set to a different pattern before each of them -- the digests must be identical.  The poisoning kernels are
test infrastructure (tests/gpu_kernels/poison_harness.hip, built by __graft_entry__.build())."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("DartHopper-v1", 64), ("DartHopper-v1", 32), ("DartWalker2d-v1", 64), ("DartWalker2d-v1", 32), ("DartHalfCheetah-v1", 64), ("DartHalfCheetah-v1", 32),
         ("DartHumanWalker-v1", 64)]


def _probe(env_id, prec, extra, n=None):
    n = n or ("16384" if env_id == "DartHumanWalker-v1" else "65536")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "gpu", "first_launch_probe.py"), "--env", env_id, "--prec", str(prec), "--n", str(n)] + (["--steps", "3"] if "--steps" not in extra else []) + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if "digests" in l][-1]
    return re.search(r"digests ((?:[0-9a-f]{10} ?)+)\|", line).group(1).split()


@pytest.mark.parametrize("env_id,prec", CASES)
def test_results_do_not_depend_on_leftovers_in_registers_scratch_or_lds(env_id, prec):
    plain = _probe(env_id, prec, ["--poison", "none"])
    assert len(set(plain)) == 1, plain                                # the round-4 form of the test: first launch == later launches
    d = _probe(env_id, prec, ["--poison", "all", "--when", "both", "--pattern", "random"])     # registers, scratch and LDS, a fresh pattern before every rollout
    assert len(set(d)) == 1 and d[0] == plain[0], (d, plain)          # same bits every time, and the plain run's


# Round 6: the exec-prologue lint matches ONE manifestation of the toolchain defect; the poison test is the real gate, and
# the same toolchain builds every kernel family.  So: the rest of the lane kernels (snake chain, cart + 1 / 2 links, the two-link arm, the
# 3-D chain), the tree kernel's other instantiations (link-link contacts: Walker3d; FreeJoint root + LDS solver: Dog; the fp32 pattern
# kernel), and a physics-only card (`envs.DartEnv` on a user's .skel: the PhysTopo kernels / task 0 of the cart, arm and chain kernels) of
# every compiled shape.  One fresh process per case (--when combined: two undisturbed rollouts, then every rollout behind a fresh random
# poisoning of registers, scratch and LDS), smaller batches than the bench configs above -- whole waves either way.
MORE_ENVS = [("DartSnake7Link-v1", 64, 16384), ("DartSnake7Link-v1", 32, 16384), ("DartCartPole-v1", 64, 16384), ("DartCartPoleSwingUp-v1", 32, 16384),
             ("DartDoubleInvertedPendulumEnv-v1", 64, 16384), ("DartReacher-v1", 64, 16384), ("DartReacher3d-v1", 64, 16384), ("DartReacher3d-v1", 32, 16384),
             ("DartWalker3d-v1", 64, 4096), ("DartWalker3d-v1", 32, 4096), ("DartDog-v1", 64, 4096), ("DartDog-v1", 32, 4096), ("DartHumanWalker-v1", 32, 4096),
             ("DartWalker3dSPD-v1", 64, 2048)]
PHYS = [("hopper", 64), ("walker2d", 64), ("walker2d", 32), ("halfcheetah", 64), ("halfcheetah", 32), ("snake7link", 64), ("cartpole", 64), ("double_pendulum", 64),
        ("reacher2d", 64), ("reacher3d", 64)]


@pytest.mark.parametrize("env_id,prec,n", MORE_ENVS)
def test_every_other_kernel_family_is_independent_of_leftovers(env_id, prec, n):
    # (the SPD task carries the previous step's constraint forces into its controller -- walker3d_spd.py:40-55 -- which set_state does not
    # touch, in the reference as here: its rollouts start equal only on a fresh handle each)
    extra = ["--fresh"] if env_id == "DartWalker3dSPD-v1" else []
    d = _probe(env_id, prec, ["--poison", "all", "--when", "combined", "--pattern", "random", "--reps", "5"] + extra, n=n)
    assert len(d) == 5 and len(set(d)) == 1, d


@pytest.mark.parametrize("model,prec", PHYS)
def test_physics_only_kernels_are_independent_of_leftovers(model, prec):
    d = _probe("-", prec, ["--phys", model, "--poison", "all", "--when", "combined", "--pattern", "random", "--reps", "5"], n=16384)
    assert len(d) == 5 and len(set(d)) == 1, d


# Round 6, late: none of the cases above resets an env -- and the one abort the GPU suite produced with a traceback (profiles/r06_crash_hunt.txt,
# part 2) had its main thread in a step of the MT19937 auto-reset test.  So the reset epilogues get the same treatment: the step kernel's own
# Philox reset, its MT19937 reset (mt19937_draw.hpp) and the two launches the latter replaced (mt_draw + masked reset: "mt-split"), with a
# 2-step TimeLimit on top of the tasks' own terminations and a fresh, equally seeded handle per rollout.
RESETS = [("DartHopper-v1", 64, 65536, "mt"), ("DartHopper-v1", 32, 65536, "mt"), ("DartHopper-v1", 64, 65536, "philox"), ("DartWalker2d-v1", 64, 65536, "mt"),
          ("DartHalfCheetah-v1", 64, 65536, "philox"), ("DartHalfCheetah-v1", 32, 65536, "mt"), ("DartSnake7Link-v1", 64, 16384, "mt"),
          ("DartReacher3d-v1", 64, 16384, "mt"), ("DartHumanWalker-v1", 64, 4096, "philox"), ("DartDog-v1", 32, 4096, "mt")]


@pytest.mark.parametrize("env_id,prec,n,mode", RESETS)
def test_reset_epilogues_are_independent_of_leftovers(env_id, prec, n, mode):
    common = ["--poison", "all", "--when", "combined", "--pattern", "random", "--reps", "5", "--steps", "6"]
    d = _probe(env_id, prec, common + ["--autoreset", mode], n=n)
    assert len(d) == 5 and len(set(d)) == 1, d
    if mode == "mt" and env_id == "DartHopper-v1":      # and the fused reset is the two-launch reset, bit for bit, at the bench's batch size
        s = _probe(env_id, prec, common + ["--autoreset", "mt-split"], n=n)
        assert set(s) == set(d), (s, d)
