"""CPU-side checks: the C-ABI library loads and exports everything include/dart_stepper.h declares, fails loudly
without a GPU (no CPU fallback), the model compiler's cards are sane, and the host layer's error behaviour."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import dart_env_amd
from dart_env_amd import stepper as st
from dart_env_amd.model_card import DartModelCard, card_for, load_model
from dart_env_amd.distributed import shard_layout
from tests.conftest import gpu_available
from tests.fake_stepper import OracleStepper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dart_stepper.h")).read()
    declared = sorted(set(re.findall(r"\b(dart_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    assert sorted(st.EXPORTS) == declared, (sorted(st.EXPORTS), declared)
    L = st.load_library()
    for name in declared:
        assert hasattr(L, name), name


def test_card_struct_layout_matches_c():
    # oracle_create() compares struct_bytes with its own sizeof(DartModelCard): passing means ctypes == C layout
    from tests.oracle_lib import OracleWorld
    c = card_for("DartHopper-v1")
    assert c.struct_bytes == C.sizeof(DartModelCard)
    OracleWorld(c)


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(st.StepperError) as e:
        st.HipStepper(card_for("DartHopper-v1"), 8)
    assert e.value.code == st.E_NO_DEVICE and "no CPU path" in str(e.value)
    with pytest.raises(st.StepperError):
        dart_env_amd.make("DartHopper-v1")
    with pytest.raises(st.StepperError):
        dart_env_amd.vector.make("DartHopper-v1", 4)


def test_create_rejects_bad_arguments():
    L = st.load_library()
    h = C.c_void_p()
    bad = card_for("DartHopper-v1")
    bad.version = 99
    assert L.dart_create(C.byref(bad), 4, 0, 32, C.byref(h)) == st.E_INVALID
    assert L.dart_create(C.byref(card_for("DartHopper-v1")), 0, 0, 32, C.byref(h)) == st.E_INVALID
    assert L.dart_create(C.byref(card_for("DartHopper-v1")), 4, 0, 16, C.byref(h)) == st.E_INVALID
    knob = card_for("DartHopper-v1")
    knob.impulse_inertia = 7                      # neither DART_IMPULSE_MASS (0) nor DART_IMPULSE_AUGMENTED (1)
    assert L.dart_create(C.byref(knob), 4, 0, 64, C.byref(h)) == st.E_INVALID
    assert b"impulse_inertia" in L.dart_last_error(None)
    assert L.dart_destroy(None) == st.DART_OK
    assert L.dart_step(None, None, None, None, None, None) == st.E_INVALID


def test_a_zero_initialised_card_gets_the_documented_defaults():
    """a C caller that memsets a DartModelCard must get DART 6's impulse rule (A3) -- the default is encoded as 0 --
    and the shipped cards carry that default."""
    z = DartModelCard()
    assert z.impulse_inertia == 0
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "dart_model_card.h")).read()
    assert "DART_IMPULSE_MASS = 0" in hdr and "DART_IMPULSE_AUGMENTED = 1" in hdr
    for env_id in ("DartHopper-v1", "DartWalker2d-v1", "DartHumanWalker-v1"):
        assert card_for(env_id).impulse_inertia == 0


def test_model_cards():
    h = load_model("hopper")
    assert h.ndofs == 6 and [b.name for b in h.bodies][:3] == ["h_pelvis_aux2", "h_pelvis_aux", "h_pelvis"]
    assert h.total_mass == pytest.approx(15.26499871)
    assert h.bodies[2].axes[0][2] == -1.0                      # j_pelvis_rot axis -z (hopper_capsule.skel:188)
    assert list(h.limited) == [False, False, False, True, True, True]
    assert h.upper[3] == 0.0 and h.lower[5] == pytest.approx(-0.785398)
    assert h.ground_y == 0.0 and h.dt == 0.002
    # A1: inertia = first shape's own-frame tensor, capsule axis z -> body Izz is the AXIAL moment
    assert h.bodies[2].inertia[2, 2] == pytest.approx(0.004292, abs=1e-6)
    w = load_model("walker2d")
    assert w.ndofs == 9 and w.bodies[2].axes[0][2] == 1.0 and w.total_mass == pytest.approx(22.69800692)
    assert [b.parent for b in w.bodies] == [-1, 0, 1, 2, 3, 4, 2, 6, 7]
    assert load_model("walker3d").ndofs == 21 and load_model("humanwalker").ndofs == 29


@pytest.mark.skipif(not os.path.exists("/root/reference"), reason="reference assets only exist in the build container")
def test_model_compiler_reproduces_committed_cards():
    from dart_env_amd.skel import parse_skel
    a = parse_skel("/root/reference/gym/envs/dart/assets/hopper_capsule.skel", dt=0.002)
    assert a.to_json() == load_model("hopper").to_json()


def test_generated_kernel_headers_are_current():
    """csrc/static_models.hpp and csrc/tree_patterns.hpp are generated from the committed model cards: regenerating must give the
    committed text, and the HumanWalker factor pattern must be the no-fill pattern of its tree (ancestor pairs only)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tree_patterns.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "dart_env_amd", "csrc", "tree_patterns.hpp")).read()
    before = open(os.path.join(ROOT, "dart_env_amd", "csrc", "static_models.hpp")).read()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_static_models.py")], capture_output=True, text=True, check=True).stdout
    assert "wrote" not in out and before == open(os.path.join(ROOT, "dart_env_amd", "csrc", "static_models.hpp")).read()   # rewrites only when stale
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_tree_patterns import factor_pattern
    rows = factor_pattern(card_for("DartHumanWalker-v1"))
    assert len(rows) == 29 and sum(bin(r).count("1") for r in rows) == 222
    assert rows[28] == (1 << 28) - 1                       # the first root dof (last storage row) couples with everything


def test_vector_env_api_errors_and_shapes():
    venv = dart_env_amd.vector.make("DartHopper-v1", 3, stepper_factory=OracleStepper)
    assert venv.observation_space.shape == (3, 11) and venv.single_action_space.shape == (3,)
    assert len(venv.action_space) == 3 and venv.single_observation_space.dtype == np.float32
    venv.seed(0)
    venv.reset()
    with pytest.raises(st.NoAsyncCallError):
        venv.step_wait()
    venv.step_async(np.zeros((3, 3)))
    with pytest.raises(st.AlreadyPendingCallError):
        venv.step_async(np.zeros((3, 3)))
    obs, rew, done, infos = venv.step_wait()
    assert obs.shape == (3, 11) and rew.shape == (3,) and done.shape == (3,) and len(infos) == 3
    a = venv.action_space.sample()
    assert len(a) == 3 and a[0].dtype == np.float32
    venv.close()
    venv.close()  # idempotent
    with pytest.raises(dart_env_amd.vector.ClosedEnvironmentError):
        venv.reset()


def test_determinism_like_reference_test():
    """reference gym/envs/tests/test_determinism.py:6-54: two fresh envs, same seeds, 4 steps, exact equality."""
    outs = []
    for _ in range(2):
        env = dart_env_amd.make("DartHopper-v1", stepper_factory=OracleStepper)
        env.seed(0); env.action_space.seed(0)
        rec = [env.reset()]
        for _ in range(4):
            ob, r, d, info = env.step(env.action_space.sample())
            rec += [ob, np.array(r), np.array(d)]
        outs.append(rec)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_smoke_like_reference_test_envs():
    """reference gym/envs/tests/test_envs.py:10-37: reset in obs space, one random step, scalar reward, bool done."""
    for env_id in ("DartHopper-v1", "DartWalker2d-v1"):
        env = dart_env_amd.make(env_id, stepper_factory=OracleStepper)
        ob = env.reset()
        assert env.observation_space.contains(ob.astype(np.float32))
        ob, r, d, info = env.step(env.action_space.sample())
        assert env.observation_space.contains(ob.astype(np.float32)) and np.isscalar(r) and isinstance(d, bool)
        env.close()


def test_shard_layout():
    assert [shard_layout(524288, 8, r) for r in range(8)] == [(r * 65536, 65536) for r in range(8)]
    lay = [shard_layout(10, 4, r) for r in range(4)]
    assert lay == [(0, 3), (3, 3), (6, 2), (8, 2)]


def test_record_episode_statistics_wrappers():
    """RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:22-34): single-env wrapper and the vector
    one (host fallback here, device accumulators in the GPU suite) agree with a hand accumulation of the rewards."""
    import dart_env_amd
    from dart_env_amd.envs import DartHopperEnv
    from dart_env_amd.wrappers import RecordEpisodeStatistics, TimeLimit, VectorRecordEpisodeStatistics
    from tests.fake_stepper import OracleStepper
    env = RecordEpisodeStatistics(TimeLimit(DartHopperEnv(stepper_factory=OracleStepper), max_episode_steps=30), deque_size=5)
    env.seed(0)
    env.reset()
    rs = np.random.RandomState(0)
    ret, ln, n_ep = 0.0, 0, 0
    for t in range(150):
        ob, r, done, info = env.step(rs.uniform(-1, 1, 3))
        ret += r; ln += 1
        if done:
            assert info["episode"]["r"] == pytest.approx(ret, abs=1e-12) and info["episode"]["l"] == ln and info["episode"]["t"] >= 0
            n_ep += 1; ret, ln = 0.0, 0
            env.reset()
        else:
            assert "episode" not in info
    assert n_ep >= 5 and len(env.return_queue) == 5 and len(env.length_queue) == 5
    venv = VectorRecordEpisodeStatistics(dart_env_amd.vector.make("DartHopper-v1", 6, stepper_factory=OracleStepper))
    venv.seed(1)
    venv.reset()
    acc, cnt, fin = np.zeros(6), np.zeros(6, dtype=int), 0
    for t in range(60):
        ob, r, done, infos = venv.step(rs.uniform(-1, 1, (6, 3)).astype(np.float32))
        acc += r; cnt += 1
        for i in range(6):
            if done[i]:
                assert infos[i]["episode"]["r"] == pytest.approx(acc[i], abs=1e-9) and infos[i]["episode"]["l"] == cnt[i]
                acc[i] = 0; cnt[i] = 0; fin += 1
            else:
                assert "episode" not in infos[i]
    assert fin >= 6 and len(venv.return_queue) == min(fin, 100)
    venv.close()


def test_batched_mt_keys_equal_the_per_seed_path():
    from dart_env_amd import seeding
    seeds = list(range(0, 3000, 7)) + [2 ** 63 + 5, 0, 12345678901234567890 % 2 ** 64]
    keys, klen = seeding.mt_keys(seeds)
    for i, s in enumerate(seeds):
        w = seeding.int_list_from_bigint(seeding.hash_seed(seeding.create_seed(s)))
        assert klen[i] == len(w) and list(keys[i, :len(w)]) == w


def test_public_headers_are_plain_c():
    """include/*.h is the C ABI: it must compile as C99 (and as C++) on its own, with no HIP / torch types in it."""
    import shutil, subprocess, tempfile
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "hc.c")
        with open(src, "w") as f:
            f.write('#include "dart_stepper.h"\n#include "dart_model_card.h"\nint main(void) { return (int)sizeof(DartModelCard) == 0; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, src])
        subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-I", inc, src])
    text = open(os.path.join(inc, "dart_stepper.h")).read() + open(os.path.join(inc, "dart_model_card.h")).read()
    assert "hipStream_t" not in text.replace("(a hipStream_t", "") and "torch" not in text and "at::" not in text


def test_tuple_space_fast_paths_equal_the_reference_loop():
    """Tuple((box,) * N): seeding the repeated object once and sampling all N in one call give the stream of the reference's
    per-space loops (gym/spaces/tuple.py, box.py:70-110)."""
    from dart_env_amd import spaces
    box = spaces.Box(np.array([-1.0, -2.0, 0.5]), np.array([1.0, 3.0, 0.75]))
    tup = spaces.Tuple((box,) * 257)
    tup.seed(11)
    fast = tup.sample(); fast2 = tup.sample()
    box.seed(11)
    slow = [box.sample() for _ in range(257)]; slow2 = [box.sample() for _ in range(257)]
    assert all(a.dtype == np.float32 and np.array_equal(a, b) for a, b in zip(fast, slow))
    assert all(np.array_equal(a, b) for a, b in zip(fast2, slow2))
    mixed = spaces.Tuple((box, spaces.Box(-np.ones(2), np.ones(2))))       # distinct objects: the plain loop
    mixed.seed(3)
    assert len(mixed.sample()) == 2


def test_output_blocks_are_leased_to_the_callers_arrays_explicitly():
    """reuse of the page-locked output blocks no longer hinges on sys.getrefcount.  A block returns to the pool when
    the last array derived from the step's outputs is garbage -- however many other references to the block itself exist."""
    import gc

    import ctypes as C

    class FakeLib:                      # the three entry points the pool needs; no device behind them
        def __init__(self):
            self.bufs, self.freed, self.total = {}, [], 0

        def dart_output_layout(self, h, tot, off):
            n, k = 8, 3
            ob, rb, db = 4 * n * k, 4 * n, n
            self.total = tot._obj.value = ob + rb + 2 * db + ((8 * n + 255) & ~255)     # ... + the float64 rewards behind the device block (round 5)
            for i, v in enumerate((0, ob, ob + rb, ob + rb + db)):
                off[i] = v
            return st.DART_OK

        def dart_alloc_output(self, h, p):          # round 6: the blocks are the library's page-locked memory, owned by the caller
            b = C.create_string_buffer(self.total)
            self.bufs[C.addressof(b)] = b
            p._obj.value = C.addressof(b)
            return st.DART_OK

        def dart_free_output(self, addr):
            self.freed.append(addr)
            return st.DART_OK

    lib = FakeLib()
    s = object.__new__(st.HipStepper)
    s.L, s.h, s.num_envs, s.obs_dim = lib, None, 8, 3
    b0 = s._free_block()
    snoop = [b0, b0, b0]                # a tool holding extra references to the block (what broke the refcount test)
    obs, rew, done, trunc = s._block_views(b0)
    assert obs.shape == (8, 3) and rew.dtype == np.float64 and done.dtype == np.bool_
    b1 = s._free_block()
    assert b1 is not b0                 # the caller still holds the arrays of b0's step: another block
    row = obs[2:4]                      # a derived view keeps the lease too
    del obs, rew, done, trunc
    gc.collect()
    assert s._free_block() is b1        # (b1 was never leased: still the first free one) ...
    o1 = s._block_views(b1)
    b2 = s._free_block()
    assert b2 is not b0 and b2 is not b1     # ... and with b1 leased and `row` alive, b0 is NOT handed out again
    del row
    gc.collect()
    assert s._free_block() is b0        # the last view is gone: b0 is free again, `snoop` notwithstanding
    # a block's memory goes back only when nothing refers to it any more -- the pool (the handle) AND every array handed out
    addrs = {b.__array_interface__["data"][0] for b in (b0, b1, b2)}
    keep = o1[0]                        # b1's observations, still held by the caller when the handle goes
    del o1, snoop, b0, b1, b2
    s.__dict__.clear(); del s
    gc.collect()
    assert len(lib.freed) == 2 and set(lib.freed) < addrs, lib.freed
    assert keep.shape == (8, 3)
    del keep
    gc.collect()
    assert set(lib.freed) == addrs and len(lib.freed) == 3
