"""The host-buffer boundary on the device: `dart_step` into caller-owned arrays that the caller page-locked with
dart_register_host_buffer gives bitwise the results of the staging path, and the page-locked output blocks of the gym.vector surface
are leased to the arrays a step returns (explicit ownership, no refcount inference).
Reference call sites: gym/vector/sync_vector_env.py:73-84 (step_wait fills caller-visible arrays), vector_env.py:68-92."""
import gc

import numpy as np
import pytest

import dart_env_amd
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for

pytestmark = pytest.mark.gpu


def _arrays(n, card):
    return (np.zeros((n, card.act_dim), np.float32), np.zeros((n, card.obs_dim), np.float32), np.zeros(n, np.float64),
            np.zeros(n, np.uint8), np.zeros(n, np.uint8))


@pytest.mark.parametrize("env_id,n", [("DartHopper-v1", 4096), ("DartHumanWalker-v1", 256)])
def test_dart_step_into_registered_caller_buffers_equals_the_staging_path(env_id, n):
    card = card_for(env_id)
    rng = np.random.RandomState(3)
    runs = []
    for registered in (False, True):
        s = st.HipStepper(card, n, precision=64)
        s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 7)
        s.reset(None, None, None, want_obs=False)
        arrs = _arrays(n, card)
        if registered:
            for x in arrs:
                s.register_host_buffer(x)
            s.register_host_buffer(arrs[1])                       # registering a buffer twice is accepted
        rng = np.random.RandomState(3)
        rec = []
        for t in range(25):
            arrs[0][:] = rng.uniform(-1, 1, arrs[0].shape)
            s.step_into(*arrs)
            rec.append([x.copy() for x in arrs[1:]])
        if registered:
            s.unregister_host_buffer(arrs[2])
            with pytest.raises(st.StepperError):
                s.unregister_host_buffer(arrs[2])                  # not registered any more
            s.step_into(*arrs)                                     # one output outside the registered ranges: the staging path serves the call
        rec.append(list(s.get_state()))
        runs.append(rec)
        s.close()
    for a, b in zip(runs[0][:25], runs[1][:25]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert any(r[2].any() for r in runs[0][:25])                  # episodes ended: the auto-reset ran on both paths
    assert runs[0][0][1].dtype == np.float64 and np.abs(runs[0][3][1]).max() > 0


def test_a_partially_registered_arena_serves_every_argument():
    """one arena holding all five arrays, registered once: the usual C layout"""
    card = card_for("DartHopper-v1")
    n = 1024
    sizes = [4 * n * card.act_dim, 4 * n * card.obs_dim, 8 * n, n, n]
    offs = np.cumsum([0] + [(b + 63) // 64 * 64 for b in sizes])
    arena = np.zeros(int(offs[-1]), np.uint8)
    act = arena[offs[0]:offs[0] + sizes[0]].view(np.float32).reshape(n, card.act_dim)
    obs = arena[offs[1]:offs[1] + sizes[1]].view(np.float32).reshape(n, card.obs_dim)
    rew = arena[offs[2]:offs[2] + sizes[2]].view(np.float64)
    done = arena[offs[3]:offs[3] + sizes[3]]; trunc = arena[offs[4]:offs[4] + sizes[4]]
    s = st.HipStepper(card, n, precision=64); ref = st.HipStepper(card, n, precision=64)
    for g in (s, ref):
        g.configure(st.CFG_SEED, 2); g.reset(None, None, None, want_obs=False)
    s.register_host_buffer(arena)
    rng = np.random.RandomState(0)
    for t in range(10):
        act[:] = rng.uniform(-1, 1, act.shape)
        s.step_into(act, obs, rew, done, trunc)
        o, r, d, tr = ref.step(act.copy())
        assert np.array_equal(o, obs) and np.array_equal(r, rew) and np.array_equal(d, done.astype(bool)) and np.array_equal(tr, trunc.astype(bool))
    s.close(); ref.close()


def test_vector_env_outputs_are_leased_blocks():
    """copy=True semantics without a copy: the arrays of step t stay intact while the caller holds them, whatever later steps do, and
    their block is handed out again once they are garbage"""
    venv = dart_env_amd.vector.make("DartHopper-v1", 512)
    venv.seed(0); venv.reset()
    a = np.random.RandomState(1).uniform(-1, 1, (512, 3)).astype(np.float32)
    kept = venv.step(a)
    snap = [np.array(x, copy=True) for x in kept[:3]]
    held = [venv.step(a) for _ in range(6)]                         # more steps than the pool has blocks, all results held
    assert all(np.array_equal(x, y) for x, y in zip(kept[:3], snap))   # nobody wrote into the first step's arrays
    addr = {h[0].__array_interface__["data"][0] for h in held} | {kept[0].__array_interface__["data"][0]}
    assert len(addr) == 7                                            # seven live results, seven distinct buffers
    pool = venv.env._stepper._blocks
    assert sum(1 for e in pool if e[1]) == len(pool)                 # every pooled block is leased ...
    del kept, held
    gc.collect()
    assert not any(e[1] for e in pool)                               # ... and all of them come back once the arrays are gone
    first = venv.step(a)[0].__array_interface__["data"][0]
    assert first == pool[0][0].__array_interface__["data"][0]
    venv.close()


def test_output_blocks_are_driver_pinned_memory_that_outlives_the_handle():
    """Round 6: the blocks a step's results arrive in are dart_alloc_output's (hipHostMalloc) -- not numpy arrays locked with hipHostRegister,
    into which GPU writes faulted about once in ten runs of this suite (profiles/r06_crash_hunt.txt part 2) -- and they belong to the arrays
    handed out: results stay readable after close() and after the handle is garbage; the caller-registered form of the ABI still works and
    gives the same bits."""
    import ctypes as C
    card = card_for("DartHopper-v1")
    n = 640
    acts = np.random.RandomState(2).uniform(-1, 1, (8, n, 3)).astype(np.float32)

    def start():
        s = st.HipStepper(card, n, precision=64)
        s.configure(st.CFG_AUTORESET, 1); s.configure(st.CFG_SEED, 9)
        s.reset(None, None, None, want_obs=False)
        return s

    s = start()
    outs = [s.step(a) for a in acts]
    copies = [tuple(np.array(x, copy=True) for x in o) for o in outs]
    assert all(isinstance(e[0].base, st._PinnedBlock) for e in s._blocks) and 1 <= len(s._blocks) <= st.HipStepper._POOL_SETS
    s.close()
    del s
    gc.collect()
    for o, c in zip(outs, copies):                       # the memory is the arrays', not the handle's
        assert all(np.array_equal(x, y) for x, y in zip(o, c))
    # the same rollout through a caller-registered numpy block (dart_register_output: kept for callers that need their own memory)
    s = start()
    total, off = s._layout()
    blk = np.zeros(total, dtype=np.uint8)
    s._check(s.L.dart_register_output(s.h, blk.ctypes.data_as(C.c_void_p)))
    for t, a in enumerate(acts):
        s._check(s.L.dart_step_async_to(s.h, a.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p)))
        s._check(s.L.dart_step_wait(s.h, None, None, None, None))
        obs = blk[off[0]:off[0] + 4 * n * card.obs_dim].view(np.float32).reshape(n, card.obs_dim)
        rew = blk[total - ((8 * n + 255) & ~255):][:8 * n].view(np.float64)
        assert np.array_equal(obs, copies[t][0]) and np.array_equal(rew, copies[t][1]) and np.array_equal(blk[off[2]:off[2] + n].view(np.bool_), copies[t][2])
    s._check(s.L.dart_unregister_output(s.h, blk.ctypes.data_as(C.c_void_p)))
    # a block of dart_alloc_output can be withdrawn from the handle and freed by hand, and is refused afterwards
    p = C.c_void_p()
    s._check(s.L.dart_alloc_output(s.h, C.byref(p)))
    s._check(s.L.dart_unregister_output(s.h, p))
    assert s.L.dart_step_async_to(s.h, acts[0].ctypes.data_as(C.c_void_p), p) == st.E_INVALID
    assert s.L.dart_free_output(p) == st.DART_OK
    s.close()
