"""SURVEY.md 8(d) parity protocol -- TEST INFRASTRUCTURE (used by tests/ and by bench.py's cpu_baseline leg only).

Same seeds and actions into the fp64 oracle and the HIP stepper; resets follow the ORACLE's done flags and draw identical
Philox noise on both sides, so every env is always compared inside the same episode ("compare per-episode; resets fed identical
noise"); the statistic is the RMS over ALL envs x dofs (nothing trimmed) of the pre-reset state difference after
1, 10, 100, 1 000 env-steps.  North-star bound: < 1e-4.
"""
import time

import numpy as np

SNAP_STEPS = (1, 10, 100, 1000)
# oracle speed on one core (env-steps/s), only used to size the samples so the CPU side stays bounded
ORACLE_RATE = {"DartHopper-v1": 1.1e5, "DartWalker2d-v1": 7e4, "DartHumanWalker-v1": 3.4e3}


def snap_list(steps):
    snaps = [s for s in SNAP_STEPS if s <= steps]
    if steps not in snaps:
        snaps.append(steps)
    return snaps


def make_reference(card, ne, steps, threads=None):
    """action tensor + oracle trace of the sample"""
    from tests import oracle_lib as ol
    acts = np.random.RandomState(7).uniform(-1, 1, (steps, ne, card.act_dim)).astype(np.float32)
    return acts, ol.rollout_trace(card, acts, snap_list(steps), seed=0, env_offset=0, solver=0, threads=threads)


def snap_stats(qg, dqg, ref, si, smax=np.inf):
    """smax = card.state_abs_max: a state the reference's own validity test rejects -- `np.abs(s[2:]) < 100` in every task's done
    condition (hopper.py:60-62; human_walker.py reports it as info['broke_sim']) -- is an env that EXPLODED in its terminal step
    (a tumbling humanoid under full-scale random torques can do that within one env-step since the impulse pass runs on M, A3);
    the digits of an explosion are not a trajectory: such envs are counted (`envs_broken_sim`), like NaN states, not averaged --
    but ONLY when the oracle and the stepper AGREE that the env is broken.  An env that exploded on one side only
    (`envs_broken_one_side`) stays in the RMS with all its digits: a stepper-only explosion must move the headline figure."""
    qo, dqo = ref["q"][si], ref["dq"][si]
    eq, edq = qg - qo, dqg - dqo
    bad = ~np.isfinite(eq).all(axis=1) | ~np.isfinite(edq).all(axis=1)
    broke = np.zeros(len(eq), dtype=bool)
    one_side = np.zeros(len(eq), dtype=bool)
    if np.isfinite(smax):
        with np.errstate(invalid="ignore"):
            sides = [(np.abs(a[:, 2:]) >= smax).any(axis=1) | (np.abs(b) >= smax).any(axis=1) for a, b in ((qo, dqo), (qg, dqg))]
        broke = sides[0] & sides[1] & ~bad
        one_side = (sides[0] ^ sides[1]) & ~bad
    eq[bad] = 0.0; edq[bad] = 0.0
    incl = (float(np.sqrt(np.mean(eq ** 2))), float(np.sqrt(np.mean(edq ** 2))))    # with the broken-sim envs averaged in
    eq[broke] = 0.0; edq[broke] = 0.0                 # exploded envs are counted, not averaged
    return {"q": float(np.sqrt(np.mean(eq ** 2))), "dq": float(np.sqrt(np.mean(edq ** 2))),
            "q_incl_broken_sim": incl[0], "dq_incl_broken_sim": incl[1],
            "max_abs_q": float(np.abs(eq).max()), "max_abs_dq": float(np.abs(edq).max()),
            "envs_beyond_1e-4": int((np.abs(eq).max(axis=1) > 1e-4).sum()), "envs_non_finite": int(bad.sum()),
            "envs_broken_sim": int(broke.sum()), "envs_broken_one_side": int(one_side.sum())}


def summarize(per_snap, snaps, mism, ne, steps, ref, seconds):
    last = per_snap[str(snaps[-1])]
    return {"q": max(v["q"] for v in per_snap.values()), "dq": max(v["dq"] for v in per_snap.values()),
            "at_last_step": {"q": last["q"], "dq": last["dq"]}, "by_step": per_snap, "untrimmed": True,
            "done_flag_mismatches": int(mism), "envs": ne, "env_steps": steps, "episodes": int(ref["done"].sum()),
            "protocol": "same Philox reset streams and action tensor; resets follow the oracle's done flags; RMS over all envs x "
                        "dofs of the pre-reset state difference; q / dq = worst over the listed steps",
            "stepper_seconds": seconds}


def run_host_api(stepper, acts, ref):
    """the protocol through the host-buffer API (`step(actions)`, `reset(mask)`): HipStepper or the kernel emulator"""
    from dart_env_amd import stepper as st
    steps, ne, _ = acts.shape
    snaps = [int(x) for x in ref["snap_steps"]]
    stepper.configure(st.CFG_AUTORESET, 0); stepper.configure(st.CFG_SEED, 0); stepper.configure(st.CFG_ENV_OFFSET, 0)
    stepper.reset(None, None, None, want_obs=False)
    per_snap, si, mism = {}, 0, 0
    t0 = time.perf_counter()
    for t in range(steps):
        _, _, dg, _ = stepper.step(acts[t])
        mism += int((dg.astype(np.uint8) != ref["done"][t]).sum())
        if si < len(snaps) and snaps[si] == t + 1:
            qg, dqg = stepper.get_state()
            per_snap[str(t + 1)] = snap_stats(qg, dqg, ref, si, float(getattr(stepper.card, "state_abs_max", np.inf)))
            si += 1
        if ref["done"][t].any():
            stepper.reset(ref["done"][t], None, None, want_obs=False)
    return summarize(per_snap, snaps, mism, ne, steps, ref, time.perf_counter() - t0)


def parity_sample_size(env_id, steps, cores, budget_s, cap=4096):
    """envs of the parity sample: as many as the oracle finishes in ~budget_s on `cores` cores, multiple of 64, <= cap"""
    ne = int(budget_s * cores * ORACLE_RATE.get(env_id, 3e3) / max(1, steps))
    return max(64, min(cap, ne // 64 * 64))


def parity_check(env_id, precision, ne, steps, local_rank, all_bodies_collide=None, ref=None, acts=None, impulse_inertia=None):
    """SURVEY.md 8(d) parity protocol on `ne` envs x `steps` env-steps.  Returns (stats, ref, acts); ref / acts can be reused
    for the other precision.  The oracle is the checker here (cpu_baseline leg) -- never the thing measured or shipped."""
    import torch
    from dart_env_amd import stepper as st
    from dart_env_amd.model_card import card_for
    card = card_for(env_id) if all_bodies_collide is None else card_for(env_id, all_bodies_collide=all_bodies_collide)
    if impulse_inertia is not None:
        card.impulse_inertia = impulse_inertia     # A3 knob (include/dart_model_card.h); None = the card's default (0 = DART_IMPULSE_MASS: DART 6)
    snaps = snap_list(steps)
    if ref is None:
        acts, ref = make_reference(card, ne, steps)
    dev = torch.device("cuda", local_rank)
    gpu = st.HipStepper(card, ne, device=local_rank, precision=precision)
    gpu.configure(st.CFG_AUTORESET, 0); gpu.configure(st.CFG_SEED, 0); gpu.configure(st.CFG_ENV_OFFSET, 0)
    d_acts = torch.from_numpy(acts).to(dev)
    d_done_ref = torch.from_numpy(ref["done"]).to(dev)
    d_done = torch.empty((ne,), device=dev, dtype=torch.uint8)
    mism = torch.zeros((), device=dev, dtype=torch.int64)
    # steps, flag compares and resets all go to ONE side stream of torch's (a null stream handle would mean "the handle's own
    # stream" to the C ABI, and torch's compares would then race the kernels)
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    ts = side.cuda_stream
    assert ts != 0
    gpu.reset_device(0, 0, ts)
    per_snap, si = {}, 0
    stride = ne * card.act_dim * 4
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for t in range(steps):
            gpu.step_device(d_acts.data_ptr() + t * stride, 0, 0, d_done.data_ptr(), 0, ts)
            mism += (d_done != d_done_ref[t]).sum()
            if si < len(snaps) and snaps[si] == t + 1:
                qg, dqg = gpu.get_state()
                per_snap[str(t + 1)] = snap_stats(qg, dqg, ref, si, float(card.state_abs_max))
                si += 1
            gpu.reset_device(d_done_ref[t].data_ptr(), 0, ts)     # resets follow the oracle's episodes, identical Philox noise
    torch.cuda.synchronize()
    gpu.sync()
    gpu_s = time.perf_counter() - t0
    gpu.close()
    stats = summarize(per_snap, snaps, int(mism.item()), ne, steps, ref, gpu_s)
    return stats, ref, acts


