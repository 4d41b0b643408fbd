#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched HIP stepper on BASELINE.json's headline config, with the parity it is quoted at.

A "step" is one batched env.step() of DartHopper-v1 over `--envs` (default 65 536) environments per GPU: clamp/scale action,
frame_skip=4 world steps (dynamics + contact/limit LCP + integration), reward, done, TimeLimit, observation and on-device
auto-reset -- one kernel launch.  Inputs (a ring of random action batches, U[-1,1) float32) are resident in HBM before the
timed region; outputs stay in HBM.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 2000 --warmup 200
    python bench.py --gpus 8 --steps 2000 --warmup 200      # no launcher (RANK unset): bench.py starts its own 8 ranks, one per GPU

Precision.  The headline runs the fp64 instantiation of the kernels (`--precision 64`, the default): it is the mode that meets
the north star's tolerance (RMS state error < 1e-4 over 1 000 env-steps, untrimmed; measured here ~1e-10), and on MI355X fp64
vector FMAs cost the Hopper kernel ~25 %.  The fp32 instantiation is reported beside it as `fast_mode` with ITS error.

Multi-GPU: envs are independent, so ranks own disjoint env shards (Philox streams keyed by global env index) and there is NO
collective inside the timed region; one RCCL all_gather of the last step's obs/reward/done runs after it (the "gather
rollouts" exchange of the north star) and is reported separately as gather_ms.

Before the measured batch is created, a scratch batch of the same kind is stepped for `--spinup-ms` (default 60 ms, untimed, reported
as `device_spinup`): W + K = 25 launches are under a millisecond, shorter than an idle GPU's clock ramp (35.3 vs 33.5 us per step).

Rank 0 prints ONE JSON line (contract in the task description) with, at N = 1, these extra objects:
  roofline       HBM roofline of the step kernel, kernel time from HIP events on the kernel's stream over the timed region
  steady_state   the same launch timed over a longer window after the episodes have de-synchronised
  fast_mode      the other precision on the same workload: throughput and state error
  other_configs  BASELINE configs 3 and 4 (DartWalker2d-v1 @ 65 536, DartHumanWalker-v1 @ 16 384): throughput, roofline, error
  cpu_baseline   the fp64 CPU oracle timed on one host core and on all usable cores, and `rms_state_err`: the GPU against it
                 under SURVEY.md 8(d)'s protocol -- same seeds and actions, resets follow the oracle's done flags with identical
                 noise, RMS over ALL envs x dofs of the pre-reset state difference after 1, 10, 100, 1 000 env-steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
RING_DEFAULT = 64               # distinct random action batches resident in HBM (time_config)
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X_MICROARCH.md: FP32 vector peak; the FP64 vector FMA runs at half that rate


def algorithmic_bytes(card) -> int:
    """read q,dq (8n) + actions (4 act) + write q,dq (8n) + obs (4 obs) + reward 4 + done 1 (fp32 state; the fp64
    mode moves 16n more, which is not counted: the figure is the algorithm's, not the implementation's)."""
    return 16 * card.ndofs + 4 * card.act_dim + 4 * card.obs_dim + 4 + 1


def shard_of(rank: int, envs_per_gpu: int):
    """contiguous env shards: rank g owns global envs [g n, (g+1) n); returns (env_offset, count)"""
    return rank * envs_per_gpu, envs_per_gpu


def default_envs(env_id: str) -> int:
    return 16384 if env_id == "DartHumanWalker-v1" else 65536


class HipBenchEnv:
    """One shard of envs on one GPU with its inputs and outputs resident in HBM (the thing bench.py times)."""

    def __init__(self, env_id, n, local_rank, precision, env_offset, ring=RING_DEFAULT, ring_seed=1234, all_bodies_collide=None, configure=()):
        import torch
        from dart_env_amd import stepper as st
        from dart_env_amd.model_card import card_for
        self.torch, self.st = torch, st
        self.card = card_for(env_id) if all_bodies_collide is None else card_for(env_id, all_bodies_collide=all_bodies_collide)
        self.n, self.ring_len = n, ring
        self.dev = torch.device("cuda", local_rank)
        self.env = st.HipStepper(self.card, n, device=local_rank, precision=precision)
        from dart_env_amd.model_card import TASK_CARTPOLE_SWINGUP, TASK_DOUBLE_PENDULUM, TASK_REACHER2D, TASK_REACHER3D
        if self.card.task in (TASK_CARTPOLE_SWINGUP, TASK_DOUBLE_PENDULUM, TASK_REACHER2D, TASK_REACHER3D):
            # reset_model of these tasks draws more than the two noise vectors (swing-up sign, reach targets, Gaussian velocities): the device MT19937
            # bank draws all of it, the in-kernel Philox reset would leave degenerate episodes (the ABI refuses that combination)
            from dart_env_amd import seeding
            keys, klen = seeding.mt_keys([env_offset + i for i in range(n)])
            self.env.seed_mt19937(keys, klen)
        self.env.configure(st.CFG_AUTORESET, 1)
        self.env.configure(st.CFG_SEED, 0)
        self.env.configure(st.CFG_ENV_OFFSET, env_offset)
        for k, v in configure:
            self.env.configure(k, v)
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(ring_seed)
        self.ring = (torch.rand((ring, n, self.card.act_dim), device=self.dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
        self.obs = torch.empty((n, self.card.obs_dim), device=self.dev, dtype=torch.float32)
        self.rew = torch.empty((n,), device=self.dev, dtype=torch.float32)
        self.done = torch.empty((n,), device=self.dev, dtype=torch.uint8)
        self.trunc = torch.empty((n,), device=self.dev, dtype=torch.uint8)
        self.stride = n * self.card.act_dim * 4

    def reset(self):
        self.env.reset_device(0, self.obs.data_ptr())
        self.env.sync()

    def run(self, k, base=0):
        for i in range(k):
            self.env.step_device(self.ring.data_ptr() + ((base + i) % self.ring_len) * self.stride, self.obs.data_ptr(),
                                 self.rew.data_ptr(), self.done.data_ptr(), self.trunc.data_ptr())

    def timed_steps(self, k) -> float:
        """k back-to-back batched steps bracketed by HIP events on the stepper's own stream; returns ms per step"""
        return self.env.time_steps(self.ring.data_ptr(), self.ring_len, k, self.obs.data_ptr(), self.rew.data_ptr(),
                                   self.done.data_ptr(), self.trunc.data_ptr())

    def mark(self, which):
        self.env.timer_mark(which)

    def elapsed_ms(self) -> float:
        return self.env.timer_elapsed()

    def sync(self):
        self.env.sync()
        self.torch.cuda.synchronize()

    def done_fraction(self) -> float:
        return float(self.done.float().mean().item())

    def packed_last(self):
        """the last step's outputs as ONE byte block in HBM -- obs f32 | reward f32 | done u8, the flags as the bytes they are -- the
        payload of the "gather rollouts" all-gather (same packing as ShardedDartVectorEnv.gather_rollout_device)"""
        from dart_env_amd.distributed import ShardedDartVectorEnv
        return ShardedDartVectorEnv._pack(self.obs, self.rew, self.done)

    def is_static(self) -> bool:
        return bool(self.env.query(self.st.Q_STATIC_KERNEL))

    def close(self):
        self.env.close()


def roofline_block(card, n, kernel_ms, note):
    ab = algorithmic_bytes(card)
    achieved = ab * n / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "algorithmic_bytes_per_env_step": ab, "kernel_ms": kernel_ms, "note": note}


def _source_state(entry, kernel_hint):
    """(family, current hash, stale?) of a committed counter entry: stale = it carries no hash, or the hash of another source tree
    than the one this process's library was built from (tools/source_hash.py)."""
    try:
        from tools.source_hash import family_hash, family_of_kernel
        fam = entry.get("kernel_family") or family_of_kernel(kernel_hint)
        cur = family_hash(fam)
    except Exception as ex:      # (a deployment without the sources cannot verify anything: say so instead of vouching)
        return None, None, "cannot hash the kernel sources here: %r" % (ex,)
    have = entry.get("source_hash")
    if have != cur:
        return fam, cur, "entry measured on %s sources %s, this tree is %s" % (fam, have or "(unstamped)", cur)
    return fam, cur, None


def attach_pmc(roof, key):
    """HBM traffic per launch from the committed PMC passes (profiles/pmc_traffic.json, written by tools/update_pmc_traffic.py
    from the rocprofv3 summaries of the same command; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).  The entry is only
    used when its `source_hash` is the hash of the kernel sources of THIS tree: after a kernel edit the line says `stale` and carries
    no traffic figure until the PMC passes have been re-run (tools/profile_round.sh + tools/update_pmc_traffic.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return
    if key in pmc:
        fam, cur, stale = _source_state(pmc[key], pmc[key].get("kernel", ""))
        roof["traffic_source"] = pmc[key].get("source")
        roof["kernel_source_hash"] = {"family": fam, "this_tree": cur, "counters": pmc[key].get("source_hash")}
        if stale:
            roof["traffic"] = None
            roof["stale"] = True
            roof["stale_note"] = "profiles/pmc_traffic.json[%s] not used: %s" % (key, stale)
            return
        roof["stale"] = False
        roof["traffic"] = pmc[key].get("bytes_per_launch")
        if "valu_issue" in pmc[key]:
            roof["valu_issue"] = pmc[key]["valu_issue"]
        if "valu_lanes" in pmc[key]:
            roof["valu_lanes"] = pmc[key]["valu_lanes"]


def attach_valu(roof, env_id, n, dtype, kernel_ms):
    """SURVEY.md 8(d)'s second axis -- the one that can bind: exact flops per env-step (profiles/flops_per_env_step.json, counted by
    tools/count_flops.py on the kernels' own source with a counting scalar type) x env-steps/s of the kernel / vector-ALU peak."""
    try:
        with open(os.path.join(ROOT, "profiles", "flops_per_env_step.json")) as f:
            fl = json.load(f)
    except (OSError, ValueError):
        return
    if env_id not in fl:
        # tree kernel (no host build to count on): the PMC pass's lane-instruction count bounds the flops from above
        lanes = roof.get("valu_lanes")
        if lanes:
            fpe = lanes["flops_upper_bound_per_env_step"]
            achieved = fpe * n / (kernel_ms * 1e-3) / 1e12
            roof["valu"] = {"bound": "valu", "flops_per_env_step": fpe, "achieved": achieved, "peak": VALU_PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                            "frac": achieved / VALU_PEAK_TFLOPS[dtype], "counted": "pmc upper bound: SQ_THREAD_CYCLES_VALU / 4 lane-instructions x 2 flops "
                            "(%s); lanes active per VALU instruction %.2f" % (lanes["source"], lanes["lanes_active_per_valu_instruction"])}
        return
    tree = "kernel" in fl[env_id]
    _, _, stale = _source_state(fl[env_id], "tree kernel" if tree else "lane kernel")
    if stale:
        roof["valu"] = {"stale": True, "stale_note": "profiles/flops_per_env_step.json[%s] not used: %s" % (env_id, stale)}
        return
    fpe = fl[env_id]["flops_per_env_step"]
    achieved = fpe * n / (kernel_ms * 1e-3) / 1e12
    peak = VALU_PEAK_TFLOPS[dtype]     # the tree kernel's count: summed over the 64 lanes of the env's wavefront (tests/kernel_emu/emu_tree_flops.cpp)
    roof["valu"] = {"bound": "valu", "flops_per_env_step": fpe, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "counted": fl[env_id].get("method", fl.get("_method")) if not tree else
                               "tests/kernel_emu/emu_tree_flops.cpp: the tree kernel's own source with a counting scalar on the fiber runtime; " + fl[env_id]["kernel"],
                    "sample": fl[env_id].get("sample"),
                    "note": ("flops the 64 lanes of an env's wavefront execute (fp64 code paths), idle lanes not counted" if tree else
                             "useful flops of one env's lane (pivoting loops end on the lane's own convergence); the wave executes more: "
                             "it iterates until its slowest lane is done -- see valu_issue for the measured issue statistics")}


def host_surface(env_id, n, local_rank, precision, budget_s=1.0, max_steps=200):
    """The drop-in surface itself -- `DartVectorEnv.step(actions)` with numpy arrays in and out, the path north_star describes as
    "observations/rewards are computed on-device and copied back once per batched step" (reference gym/vector/vector_env.py:68-92):
    H2D of the actions, kernel, device auto-reset from the reference-exact MT19937 bank, ONE D2H of obs | reward | done | truncated
    through pinned memory, and the gym.vector return types.  PCIe-inclusive, so never `value`."""
    import dart_env_amd.vector as V
    out = {"api": "DartVectorEnv.step", "noise": "mt19937 (device bank, reference-exact resets)", "envs": n,
           "dtype": "f32" if precision == 32 else "f64"}
    for copy in (True, False):
        venv = V.make(env_id, n, device=local_rank, precision=precision, copy=copy)
        venv.seed(0)
        venv.reset()
        # a RING of action batches, as in the timed region -- NOT one batch applied at every step (rounds 3-4 did that: under constant
        # torques the hoppers spend their episodes lying on several capsules, the step kernel takes 96 us instead of 32 -- rocprof trace of
        # tools/gpu/host_latency_probe.py, profiles/r05_host_path.txt -- and what was reported as the host path's cost was 2/3 kernel)
        ring = np.random.RandomState(0).uniform(-1, 1, (16, n, venv.env.act_dim)).astype(np.float32)
        for i in range(300):      # (the first few hundred steps of a fresh handle run at half speed in a process that also holds
            venv.step(ring[i % 16])   # torch's HIP context -- tools/gpu/host_loop_probe.py: 459 then 249 us per step)
        k, t0 = 0, time.perf_counter()
        while k < max_steps and time.perf_counter() - t0 < budget_s:
            obs, rew, done, info = venv.step(ring[k % 16])
            k += 1
        dt = time.perf_counter() - t0
        out["copy_true" if copy else "copy_false"] = {"copy": copy, "steps": k, "ms_per_step": dt / k * 1e3, "value": n * k / dt,
                                                      "unit": "env-steps/s"}
        venv.close()
    return out


def time_config(env_id, n, local_rank, precision, steps, warmup, all_bodies_collide=None, configure=()):
    # The workload is "random actions"; a ring of resident batches approximates it by a PERIODIC forcing, and a short period changes what the
    # robots do.  Measured (tools/gpu/kernel_time_windows.py, profiles/r06_walker2d_windows.txt; fp64, 65 536 envs, flat over 1 200+ steps from
    # reset): DartWalker2d-v1 122.8 us per step with a ring of 8, 103.8 with 16, 101.9 with 64, 101.4 with 256; DartHopper-v1 32.0-32.4 whatever
    # the ring.  Rounds 1-5 used 16 for the headline and 8 here -- which is why the same Walker2d kernel was quoted at 101.9 (rocprof of the
    # headline path) and 113.5 us (other_configs).  Round 6: 64 everywhere (within 0.5 % of the long-ring limit; 100 MB of actions at most).
    b = HipBenchEnv(env_id, n, local_rank, precision, 0, ring=RING_DEFAULT, all_bodies_collide=all_bodies_collide, configure=configure)
    b.reset()
    b.run(warmup)
    b.sync()
    ms = b.timed_steps(steps)
    card, static = b.card, b.is_static()
    b.close()
    return ms, card, static


def other_solver_block(env_id, n, local_rank, precision, sweeps, steps, warmup, with_oracle):
    """`north_star` names an iterative PGS solver: the kernels' DART_CFG_SOLVER = 1 mode (fixed sweep counts) timed on the headline
    workload, with what a fixed sweep count costs in accuracy -- `rms_vs_exact`: RMS over 4 096 envs x dofs of the velocity difference
    to the pivoting solve after ONE env-step from identical states (worst of 20 env-steps of a random-action rollout) -- and, when the
    oracle may be used (cpu_baseline leg), the same iterate checked against the oracle's PGS at the same sweep count
    (tests/pgs_protocol.py; the gate is tests/test_gpu_pgs_parity.py)."""
    from dart_env_amd import stepper as st
    from dart_env_amd.model_card import card_for
    cfg = [(st.CFG_SOLVER, st.SOLVER_PGS), (st.CFG_ITERS_STAGE1, sweeps), (st.CFG_ITERS_STAGE2, sweeps)]
    ms, card, _ = time_config(env_id, n, local_rank, precision, steps, warmup, configure=cfg)
    out = {"solver": "pgs", "sweeps": sweeps, "value": n / (ms * 1e-3), "unit": "env-steps/s", "kernel_ms": ms, "envs": n,
           "dtype": "f32" if precision == 32 else "f64"}
    ne = 4096
    pgs = st.HipStepper(card, ne, device=local_rank, precision=precision)
    for k, v in cfg:
        pgs.configure(k, v)
    exact = st.HipStepper(card, ne, device=local_rank, precision=precision)
    rng = np.random.RandomState(11)
    for s_ in (pgs, exact):
        s_.configure(st.CFG_SEED, 5)
        s_.reset(None, None, None, want_obs=False)
    worst_dq = worst_q = 0.0
    for t in range(20):
        a = rng.uniform(-1, 1, (ne, card.act_dim)).astype(np.float32)
        q0, dq0 = exact.get_state()
        pgs.set_state(q0, dq0)
        _, _, de, _ = exact.step(a)
        pgs.step(a)
        (qe, dqe), (qp, dqp) = exact.get_state(), pgs.get_state()
        ok = np.isfinite(dqe).all(axis=1) & np.isfinite(dqp).all(axis=1)
        worst_dq = max(worst_dq, float(np.sqrt(np.mean((dqe[ok] - dqp[ok]) ** 2))))
        worst_q = max(worst_q, float(np.sqrt(np.mean((qe[ok] - qp[ok]) ** 2))))
        if de.any():
            exact.reset(de.astype(np.uint8), None, None, want_obs=False)
    pgs.close(); exact.close()
    out["rms_vs_exact"] = {"dq": worst_dq, "q": worst_q, "envs": ne, "protocol": "one env-step from the pivoting solver's state, worst of 20 env-steps"}
    if with_oracle:
        from tests.pgs_protocol import pgs_rollout
        r = pgs_rollout(lambda c, m: st.HipStepper(c, m, device=local_rank, precision=precision), card_for(env_id), 256, 20, sweeps)
        out["vs_oracle_pgs_same_sweeps"] = {"max_abs_dq": max(r["dq"]), "max_abs_q": max(r["q"]), "done_flag_mismatches": r["done_mismatches"],
                                            "envs": 256, "env_steps": 20}
    return out


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here -- the same command line re-executed once per GPU with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets them (rendezvous on 127.0.0.1,
    a free port).  Rank 0's stdout (the ONE JSON line) is this process's stdout; the other ranks' stdout goes to stderr.  Returns the
    worst exit code; a rank that dies takes the others down instead of leaving them in a barrier."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DART_BENCH_SELF_LAUNCHED="1")
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in env:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"           # dmabuf IPC: what RCCL needs on this driver; said once, on stderr
            if r == 0:
                print("bench.py: self-launch sets HSA_ENABLE_IPC_MODE_LEGACY=0 for its ranks (it was unset)", file=sys.stderr)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0:
                rc = rc or c
                for o in live:          # exact PIDs this function started
                    o.terminate()
    return rc


def _resolve(spec):
    """'package.module:attr' -> the object (the hidden --env-factory of the CPU plumbing test)"""
    import importlib
    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr)


def main(argv=None, env_factory=None, dist_backend="nccl"):
    """env_factory / dist_backend exist for the CPU unit test of the N > 1 plumbing (tests/test_bench_plumbing.py injects a
    stand-in shard and gloo); the command line always runs HipBenchEnv over RCCL."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (0: 65536, or 16384 for DartHumanWalker-v1 = BASELINE config 4)")
    ap.add_argument("--all-bodies-collide", type=int, default=-1, help="1 / 0: force every capsule vs feet only (default: the card's default)")
    ap.add_argument("--env-id", default="DartHopper-v1")
    ap.add_argument("--precision", type=int, default=64, choices=[32, 64])
    ap.add_argument("--ring", type=int, default=64, help="distinct action batches resident in HBM, cycled through (round 6: 64, was 16 -- see time_config)")
    ap.add_argument("--spinup-ms", type=float, default=60.0, help="untimed device spin-up on a scratch batch before the measured one is created (0: none)")
    ap.add_argument("--block", type=int, default=0, help="envs per wave64 workgroup (0 = library default)")
    ap.add_argument("--stats", action="store_true", help="print wave-level pivoting iteration histograms")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline, fast_mode parity and other_configs parity")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no steady_state / fast_mode / other_configs")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path (RCCL process group, barriers, max-over-ranks, rollout "
                    "all_gather) even with one rank: the only way to execute it on a 1-GPU box (tests/test_gpu_bench_dist.py)")
    ap.add_argument("--parity-steps", type=int, default=1000)
    ap.add_argument("--parity-budget", type=float, default=12.0, help="seconds of oracle wall time per parity sample")
    ap.add_argument("--dist-backend", default=None, help=argparse.SUPPRESS)    # tests/test_bench_plumbing.py: gloo + a stand-in shard, so that
    ap.add_argument("--env-factory", default=None, help=argparse.SUPPRESS)     # the self-launch path runs on a box without GPUs
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us (torch.distributed.run always sets RANK): start the ranks ourselves.  Another launcher's rank
        # variables without RANK (srun: SLURM_PROCID, mpirun: OMPI_COMM_WORLD_RANK / PMI_RANK, or a bare LOCAL_RANK) mean N processes
        # like this one already exist: each starting N more would put N^2 processes on the GPUs -- refuse and say what to set.
        foreign = [v for v in ("SLURM_PROCID", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "PMIX_RANK", "LOCAL_RANK") if v in os.environ]
        if foreign:
            raise SystemExit("bench.py: --gpus %d under a launcher that set %s but not RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT: "
                             "export those per process (as torch.distributed.run does), or run plain `python bench.py --gpus %d` "
                             "outside the launcher to let it start its own ranks" % (args.gpus, ", ".join(foreign), args.gpus))
        rc = self_launch(args.gpus, argv)
        if rc != 0:
            raise SystemExit("bench.py: a self-launched rank exited with code %d" % rc)
        return None
    if args.dist_backend:
        dist_backend = args.dist_backend
    if args.env_factory:
        env_factory = _resolve(args.env_factory)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with `python -m torch.distributed.run --nproc-per-node %d ...` "
                         "(one rank per GPU)" % (args.gpus, world, args.gpus))
    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
    elif env_factory is None:
        torch.cuda.set_device(local_rank)

    def device_sync():
        if env_factory is None:
            torch.cuda.synchronize()

    abc = None if args.all_bodies_collide < 0 else bool(args.all_bodies_collide)
    n = args.envs or default_envs(args.env_id)
    env_offset, _ = shard_of(rank, n)
    cfg = []
    if env_factory is None:
        from dart_env_amd import stepper as st
        if args.stats:
            cfg.append((st.CFG_STATS, 1))
        if args.block:
            cfg.append((st.CFG_BLOCK_THREADS, args.block))
    make = env_factory or HipBenchEnv
    b = make(args.env_id, n, local_rank, args.precision, env_offset, ring=args.ring, ring_seed=1234 + rank,
             all_bodies_collide=abc, configure=cfg)
    card = b.card
    # Device spin-up, outside the measurement: a SCRATCH batch of the same kind is stepped for ~60 ms so that the timed launches run at
    # the clocks the device settles at.  A window of W + K = 25 launches is under a millisecond -- shorter than the power-management
    # ramp of an idle GPU -- and measured 5 % slower without this (35.3 vs 33.5 us per step) for reasons that are not the kernel's.  The
    # measured batch is created fresh after it: its W warm-up steps and K timed steps are the first steps of its episodes, as before.
    spinup = {"ms": 0.0, "steps": 0}
    if env_factory is None and args.spinup_ms > 0:
        sb = make(args.env_id, n, local_rank, args.precision, env_offset, ring=args.ring, ring_seed=99 + rank, all_bodies_collide=abc, configure=cfg)
        sb.reset()
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.spinup_ms and spinup["steps"] < 20000:
            sb.run(25); sb.sync(); spinup["steps"] += 25
        spinup["ms"] = round((time.perf_counter() - t_s) * 1e3, 1)
        # (the scratch batch is released after the timed region: a hipFree here is a device-wide stall right before the measurement, and
        # 20 ms of idle are enough for the clocks to drop again -- tools/gpu/spin_probe.py: 32.6 us released first, 32.2 kept, 34.7 after 20 ms idle)
    b.reset()
    b.mark(0)                                  # creates the handle's timing events outside the timed region
    b.run(args.warmup)
    b.sync()
    if dist is not None:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    b.mark(0)                                  # HIP events on the kernels' stream around EXACTLY args.steps launches (enqueue only)
    b.run(args.steps, args.warmup)
    b.mark(1)
    if env_factory is not None:
        b.sync()                               # (the stand-in shard of the CPU plumbing test has no device to synchronise)
    if dist is not None:
        dist.barrier()
    device_sync()                              # torch.cuda.synchronize(): waits for every stream of the device, the stepper's included
    elapsed = time.perf_counter() - t0
    ms_kernel = b.elapsed_ms() / args.steps    # the event wait and read-out sit outside the wall-clock region
    if spinup["steps"]:
        sb.close()
    offsets = [env_offset]
    per_rank = [{"rank": rank, "wall_ms_per_step": elapsed / args.steps * 1e3, "kernel_ms_per_step": ms_kernel}]
    if dist is not None:
        local_elapsed = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=b.dev if env_factory is None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        got = [None] * world
        dist.all_gather_object(got, (env_offset, local_elapsed / args.steps * 1e3, ms_kernel))
        offsets = [g[0] for g in got]
        per_rank = [{"rank": r, "wall_ms_per_step": g[1], "kernel_ms_per_step": g[2]} for r, g in enumerate(got)]

    done_frac = b.done_fraction()
    if args.stats and rank == 0 and env_factory is None:
        h1, h2 = b.env.solver_stats()
        print("pivoting iterations per wave, stage 1:", h1.tolist(), file=sys.stderr)
        print("pivoting iterations per wave, stage 2:", h2.tolist(), file=sys.stderr)

    # ---- the north star's "gather rollouts" exchange: one all_gather, outside the timed region
    gather_ms = gather_note = None
    if dist is not None:
        try:
            packed = b.packed_last()
            out = torch.empty((world * packed.numel(),), device=packed.device, dtype=packed.dtype)   # rank-major shards
            dist.all_gather_into_tensor(out, packed)
            device_sync()
            g0 = time.perf_counter()
            for _ in range(10):
                dist.all_gather_into_tensor(out, packed)
            device_sync()
            gather_ms = (time.perf_counter() - g0) / 10 * 1e3
        except Exception as ex:      # the step throughput above does not depend on this exchange: report it, keep the line
            gather_note = "rollout all_gather failed: %r" % (ex,)

    if rank != 0:
        b.close()
        if dist is not None:
            dist.destroy_process_group()
        return None

    dtype = "f32" if args.precision == 32 else "f64"
    total_steps = world * n * args.steps
    result = {
        "metric": "env_steps_per_sec", "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s batch %d per GPU, random actions U[-1,1), on-device auto-reset" % (args.env_id, n),
                   "envs_per_gpu": n, "frame_skip": int(card.frame_skip), "physics_dt": card.dt,
                   "contact_set": describe_contacts(card),
                   "lcp_solver": "two-stage boxed LCP, block principal pivoting (exact)",
                   "parallelism": "env-sharded x%d, no data-path collective" % world, "env_offsets": offsets,
                   "done_fraction_last_step": done_frac},
        "roofline": roofline_block(card, n, ms_kernel, "HIP events on the kernel's stream around the timed region's launches; the "
                                   "kernel is VALU-issue / latency bound, neither HBM nor MFMA bound (DESIGN.md section 5)"),
    }
    if spinup["steps"]:
        result["device_spinup"] = {"untimed_ms": spinup["ms"], "steps_on_a_scratch_batch": spinup["steps"],
                                   "note": "before the measured batch is created; its warm-up and timed steps are the first of its episodes"}
    attach_pmc(result["roofline"], "%s/%d/%s" % (args.env_id, n, dtype))
    attach_valu(result["roofline"], args.env_id, n, dtype, ms_kernel)
    if dist is not None:
        result["per_rank"] = per_rank          # every rank's own clock: `value` uses the slowest (max over ranks, contract)
        result["dist_backend"] = dist.get_backend()
        result["nccl_world_size"] = dist.get_world_size()       # as the process group (RCCL on the GPU box) reports it, not as --gpus says
        result["launcher"] = "self" if os.environ.get("DART_BENCH_SELF_LAUNCHED") else "torch.distributed.run / external"
    if gather_ms is not None:
        result["gather_ms"] = gather_ms
        result["gather_bytes_per_rank"] = int(packed.numel() * packed.element_size())
    if gather_note is not None:
        result["gather_note"] = gather_note
    b_static = b.is_static() if hasattr(b, "is_static") else None
    result["config"]["compile_time_model"] = b_static

    extras = world == 1 and env_factory is None and not args.no_extras
    if extras:
        # the same launch over a longer window, episodes de-synchronised (the driver's timed region is 20 launches right after a reset)
        k = 2000 if args.env_id != "DartHumanWalker-v1" else 40
        ms = b.timed_steps(k)
        result["steady_state"] = {"steps": k, "ms_per_step": ms, "value": n / (ms * 1e-3), "after_steps": args.warmup + args.steps}
    b.close()

    if extras:
        result["host_surface"] = host_surface(args.env_id, n, local_rank, args.precision)
        other_prec = 32 if args.precision == 64 else 64
        do_parity = not args.no_cpu_baseline
        if do_parity:
            from tests import oracle_lib as ol      # cpu_baseline leg: the only place bench.py touches the oracle
            from tests.parity_protocol import ORACLE_RATE, parity_check, parity_sample_size
            cores = ol.usable_cores()
        # ---- headline parity + CPU baseline
        if do_parity:
            ne = parity_sample_size(args.env_id, args.parity_steps, cores, args.parity_budget)
            stats, ref, acts = parity_check(args.env_id, args.precision, ne, args.parity_steps, local_rank, abc)
            # one core alone on a bounded slice of the same sample
            ne1 = max(64, min(ne, int(4.0 * ORACLE_RATE.get(args.env_id, 3e3) / 200) // 64 * 64))
            t1 = time.perf_counter()
            one = ol.rollout(card, acts[:200, :ne1], seed=0, env_offset=0, solver=0)
            one_s = time.perf_counter() - t1
            result["cpu_baseline"] = {
                "value": one["env_steps"] / one_s, "unit": "env-steps/s", "cores": 1, "kind": "port",
                "sample": "%d envs x 200 env-steps of %s on one core (the first envs / steps of the parity sample: %d envs x %d env-steps, "
                          "same Philox reset streams and action tensor as the GPU check); fp64 DART-semantics restatement (oracle/), "
                          "not DART; BASELINE.md fallback (ii)" % (ne1, args.env_id, ne, args.parity_steps),
                "host_cpus": os.cpu_count(),
                "all_cores": {"value": ref["env_steps"] / ref["seconds"], "unit": "env-steps/s", "cores": ref["threads"],
                              "note": "the whole parity sample, one oracle thread per usable core (affinity / cgroup quota)"},
                "rms_state_err": stats,
            }
            if args.env_id in ("DartHopper-v1", "DartWalker2d-v1"):
                from dart_env_amd.model_card import card_for
                alt = not all(card.shape_collidable[s] for s in range(card.nshapes))   # the other contact set
                ref_alt = ol.rollout_trace(card_for(args.env_id, all_bodies_collide=alt), acts, ref["snap_steps"], seed=0, env_offset=0)
                result["cpu_baseline"]["feet_only_contacts_identical"] = bool(
                    np.array_equal(ref_alt["q"], ref["q"]) and np.array_equal(ref_alt["dq"], ref["dq"]) and np.array_equal(ref_alt["done"], ref["done"]))
        # ---- the other precision on the headline workload
        ms_o, _, _ = time_config(args.env_id, n, local_rank, other_prec, max(args.steps, 500), args.warmup, abc)
        mode = {"dtype": "f%d" % other_prec, "value": n / (ms_o * 1e-3), "ms_per_step": ms_o, "kernel_ms": ms_o}
        if do_parity:
            mode["rms_state_err"], _, _ = parity_check(args.env_id, other_prec, ne, args.parity_steps, local_rank, abc, ref=ref, acts=acts)
        result["fast_mode" if other_prec == 32 else "parity_mode"] = mode
        # ---- the solver north_star names, beside the default (exact pivoting): fixed-sweep PGS on the headline workload
        result["other_solver"] = other_solver_block(args.env_id, n, local_rank, args.precision, 30, max(args.steps, 300), args.warmup, do_parity)
        # ---- BASELINE configs 3 and 4
        others = []
        for oid, osteps, owarm in (("DartWalker2d-v1", 500, 100), ("DartHumanWalker-v1", 30, 5)):
            if oid == args.env_id:
                continue
            on = default_envs(oid)
            ms_h, ocard, ostatic = time_config(oid, on, local_rank, args.precision, osteps, owarm)
            ms_f, _, _ = time_config(oid, on, local_rank, other_prec, osteps, owarm)
            entry = {"workload": "%s batch %d, random actions U[-1,1), on-device auto-reset" % (oid, on), "dtype": dtype,
                     "value": on / (ms_h * 1e-3), "ms_per_step": ms_h, "steps": osteps, "warmup": owarm,
                     "contact_set": describe_contacts(ocard), "compile_time_model": ostatic,
                     "roofline": roofline_block(ocard, on, ms_h, "HIP events around %d launches" % osteps),
                     "f%d" % other_prec: {"value": on / (ms_f * 1e-3), "kernel_ms": ms_f}}
            attach_pmc(entry["roofline"], "%s/%d/%s" % (oid, on, dtype))
            attach_valu(entry["roofline"], oid, on, dtype, ms_h)
            if do_parity:
                one_ne = parity_sample_size(oid, args.parity_steps, cores, args.parity_budget, cap=4096)
                entry["rms_state_err"], oref, oacts = parity_check(oid, args.precision, one_ne, args.parity_steps, local_rank)
                entry["cpu_oracle_all_cores"] = {"value": oref["env_steps"] / oref["seconds"], "cores": oref["threads"]}
                entry["f%d" % other_prec]["rms_state_err"], _, _ = parity_check(oid, other_prec, one_ne, args.parity_steps, local_rank,
                                                                                 ref=oref, acts=oacts)
            others.append(entry)
        result["other_configs"] = others
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return result


def describe_contacts(card) -> str:
    nb = sum(1 for s in range(card.nshapes) if card.shape_collidable[s])
    return "%d of %d collision shapes vs ground%s" % (nb, card.nshapes, ", link-link pairs" if card.self_collision else "")


if __name__ == "__main__":
    main()
