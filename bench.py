#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched HIP stepper on BASELINE.json's headline config.

A "step" is one batched env.step() of DartHopper-v1 over `--envs` (default 65 536) environments per GPU:
clamp/scale action, frame_skip=4 world steps (dynamics + contact/limit LCP + integration), reward, done,
TimeLimit, observation and on-device auto-reset -- one kernel launch.  Inputs (a ring of random action
batches, U[-1,1) float32) are resident in HBM before the timed region; outputs stay in HBM.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 2000 --warmup 200

Multi-GPU: envs are independent, so ranks own disjoint env shards (Philox streams keyed by global env index) and
there is NO collective inside the timed region; one RCCL all_gather of the last step's obs/reward/done runs after it
(the "gather rollouts" exchange of the north star) and is reported separately as gather_ms.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      HBM roofline of the step kernel from HIP-event timing of the same launches
  cpu_baseline  (N=1 only) the fp64 CPU oracle timed on one host core on a bounded sample, plus the RMS state
                error of the GPU against it on that sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def algorithmic_bytes(card) -> int:
    """read q,dq (8n) + actions (4 act) + write q,dq (8n) + obs (4 obs) + reward 4 + done 1."""
    return 16 * card.ndofs + 4 * card.act_dim + 4 * card.obs_dim + 4 + 1


# algorithmic flops per env-step (Featherstone operation counts F(n, m, K) x frame_skip); the kernels are
# bound by the fp32 vector ALU (157.3 TFLOP/s with packed fp32), neither by HBM nor by MFMA -- reported beside the HBM figure
ALGORITHMIC_FLOPS = {"DartHopper-v1": 2.2e4, "DartWalker2d-v1": 5.3e4, "DartHumanWalker-v1": 3.3e6}
VALU_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (0: 65536, or 16384 for DartHumanWalker-v1 = BASELINE config 4)")
    ap.add_argument("--all-bodies-collide", action="store_true",
                    help="every collision shape vs the floor (DART's behaviour) instead of the feet-only default cards of Hopper / Walker2d")
    ap.add_argument("--env-id", default="DartHopper-v1")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--solver", default="bpp", choices=["bpp", "pgs"])
    ap.add_argument("--pgs-iters", type=int, default=30)
    ap.add_argument("--ring", type=int, default=16, help="distinct action batches resident in HBM")
    ap.add_argument("--block", type=int, default=0, help="envs per wave64 workgroup (0 = library default)")
    ap.add_argument("--stats", action="store_true", help="print wave-level pivoting iteration histograms")
    ap.add_argument("--iters", type=int, default=0, help="pivoting iteration cap per stage (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=200)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dart_env_amd.model_card import card_for
    from dart_env_amd import stepper as st

    card = card_for(args.env_id, all_bodies_collide=args.all_bodies_collide)
    n = args.envs or (16384 if args.env_id == "DartHumanWalker-v1" else 65536)
    env = st.HipStepper(card, n, device=local_rank, precision=args.precision)
    env.configure(st.CFG_AUTORESET, 1)
    env.configure(st.CFG_SEED, 0)
    env.configure(st.CFG_ENV_OFFSET, rank * n)
    if args.stats:
        env.configure(st.CFG_STATS, 1)
    if args.iters:
        env.configure(st.CFG_ITERS_STAGE1, args.iters); env.configure(st.CFG_ITERS_STAGE2, args.iters)
    if args.block:
        env.configure(st.CFG_BLOCK_THREADS, args.block)
    if args.solver == "pgs":
        env.configure(st.CFG_SOLVER, st.SOLVER_PGS)
        env.configure(st.CFG_ITERS_STAGE1, args.pgs_iters)
        env.configure(st.CFG_ITERS_STAGE2, args.pgs_iters)

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    ring = (torch.rand((args.ring, n, card.act_dim), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
    obs = torch.empty((n, card.obs_dim), device=dev, dtype=torch.float32)
    rew = torch.empty((n,), device=dev, dtype=torch.float32)
    done = torch.empty((n,), device=dev, dtype=torch.uint8)
    trunc = torch.empty((n,), device=dev, dtype=torch.uint8)
    stride = n * card.act_dim * 4
    env.reset_device(0, obs.data_ptr())
    env.sync()

    def run(k, base=0):
        for i in range(k):
            env.step_device(ring.data_ptr() + ((base + i) % args.ring) * stride, obs.data_ptr(), rew.data_ptr(),
                            done.data_ptr(), trunc.data_ptr())

    run(args.warmup)
    env.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    env.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    done_frac = float(done.float().mean().item())
    if args.stats and rank == 0:
        h1, h2 = env.solver_stats()
        print("pivoting iterations per wave, stage 1:", h1.tolist(), file=sys.stderr)
        print("pivoting iterations per wave, stage 2:", h2.tolist(), file=sys.stderr)

    # ---- kernel-only timing with HIP events on the stepper's own stream (roofline figure)
    ms_kernel = env.time_steps(ring.data_ptr(), args.ring, min(args.steps, 500), obs.data_ptr(), rew.data_ptr(),
                               done.data_ptr(), trunc.data_ptr())

    # ---- the north star's "gather rollouts" exchange: one RCCL all_gather, outside the timed region
    gather_ms = None
    gather_note = None
    if dist is not None:
        try:
            packed = torch.cat([obs.reshape(-1), rew, done.float()]).contiguous()
            out = torch.empty((world, packed.numel()), device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(out, packed)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dist.all_gather_into_tensor(out, packed)
            e1.record(); torch.cuda.synchronize()
            gather_ms = e0.elapsed_time(e1) / 10
        except Exception as ex:      # the step throughput above does not depend on this exchange: report it, keep the line
            gather_note = "rollout all_gather failed: %r" % (ex,)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    abytes = algorithmic_bytes(card)
    total_steps = world * n * args.steps
    value = total_steps / elapsed
    achieved = abytes * n / (ms_kernel * 1e-3) / 1e9
    result = {
        "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
        "config": {"workload": "%s batch %d per GPU, random actions U[-1,1), on-device auto-reset%s"
                               % (args.env_id, n, ", every capsule collides" if args.all_bodies_collide else ""),
                   "envs_per_gpu": n, "frame_skip": int(card.frame_skip), "physics_dt": card.dt,
                   "lcp_solver": "two-stage boxed LCP, %s" % ("block principal pivoting (exact)" if args.solver == "bpp"
                                                               else "PGS x%d" % args.pgs_iters),
                   "parallelism": "env-sharded x%d, no data-path collective" % world,
                   "done_fraction_last_step": done_frac},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": None, "algorithmic_bytes_per_env_step": abytes, "kernel_ms": ms_kernel,
                     "note": "kernel is VALU-issue bound (fp32 vector), not HBM or MFMA bound; see DESIGN.md"},
    }
    if args.env_id in ALGORITHMIC_FLOPS and args.precision == 32:
        tf = ALGORITHMIC_FLOPS[args.env_id] * n / (ms_kernel * 1e-3) / 1e12
        result["roofline"]["valu"] = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                                      "algorithmic_flops_per_env_step": ALGORITHMIC_FLOPS[args.env_id]}
    if gather_ms is not None:
        result["gather_ms"] = gather_ms
    if gather_note is not None:
        result["gather_note"] = gather_note
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs;
    # FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction) -- only valid for the profiled configuration
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        key = "%s/%d/f%d%s" % (args.env_id, n, args.precision, "/allcaps" if args.all_bodies_collide else "")
        if key in pmc:
            result["roofline"]["traffic"] = pmc[key]["bytes_per_launch"]
            result["roofline"]["traffic_source"] = pmc[key]["source"]
            if "valu_issue" in pmc[key] and "valu" in result["roofline"]:
                result["roofline"]["valu"]["issue"] = pmc[key]["valu_issue"]
    except OSError:
        pass

    if world == 1 and not args.no_cpu_baseline:
        from tests import oracle_lib as ol  # cpu_baseline leg: the only place bench.py touches the oracle
        ne, ns = args.cpu_envs, args.cpu_steps
        acts = np.random.RandomState(7).uniform(-1, 1, (ns, ne, card.act_dim)).astype(np.float32)
        t0 = time.perf_counter()
        ref = ol.rollout(card, acts, seed=0, env_offset=0, solver=0)
        cpu_s = time.perf_counter() - t0
        small = st.HipStepper(card, ne, device=local_rank, precision=args.precision)
        small.configure(st.CFG_AUTORESET, 1); small.configure(st.CFG_SEED, 0); small.configure(st.CFG_ENV_OFFSET, 0)
        d_acts = torch.from_numpy(acts).to(dev)
        small.reset_device(0, 0)
        for t in range(ns):
            small.step_device(d_acts.data_ptr() + t * ne * card.act_dim * 4)
        small.sync()
        qg, dqg = small.get_state()
        el, ep = small.counters()
        same = (ep == ref["episode"]) & (el == ref["elapsed"])
        eq = (qg - ref["q"])[same]; edq = (dqg - ref["dq"])[same]
        close = np.abs(eq).max(axis=1) < 1e-4   # envs that did not take a contact event a substep early/late
        result["cpu_baseline"] = {
            "value": ref["env_steps"] / cpu_s, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d env-steps of %s, same Philox reset streams and action tensor as the GPU check; "
                      "fp64 DART-semantics restatement (oracle/), not DART" % (ne, ns, args.env_id),
            "host_cpus": os.cpu_count(),
            "rms_state_err": {"q": float(np.sqrt(np.mean(eq ** 2))), "dq": float(np.sqrt(np.mean(edq ** 2))),
                              "q_trimmed": float(np.sqrt(np.mean(eq[close] ** 2))),
                              "dq_trimmed": float(np.sqrt(np.mean(edq[close] ** 2))),
                              "median_abs_q": float(np.median(np.abs(eq))), "median_abs_dq": float(np.median(np.abs(edq))),
                              "envs_within_1e-4": int(close.sum()),
                              "envs_same_episode_history": int(same.sum()), "envs": ne, "env_steps": ns},
        }
        small.close()
        # the same sample spread over the host's cores, one oracle process per core ("1 core, then P
        # processes"): compute time only, slowest worker; bounded, and never allowed to lose the line
        try:
            import subprocess
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:      # a container may be capped below its visible cores (cgroup v2 cpu.max = "<quota> <period>" or "max <period>")
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    avail = min(avail, max(1, int(float(quota) / float(period) + 0.5)))
            except (OSError, ValueError):
                try:  # cgroup v1
                    quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if quota > 0:
                        avail = min(avail, max(1, int(quota / period + 0.5)))
                except (OSError, ValueError):
                    pass
            P = max(1, min(avail, 64, ne))
            per = ne // P
            procs = [subprocess.Popen([sys.executable, "-m", "tests.oracle_worker", args.env_id, str(int(args.all_bodies_collide)),
                                       str(i * per), str(per), str(ns), str(card.act_dim), str(ne)],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(P)]
            tot, slowest = 0, 0.0
            deadline = time.time() + 120
            for pr in procs:
                out, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                a, b = out.split()
                tot += int(a); slowest = max(slowest, float(b))
            result["cpu_baseline"]["all_cores"] = {"value": tot / slowest, "unit": "env-steps/s", "cores": P,
                                                   "note": "one oracle process per usable core (affinity / cgroup quota, at most 64) "
                                                           "on %d of the %d envs each; compute time of the slowest process" % (per, ne)}
        except Exception as ex:
            for pr in locals().get("procs", []):
                if pr.poll() is None:
                    pr.kill()
            result["cpu_baseline"]["all_cores"] = {"value": None, "note": "not measured: %r" % (ex,)}
        if args.env_id in ("DartHopper-v1", "DartWalker2d-v1"):
            # The default cards of these two envs test only the feet against the floor (BASELINE config[1]); rerun the
            # CPU sample with EVERY capsule collidable, as DART has it: identical final states = the deviation is
            # never exercised by this workload (a second ~7 s of host time, same sample).
            ref_all = ol.rollout(card_for(args.env_id, all_bodies_collide=True), acts, seed=0, env_offset=0, solver=0)
            result["cpu_baseline"]["all_capsule_contacts_identical"] = bool(
                np.array_equal(ref_all["q"], ref["q"]) and np.array_equal(ref_all["dq"], ref["dq"])
                and np.array_equal(ref_all["episode"], ref["episode"]))
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
