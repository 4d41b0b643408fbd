"""What do the pivoting iterations beyond the first cost the headline kernel?  Times DartHopper-v1 x 65 536 (fp64 / fp32) with the
iteration caps of the two LCP stages (DART_CFG_ITERS_STAGE1 / 2) lowered: a cap of 1 is what a wave of perfectly homogeneous lanes
would pay (every lane's first solve) -- the UPPER BOUND of what binning envs to lanes can buy.  A capped run keeps a clamped,
unconverged iterate in the lanes that needed more, so its trajectories are NOT the product's: a timing experiment only.
usage (GPU box): python tools/gpu/hopper_iter_ablation.py [env-id]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import time_config
from dart_env_amd import stepper as st
env_id = sys.argv[1] if len(sys.argv) > 1 else "DartHopper-v1"
n = int(os.environ.get("N", "65536"))
steps = int(os.environ.get("STEPS", "2000"))
for prec in (64, 32):
    for caps in (None, (1, 1), (2, 1), (1, 2), (2, 2), (3, 3), None):
        cfg = [] if caps is None else [(st.CFG_ITERS_STAGE1, caps[0]), (st.CFG_ITERS_STAGE2, caps[1])]
        ms, card, static = time_config(env_id, n, 0, prec, steps, 200, configure=cfg)
        print("%s f%d caps %-8s  %.3f us per step" % (env_id, prec, "default" if caps is None else "%d/%d" % caps, ms * 1e3), flush=True)
