#!/bin/bash
# round 6, GPU session 11: are the compacted-limit tier and the all-limits tier bitwise the same on the device?  c6 = the half cheetah with one row per limited joint
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s11; mkdir -p $O
cd $R
for env in DartHalfCheetah-v1; do
  for p in 64 32; do
    ENV_ID=$env N=8192 STEPS=12 PREC=$p python tools/gpu/ab_states.py $O/base_$p.npz > /dev/null 2>&1
    ENV_ID=$env N=8192 STEPS=12 PREC=$p DART_STEPPER_LIB=$R/abtest/lib_c6.so python tools/gpu/ab_states.py $O/c6_$p.npz > /dev/null 2>&1
    python - <<PY
import numpy as np
a = np.load("$O/base_$p.npz"); b = np.load("$O/c6_$p.npz")
for t in range(a["q"].shape[0]):
    dq = np.abs(a["q"][t] - b["q"][t]); bad = (dq > 0).any(axis=1)
    print("$env f$p step %d: envs whose q differs between the compacted and the all-limits build: %d of %d, max |dq| %.3e" % (t, bad.sum(), len(bad), dq.max()))
    if bad.any() and t < 3: print("   first differing envs:", np.nonzero(bad)[0][:8])
PY
  done
done 2>&1 | tee $O/tier_bits.txt
rm -f $O/*.npz
