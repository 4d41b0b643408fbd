#!/bin/bash
# Where do a wave's cycles go?  Runs bench.py --stats against a library built with -DDART_WAVE_TIMING (tools/build_variant.sh) and
# decodes the 64 counters (planar_kernel.hpp: wave_timing_add).   usage: tools/gpu/wave_timing.sh <variant> <env-id> [precisions]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; ENV=$2; PRECS=${3:-"32 64"}
export DART_STEPPER_LIB=$R/abtest/lib_$V.so
for p in $PRECS; do
  python $R/bench.py --no-extras --env-id $ENV --precision $p --envs 65536 --steps 100 --warmup 20 --stats 2>&1 | python -c "
import sys, json, re
h = {}
for l in sys.stdin:
    m = re.match(r'pivoting iterations per wave, stage (\d): (\[.*\])', l)
    if m: h[int(m.group(1))] = json.loads(m.group(2))
    if l.startswith('{'): ms = json.loads(l)['ms_per_step']
a, b = h[1], h[2]
n = a[2]
print('$ENV f$p  %.3f ms/step;  waves %d: mean %.0f cycles, max %d;  per wave: fallback %.0f, big tier %.0f, small tier %.0f' % (ms, n, a[0]/n, a[1], a[3]/n, a[4]/n, a[5]/n))
print('  per wave: passes of the four-env solver %.2f, envs served one at a time %.2f' % (a[6]/n, a[7]/n))
print('  waves by log2(cycles) 14..21+:', a[8:16])
print('  fallback invocations by log2(cycles) 8..23:', a[16:32])
print('  big-tier invocations:', b[0:16])
print('  small-tier invocations:', b[16:32])
"
done
