# round 5, session 12: the in-tree library with the four-env wave solver: hand-off statistics (timing build), the whole GPU suite, smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s12; mkdir -p $O
cd $R
DART_STEPPER_LIB=$R/abtest/lib_c4f.so python bench.py --no-extras --env-id DartHalfCheetah-v1 --precision 64 --envs 65536 --steps 100 --warmup 20 --stats 2>&1 | grep "stage 1" | python -c "
import sys, json, re
a = json.loads(re.search(r'(\[.*\])', sys.stdin.read()).group(1))
n = max(a[28], 1)
print('passes %d (120 launches x 1024 waves): cycles per pass: rows + Y + A %.0f, stage 1 %.0f, stage 2 %.0f, velocity update %.0f; rows per pass (all groups) %.1f' % (a[28], a[24]/n, a[25]/n, a[26]/n, a[27]/n, a[29]/n))
print('hand-offs from the small tier: %d envs (%.2f per wave and env-step), %.0f cycles each' % (a[30], a[30] / (120.0 * 1024), a[31] / max(a[30], 1)))
" | tee $O/timing_c4f.txt
python tools/gpu/cheetah_coop4_probe.py tree -1 2>&1 | grep -v Warning | grep Dart | tee $O/probe.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/tests.txt
