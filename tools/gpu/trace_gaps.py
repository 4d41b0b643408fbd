#!/usr/bin/env python3
"""Per-dispatch durations and inter-dispatch gaps of the step kernel from a rocprofv3 --kernel-trace run (rocpd sqlite output):
   python tools/gpu/trace_gaps.py <trace_dir> [kernel-name substring]
Prints, per distinct grid size, count / median duration / median gap to the previous dispatch of the same kernel -- the split of a
host-synchronous step into "kernel" and "everything between two kernels" (VERDICT r4 item 5-i)."""
import glob, os, sqlite3, sys
import numpy as np

prof = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
dbs = sorted(glob.glob(os.path.join(prof, "**", "*.db"), recursive=True))
if not dbs:
    sys.exit("no rocpd database under " + prof)
for db in dbs:
    con = sqlite3.connect(db); cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = None
    for t in names:
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % t)]
        low = [c.lower() for c in cols]
        if "start" in low and "end" in low and any(c in low for c in ("name", "kernel_name")):
            cand = (t, cols); 
            if t.lower() in ("kernels", "top_kernels"): break
    if cand is None:
        print(db, "tables:", names); continue
    t, cols = cand
    ncol = "name" if "name" in [c.lower() for c in cols] else "kernel_name"
    extra = [c for c in cols if c.lower() in ("grid_size", "grid_size_x", "grid_x", "workgroup_size", "queue_id", "stream_id")]
    rows = list(cur.execute("select %s, start, end %s from %s order by start" % (ncol, "".join(", " + c for c in extra), t)))
    rows = [r for r in rows if pat in str(r[0])]
    print("%s: table %s, %d dispatches of *%s*; extra columns %s" % (os.path.basename(db), t, len(rows), pat, extra))
    groups = {}
    for r in rows:
        groups.setdefault((str(r[0])[:60],) + tuple(r[3:]), []).append((r[1], r[2]))
    for k, v in groups.items():
        v = np.array(v, dtype=np.float64)
        dur = (v[:, 1] - v[:, 0]) / 1e3
        gap = (v[1:, 0] - v[:-1, 1]) / 1e3
        print("  %s n=%d  duration us: median %.1f p10 %.1f p90 %.1f | gap to previous us: median %.1f p10 %.1f p90 %.1f" % (
            k, len(v), np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90),
            np.median(gap) if len(gap) else -1, np.percentile(gap, 10) if len(gap) else -1, np.percentile(gap, 90) if len(gap) else -1))
