#!/usr/bin/env python3
"""Turn rocprofv3 rocpd sqlite output (gpurun_out/prof/*/hopper_results.db) into the text summaries kept under
profiles/.  usage: summarize_rocprof.py <prof_dir> <out_prefix>"""
import glob
import os
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:90]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    lines = []
    dbs = sorted(glob.glob(os.path.join(prof, "*", "*_results.db")) + glob.glob(os.path.join(prof, "*", "*", "*_results.db")))
    for db in dbs:
        rel = os.path.relpath(db, prof).split(os.sep)
        tag = rel[0] if len(rel) > 2 else os.path.basename(os.path.dirname(db))   # <run>/<host>/<pid>_results.db -> <run>
        con = sqlite3.connect(db)
        cur = con.cursor()
        lines.append("## %s" % tag)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        except Exception:
            rows = []
        if rows and "trace" in tag:
            lines.append("kernel-trace --stats (durations in us)")
            lines.append("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for n, c, t, a, p in rows:
                lines.append("%-90s %8d %14.2f %12.3f %8.3f" % (short(n), c, t, a, p))
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                 "group by kernel_name, counter_name")
            rows = list(cur.execute(q))
        except Exception:
            rows = []
        if rows:
            lines.append("PMC counters (average per dispatch)")
            lines.append("%-60s %-28s %8s %18s" % ("kernel", "counter", "disp", "avg/dispatch"))
            for n, c, k, a in rows:
                if "dartk" in n:
                    lines.append("%-60s %-28s %8d %18.3f" % (short(n)[:60], c, k, a))
        lines.append("")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
