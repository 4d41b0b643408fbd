#!/usr/bin/env python3
"""Tabulate the compiler's per-kernel resource remarks (hipcc -Rpass-analysis=kernel-resource-usage, written by
__graft_entry__.build() to build/obj/<unit>.res.txt): VGPRs, AGPRs, SGPRs, scratch, LDS, waves per SIMD.
    python tools/kernel_resources.py [substring-filter]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.res.txt"))):
        cur = None
        for line in open(path):
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(?:Function Name|Name): (\S+)", line)   # (with -save-temps the location follows "remark:")
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?([\w][\w \[\]/]*?):\s+(\S+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = [r["name"] for r in rows]
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        dem = names
    print("%-110s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("dartk::", "")
        if flt and flt not in d:
            continue
        print("%-110s %5s %5s %5s %7s %6s %4s" % (d[:110], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                 r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"),
                                                 r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
