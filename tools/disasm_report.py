#!/usr/bin/env python3
"""What round 4 read out of the disassembly, as a tool: compile one translation unit of the product library to gfx950 assembly and report,
per kernel / device function,
  * v_writelane / v_readlane counts (SGPR values parked in VGPR lanes: the signature of hoisted v_readlane broadcasts),
  * scratch_ and v_accvgpr_ instructions (VGPR spills to memory / to the AGPR half of the register file),
  * flat_load / flat_store (LDS reached through generic pointers inside noinline functions),
  * references to __const.* tables (constexpr functions the compiler evaluates at run time),
  * loop headers by depth.
usage: python tools/disasm_report.py planar_f64|planar_f32|spatial_f64|spatial_f32|dart_stepper [substring of a mangled name ...]
No GPU needed (hipcc cross-compiles); ~1-3 minutes per unit."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    unit, filters = sys.argv[1], sys.argv[2:]
    out = os.path.join(tempfile.gettempdir(), "dart_%s.s" % unit)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "dart_env_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
           os.path.join(ROOT, "dart_env_amd", "csrc", unit + ".hip"), "-o", out] + os.environ.get("DART_EXTRA_HIPFLAGS", "").split()
    sys.path.insert(0, ROOT)
    from __graft_entry__ import UNIT_FLAGS      # the product build's per-unit flags (tree kernels: no machine LICM)
    if os.environ.get("DART_NO_UNIT_FLAGS") != "1":
        cmd += UNIT_FLAGS.get(unit, [])
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    stats, name = {}, None
    for line in open(out):
        m = re.match(r"^(_Z[A-Za-z0-9_]*):", line)
        if m:
            name = m.group(1); stats[name] = dict(lines=0, writelane=0, readlane=0, scratch=0, accvgpr=0, flat=0, const_tables=0, loops={})
            continue
        if line.startswith(".Lfunc_end"):
            name = None; continue
        if name is None:
            continue
        st = stats[name]
        t = line.strip()
        if "Loop Header: Depth=" in line:
            d = int(line.rsplit("Depth=", 1)[1]); st["loops"][d] = st["loops"].get(d, 0) + 1
        if not t or t[0] in ";.":
            continue
        st["lines"] += 1
        op = t.split()[0]
        st["writelane"] += op == "v_writelane_b32"; st["readlane"] += op == "v_readlane_b32"
        st["scratch"] += op.startswith("scratch_"); st["accvgpr"] += "accvgpr" in op; st["flat"] += op.startswith("flat_")
        st["const_tables"] += "__const." in t
    print("%-7s %-6s %-6s %-7s %-7s %-5s %-6s %-14s %s" % ("instr", "wlane", "rlane", "scratch", "accvgpr", "flat", "tables", "loops/depth", "function"))
    for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["lines"]):
        if st["lines"] < 200 or (filters and not any(f in k for f in filters)):
            continue
        demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        print("%-7d %-6d %-6d %-7d %-7d %-5d %-6d %-14s %s" % (st["lines"], st["writelane"], st["readlane"], st["scratch"], st["accvgpr"], st["flat"],
                                                          st["const_tables"], ",".join("%d:%d" % kv for kv in sorted(st["loops"].items())), demangled[:150]))
    print("assembly kept at", out)


if __name__ == "__main__":
    main()
