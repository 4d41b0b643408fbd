#!/usr/bin/env python3
"""Lint gfx950 assembly for vector instructions that sit BEFORE the EXEC restore at the top of a control-flow join block.

Round 5's root cause of "the first launch differs from the later ones" (profiles/r05_first_launch.txt): in one build of the half
cheetah's fp32 kernel the register allocator's VGPR -> AGPR spill copies (five v_accvgpr_write_b32) were placed at the very top of a join
block, ahead of the `s_or_b64 exec, exec, s[a:b]` that re-enables the lanes which skipped the divergent region -- so they ran for the lanes
that took the `if` only (for none at all when the branch was skipped), and the reloads further down, with every lane on again, read whatever
the accumulator registers held before the launch.  The pattern is mechanical, so it can be looked for in every kernel that ships:

    python tools/exec_prologue_lint.py <file.s> [function-name substring]       (assembly: tools/disasm_report.py keeps it under /tmp)
    python tools/exec_prologue_lint.py --all                                    (compiles the five units of the product library, ~6 min)

A hit = a label, at most WINDOW instructions, an `s_or_b64 exec, exec, ...` -- and a vector / memory spill-class instruction (v_accvgpr_write / scratch_store) between the label and it.
Exit status 1 when any function of the file has a hit."""
import os, re, subprocess, sys, tempfile

WINDOW = 12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# spill-class instructions: a register saved to an accumulator register or to the private segment.  (Other vector instructions ahead of the
# restore are common and legitimate -- v_readlane of the spilled EXEC mask itself, v_cmp building the next mask -- `-v` lists those too.)
VEC = re.compile(r"^(v_accvgpr_write|v_accvgpr_mov|scratch_store)")
VEC_ALL = re.compile(r"^(v_|scratch_|global_|flat_|buffer_|ds_)")


def lint(path, filt="", verbose=False):
    lines = open(path).read().split("\n")
    hits, func, nfunc = [], None, 0
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z[A-Za-z0-9_]*):", l)
        if m:
            func = m.group(1); nfunc += 1
        elif l.startswith(".Lfunc_end"):
            func = None
        elif func and (not filt or filt in func) and re.match(r"^\.LBB\d+_\d+:", l):
            label = l.split(":")[0]
            seen, k, j = [], 0, i + 1
            while j < len(lines) and k < WINDOW:
                t = lines[j].split(";")[0].strip()
                j += 1
                if not t or t.startswith("."):
                    if re.match(r"^\.LBB\d+_\d+:", t):
                        break
                    continue
                k += 1
                if re.match(r"s_or_b64\s+exec,\s*exec,", t):
                    if seen:
                        hits.append((func, label, i + 1, seen))
                    break
                if t.split()[0].startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
                    break
                if (VEC_ALL if verbose else VEC).match(t) and not t.startswith(("v_readlane", "v_writelane", "v_cmp", "v_readfirstlane")):
                    seen.append(t)
        i += 1
    return hits, nfunc


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--all":
        sys.path.insert(0, ROOT)
        from __graft_entry__ import UNIT_FLAGS, UNITS
        total = 0
        for u in UNITS:
            out = os.path.join(tempfile.gettempdir(), "lint_%s.s" % u)
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "--cuda-device-only", "-S",
                   os.path.join(ROOT, "dart_env_amd", "csrc", u + ".hip"), "-o", out] + UNIT_FLAGS.get(u, []) + os.environ.get("DART_EXTRA_HIPFLAGS", "").split()
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            hits, nf = lint(out)
            print("%-12s %4d functions, %d join blocks with vector instructions ahead of their EXEC restore" % (u, nf, len(hits)))
            for f, lab, ln, seen in hits[:20]:
                d = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()[:110]
                print("    %s %s (line %d): %s" % (d, lab, ln, "; ".join(seen[:3]) + (" ..." if len(seen) > 3 else "")))
            total += len(hits)
        sys.exit(1 if total else 0)
    verbose = "-v" in sys.argv
    args = [a for a in sys.argv[1:] if a != "-v"]
    hits, nf = lint(args[0], args[1] if len(args) > 1 else "", verbose)
    print("%s: %d functions, %d join blocks with vector instructions ahead of their EXEC restore" % (os.path.basename(sys.argv[1]), nf, len(hits)))
    for f, lab, ln, seen in hits:
        d = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()[:120]
        print("  %s\n      %s (line %d): %s" % (d, lab, ln, "; ".join(seen)))
    sys.exit(1 if hits else 0)


if __name__ == "__main__":
    main()
