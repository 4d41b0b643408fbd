#!/usr/bin/env python3
"""Read-before-write of the private segment, statically: for one function of a gfx950 assembly file, a forward
"definitely written on every path" dataflow over its basic blocks for the accumulator registers (AGPR spills), the VGPR lanes that hold spilled SGPRs (v_writelane / v_readlane with a constant lane) and the frame bytes addressed absolutely (scratch_* off, off offset:N --
spill slots and by-value aggregates), and a report of every scratch_load that can execute before some byte it reads has been stored.
A hit is either a compiler bug (a spill reload without its spill on that path) or an uninitialised local that reached memory.

    python tools/scratch_dataflow.py <file.s> <mangled-name-prefix>          (assembly: tools/disasm_report.py keeps it under /tmp)

Limits, stated: a callee's frame is addressed through s32 / s33, taken as constant inside the function; stores / loads through a VGPR address (callees writing through pointers into this frame) are treated as writing
nothing and reading nothing; a call (s_swappc_b64) is assumed not to touch the caller's absolute slots."""
import re, sys
from collections import defaultdict

W = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "ubyte": 1, "sbyte": 1, "byte": 1, "ushort": 2, "sshort": 2, "short": 2,
     "ubyte_d16": 1, "ubyte_d16_hi": 1, "short_d16": 2, "short_d16_hi": 2, "byte_d16_hi": 1}


def main():
    lines = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if re.match(r"^" + re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []; order.append(cur)
    for i in range(start + 1, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            cur = m.group(1); blocks[cur] = []; order.append(cur); continue
        t = lines[i].split(";")[0].strip()
        if t and not t.startswith("."):
            blocks[cur].append((i - start, t))
    succ = defaultdict(set)
    for k, name in enumerate(order):
        ins = blocks[name]
        nxt = order[k + 1] if k + 1 < len(order) else None
        falls = True
        pending_label = None
        for pos, t in ins:
            op = t.split()[0]
            m = re.search(r"\(?(\.LBB\d+_\d+)-\.Lpost_getpc", t)
            if m:
                pending_label = m.group(1)
            if op == "s_branch":
                succ[name].add(t.split()[1]); falls = False
            elif op.startswith("s_cbranch"):
                succ[name].add(t.split()[1])
            elif op == "s_setpc_b64":
                if pending_label:
                    succ[name].add(pending_label); pending_label = None; falls = False
                else:
                    falls = False      # return
            elif op == "s_endpgm":
                falls = False
        if falls and nxt:
            succ[name].add(nxt)
    pred = defaultdict(set)
    for a, bs in succ.items():
        for b in bs:
            pred[b].add(a)

    def areg_set(tok):
        m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"a(\d+)", tok)
        return [int(m.group(1))] if m else []

    def accesses(t):
        """-> list of (kind, first byte, width, modelled): frame bytes are >= 0, accumulator register aN is the 4 bytes at -4 (N + 1)"""
        out = []
        op = t.split()[0]
        toks = [x.strip() for x in re.sub(r"offset:\d+|op_sel\S*|bitop3:\S+|glc|slc|sc[01]|nt", "", t[len(op):]).split(",")]
        toks = [x.split()[0] if x.split() else "" for x in toks]
        # accumulator registers: the first operand of a load / v_accvgpr_write / any VALU op is the destination, every other mention a source
        is_store = op.startswith(("scratch_store", "global_store", "flat_store", "ds_write", "buffer_store"))
        for k, tok in enumerate(toks):
            for n in areg_set(tok):
                out.append((("store" if (k == 0 and not is_store) else "load"), -4 * (n + 1), 4, True))
        # SGPR spills into VGPR lanes: v_writelane_b32 vN, sX, L defines slot (N, L), v_readlane_b32 sX, vN, L reads it (constant lanes only)
        m = re.match(r"v_(writelane|readlane)_b32\s+(\S+),\s*(\S+),\s*(\d+)\s*$", t)
        if m:
            vreg = (m.group(2) if m.group(1) == "writelane" else m.group(3)).strip(",")
            if re.fullmatch(r"v\d+", vreg):
                out.append((("store" if m.group(1) == "writelane" else "load"), -4 * (100000 + int(vreg[1:]) * 64 + int(m.group(4))), 4, True))
        m = re.match(r"scratch_(load|store)_(\w+)\s+(.*)", t)
        if m:
            kind, w, rest = m.group(1), W.get(m.group(2), 4), m.group(3)
            off = re.search(r"offset:(\d+)", rest)
            off = int(off.group(1)) if off else 0
            ops = [o.strip() for o in re.sub(r"offset:\d+", "", rest).split(",")]
            va, sa = (ops[0], ops[2].split()[0]) if kind == "store" else (ops[1], ops[2].split()[0])
            absolute = va == "off" and (sa == "off" or re.fullmatch(r"s3[23]", sa) is not None)   # frame slots: no base, or the callee's SP / FP
            if absolute and sa != "off":
                off += (1 << 20) * (int(sa[1:]) - 31)      # keep the s32- and s33-relative slots apart from each other
            out.append((kind, off, w, absolute))
        # reads first, then writes (an instruction may read and write the same register)
        return sorted(out, key=lambda a: a[0] != "load")

    universe = set()
    for name in order:
        for pos, t in blocks[name]:
            for a in accesses(t):
                if a[3] and a[0] == "store":
                    universe.update(range(a[1], a[1] + a[2]))
    OUT = {n: set(universe) for n in order}
    OUT["entry"] = set()
    changed = True
    while changed:
        changed = False
        for n in order:
            inn = set(universe) if pred[n] else set()
            for p in pred[n]:
                inn &= OUT[p]
            if n == "entry":
                inn = set()
            cur = set(inn)
            for pos, t in blocks[n]:
                for a in accesses(t):
                    if a[3] and a[0] == "store":
                        cur.update(range(a[1], a[1] + a[2]))
            if cur != OUT[n]:
                OUT[n] = cur; changed = True
    hits, nload, ndyn = [], 0, 0
    for n in order:
        inn = set(universe) if pred[n] else set()
        for p in pred[n]:
            inn &= OUT[p]
        if n == "entry":
            inn = set()
        cur = set(inn)
        for pos, t in blocks[n]:
            for a in accesses(t):
                if not a[3]:
                    ndyn += 1; continue
                if a[0] == "store":
                    cur.update(range(a[1], a[1] + a[2]))
                else:
                    nload += 1
                    missing = [b for b in range(a[1], a[1] + a[2]) if b not in cur and (b in universe or b < 0)]
                    if missing:
                        hits.append((n, pos, t, len(missing)))
    print("%s: %d blocks, %d absolute scratch loads checked, %d register-addressed scratch ops not modelled" % (pat[:60], len(order), nload, ndyn))
    print("loads that can run before all of their bytes were stored on every path: %d" % len(hits))
    for h in hits[:60]:
        print("   block %-12s +%-6d %s   (%d bytes undefined on some path)" % h)


if __name__ == "__main__":
    main()
