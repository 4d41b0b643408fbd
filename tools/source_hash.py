#!/usr/bin/env python3
"""Hash of the kernel sources a measurement belongs to.

    python tools/source_hash.py            -> {"planar": "...", "spatial": "...", "dart_stepper": "..."} (JSON)

Every counter file under profiles/ that bench.py copies numbers from (pmc_traffic.json, flops_per_env_step.json) carries the hash of
the kernel family it was measured on; bench.py recomputes the hash from the tree it runs in and reports `"stale": true` (dropping the
copied number) on a mismatch, so a kernel edit cannot leave an old HBM-traffic or flop figure standing in the bench line.
The file sets are the translation units' dependency lists of __graft_entry__.py (UNIT_DEPS): exactly what is compiled into the library.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def family_files(family):
    import __graft_entry__ as g
    units = {"planar": ["planar_f32", "planar_f64"], "spatial": ["spatial_f32", "spatial_f64"], "dart_stepper": ["dart_stepper"]}[family]
    files = set()
    for u in units:
        files.update(g._unit_sources(u))
    return sorted(files)


def family_hash(family):
    import __graft_entry__ as g
    h = hashlib.sha256()
    units = {"planar": ["planar_f32", "planar_f64"], "spatial": ["spatial_f32", "spatial_f64"], "dart_stepper": ["dart_stepper"]}[family]
    h.update(" ".join(" ".join(g.UNIT_FLAGS.get(u, [])) for u in units).encode())     # the per-unit compiler flags are part of what was measured
    for path in family_files(family):
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def all_hashes():
    return {fam: family_hash(fam) for fam in ("planar", "spatial", "dart_stepper")}


def family_of_kernel(kernel_name):
    """profile sections name the step kernel: dartk::sp_step_kernel<...> is the tree kernel, everything else a lane kernel"""
    return "spatial" if "sp_step_kernel" in kernel_name or "tree kernel" in kernel_name else "planar"


if __name__ == "__main__":
    print(json.dumps(all_hashes()))
