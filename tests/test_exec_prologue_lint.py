"""tools/exec_prologue_lint.py on hand-written assembly: the pattern of round 5's toolchain defect -- spill copies between
a join block's label and the `s_or_b64 exec, exec, ...` that re-enables the lanes which skipped the divergent region -- must be flagged, the
same copies AFTER the restore, or ahead of a restore that is not at the top of its block, must not.  (The real instances: profiles/r05_first_launch.txt;
`__graft_entry__.build()` runs the lint over every unit's device assembly and fails on a hit.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

BAD = """
_ZN5dartk4testEv:
	s_and_saveexec_b64 s[36:37], s[12:13]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v0, 0
.LBB0_2:
	v_accvgpr_write_b32 a28, v67
	v_accvgpr_write_b32 a2, v0
	s_mov_b64 s[4:5], s[44:45]
	s_or_b64 exec, exec, s[36:37]
	v_accvgpr_read_b32 v0, a2
	s_endpgm
.Lfunc_end0:
"""
GOOD = BAD.replace("""	v_accvgpr_write_b32 a28, v67
	v_accvgpr_write_b32 a2, v0
	s_mov_b64 s[4:5], s[44:45]
	s_or_b64 exec, exec, s[36:37]
""", """	s_mov_b64 s[4:5], s[44:45]
	s_or_b64 exec, exec, s[36:37]
	v_accvgpr_write_b32 a28, v67
	v_accvgpr_write_b32 a2, v0
""")
SCRATCH = BAD.replace("v_accvgpr_write_b32 a28, v67", "scratch_store_dword off, v67, off offset:16")
# a restore deep inside a block (an inner region's end) with ordinary code in front of it is not a join-block prologue
FAR = BAD.replace(".LBB0_2:\n", ".LBB0_2:\n" + "".join("\ts_add_u32 s%d, s%d, 1\n" % (k, k) for k in range(20)))
# v_readlane of the spilled EXEC mask and v_cmp building the next mask are what legitimately sits there
LEGIT = BAD.replace("""	v_accvgpr_write_b32 a28, v67
	v_accvgpr_write_b32 a2, v0
""", """	v_readlane_b32 s36, v255, 3
	v_readlane_b32 s37, v255, 4
	v_cmp_lt_f32_e32 vcc, v1, v2
""")


def _hits(text, tmp_path, name):
    from exec_prologue_lint import lint
    p = tmp_path / (name + ".s")
    p.write_text(text)
    hits, nf = lint(str(p))
    assert nf == 1
    return hits


def test_spill_copies_ahead_of_the_exec_restore_are_flagged(tmp_path):
    h = _hits(BAD, tmp_path, "bad")
    assert len(h) == 1 and h[0][1] == ".LBB0_2" and [i.split()[0] for i in h[0][3]] == ["v_accvgpr_write_b32", "v_accvgpr_write_b32"]
    h = _hits(SCRATCH, tmp_path, "scratch")
    assert len(h) == 1 and h[0][3][0].startswith("scratch_store_dword")


def test_the_same_copies_behind_the_restore_and_legitimate_prologues_are_not(tmp_path):
    assert _hits(GOOD, tmp_path, "good") == []
    assert _hits(FAR, tmp_path, "far") == []
    assert _hits(LEGIT, tmp_path, "legit") == []


def test_the_build_records_a_clean_lint_for_every_unit():
    """build() writes build/obj/<unit>.lint.txt next to every object it compiles (and raises on a hit): the shipped library was linted"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.lint.txt")))
    if not files:
        import pytest
        pytest.skip("no build directory here (the library was built elsewhere)")
    assert {os.path.basename(f).split(".")[0] for f in files} >= {"planar_f32", "planar_f64", "spatial_f32", "spatial_f64"}
    for f in files:
        first = open(f).read().splitlines()[0]
        assert first.rstrip().endswith(" 0 join blocks with spill-class instructions ahead of their EXEC restore"), first
