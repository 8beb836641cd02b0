"""scripts/lint_lane_masks.py on two hand-written ISA snippets: the hazard it exists for (a VALU compare
inside an EXEC-narrowing loop feeding a uniform branch after the loop) is flagged, the harmless forms
(scalar select, compare outside the loop) are not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "scripts", "lint_lane_masks.py")

BAD = """
hdk_bad:                               ; @hdk_bad
\ts_mov_b64 s[12:13], 0
.LBB0_1:                                ; =>This Loop Header: Depth=1
\tv_cndmask_b32_e64 v4, 0, 1, s[28:29]
\tv_cmp_ne_u32_e64 s[8:9], 1, v4
\tds_read_b32 v6, v8
\tv_cmp_ne_u64_e32 vcc, -2, v[4:5]
\ts_or_b64 s[12:13], vcc, s[12:13]
\ts_andn2_b64 exec, exec, s[12:13]
\ts_cbranch_execnz .LBB0_1
.LBB0_2:
\ts_or_b64 exec, exec, s[12:13]
\ts_and_b64 vcc, exec, s[8:9]
\ts_cbranch_vccnz .LBB0_3
.LBB0_3:
\ts_endpgm
.Lfunc_end0:
"""

GOOD = """
hdk_good:                              ; @hdk_good
\ts_cmp_lg_u32 s8, 0
\ts_cselect_b64 s[8:9], -1, 0
\tv_cmp_ne_u32_e64 s[10:11], 1, v4
.LBB1_1:                                ; =>This Loop Header: Depth=1
\tds_read_b32 v6, v8
\tv_cmp_ne_u64_e32 vcc, -2, v[4:5]
\ts_or_b64 s[12:13], vcc, s[12:13]
\ts_andn2_b64 exec, exec, s[12:13]
\ts_cbranch_execnz .LBB1_1
.LBB1_2:
\ts_or_b64 exec, exec, s[12:13]
\ts_and_b64 vcc, exec, s[8:9]
\ts_cbranch_vccnz .LBB1_3
\ts_and_b64 vcc, exec, s[10:11]
\ts_cbranch_vccnz .LBB1_3
.LBB1_3:
\ts_endpgm
.Lfunc_end1:
"""

# a block-uniform tile loop (scalar back edge) around a divergent row loop: the compare sits in the OUTER loop's own
# lines, where EXEC is what the loop was entered with
OUTER = """
hdk_outer:                             ; @hdk_outer
.LBB2_1:                                ; =>This Loop Header: Depth=1
                                        ;     Child Loop BB2_2 Depth 2
\tv_cmp_ne_u32_e64 s[8:9], 1, v4
.LBB2_2:                                ;   Parent Loop BB2_1 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
\tds_read_b32 v6, v8
\tv_cmp_ne_u64_e32 vcc, -2, v[4:5]
\ts_or_b64 s[12:13], vcc, s[12:13]
\ts_andn2_b64 exec, exec, s[12:13]
\ts_cbranch_execnz .LBB2_2
.LBB2_3:                                ;   in Loop: Header=BB2_1 Depth=1
\ts_or_b64 exec, exec, s[12:13]
\ts_add_u32 s4, s4, 1
\ts_cmp_lt_u32 s4, s5
\ts_cbranch_scc1 .LBB2_1
.LBB2_4:
\ts_and_b64 vcc, exec, s[8:9]
\ts_cbranch_vccnz .LBB2_5
.LBB2_5:
\ts_endpgm
.Lfunc_end2:
"""


def _lint(tmp_path, text):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, LINT, str(f)], capture_output=True, text=True, check=True).stdout


def test_flags_mask_defined_in_a_narrowing_loop_and_used_after_it(tmp_path):
    out = _lint(tmp_path, BAD)
    assert "hdk_bad" in out and "s[8:9]" in out and ".LBB0_1" in out


def test_scalar_selects_and_compares_outside_the_loop_are_clean(tmp_path):
    assert _lint(tmp_path, GOOD).strip() == ""


def test_compare_in_a_uniform_loop_around_a_narrowing_one_is_clean(tmp_path):
    assert _lint(tmp_path, OUTER).strip() == ""
    # ... and the same compare moved INTO the row loop is the hazard
    moved = OUTER.replace("\tv_cmp_ne_u32_e64 s[8:9], 1, v4\n", "").replace("\tds_read_b32 v6, v8\n", "\tv_cmp_ne_u32_e64 s[8:9], 1, v4\n\tds_read_b32 v6, v8\n")
    assert "hdk_outer" in _lint(tmp_path, moved)
