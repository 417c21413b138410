"""The build's device code is free of VGPR spills in the shadow of a partial EXEC mask.

Round 6 traced the wrong models of a sibling form of cd_gramr_kernel<10,3> to that: the compiler
placed a spill store of the batch header's row record in a join block BEFORE the `s_or_b64 exec`
that restores the full mask (scripts/isa_lint.py).  Whether it does depends on
register allocation, so the check runs on every build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import isa_lint  # noqa: E402

BAD = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], s[14:15]
	s_cbranch_execz .LBB1_71
	ds_read_b32 v164, v164
	s_waitcnt lgkmcnt(0)
	ds_write_b32 v163, v164
.LBB1_71:
	s_mov_b32 s26, s40
	v_writelane_b32 v255, s51, 43
	s_waitcnt vmcnt(0)
	scratch_store_dwordx4 off, v[178:181], off ; 16-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	s_waitcnt lgkmcnt(0)
	s_barrier
"""
GOOD = BAD.replace("""	s_waitcnt vmcnt(0)
	scratch_store_dwordx4 off, v[178:181], off ; 16-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
""", """	s_or_b64 exec, exec, s[0:1]
	s_waitcnt vmcnt(0)
	scratch_store_dwordx4 off, v[178:181], off ; 16-byte Folded Spill
""")
# a labeled block INSIDE a region (reached by a loop branch) may reload what the region uses
INSIDE = """
_Z6kernelv:
	s_and_saveexec_b64 s[4:5], s[68:69]
	s_cbranch_execz .LBB5_19
.LBB5_38:
	scratch_load_dword v2, off, off offset:116 ; 4-byte Folded Reload
	v_cmp_ne_u32_e32 vcc, s96, v2
	s_or_b64 exec, exec, s[4:5]
	s_cbranch_vccnz .LBB5_38
.LBB5_19:
	s_or_b64 exec, exec, s[4:5]
	s_endpgm
"""


def test_lint_flags_the_spill_before_the_exec_restore():
    f = isa_lint.lint_text(BAD)
    assert len(f) == 1 and f[0][0] == "_Z6kernelv" and f[0][3] == ".LBB1_71"
    assert "scratch_store_dwordx4" in f[0][2]
    assert isa_lint.lint_text(GOOD) == []
    assert isa_lint.lint_text(INSIDE) == []


def test_device_code_of_every_translation_unit_passes_the_lint(tmp_path):
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(os.path.join(isa_lint.CSRC, f) for f in os.listdir(isa_lint.CSRC) if f.endswith(".hip"))
    assert len(srcs) >= 13
    with ThreadPoolExecutor(max_workers=8) as ex:
        asm = list(ex.map(lambda s: isa_lint.build_asm(s, str(tmp_path)), srcs))
    findings = {os.path.basename(a): isa_lint.lint_file(a) for a in asm}
    assert all(not v for v in findings.values()), findings
    # the item-space kernels are in what was checked
    text = open([a for a in asm if "gramr_k13" in a][0]).read()
    assert "cd_gramr_kernelILi10ELi3E" in text
