"""tools/vmcnt_audit.py (the build-time audit of counted vector-memory waits): its queue walk over the
control-flow graph on hand-written instruction streams — what it must flag, what it must not, and that a path through a
skipped branch is covered."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vmcnt_audit", os.path.join(ROOT, "tools", "vmcnt_audit.py"))
A = importlib.util.module_from_spec(spec)
spec.loader.exec_module(A)


def _walk(text):
    lines = text.strip("\n").split("\n")
    found = []
    A.walk(lines, 0, len(lines), found)
    return [(n, loads) for _, n, loads in found]


def test_round3_failure_shape_is_flagged():
    # a 12-byte load through a 64-bit VGPR address among 16-byte loads through an SGPR base, retired by a counted wait
    asm = """
	global_load_dwordx4 v[0:3], v20, s[4:5]
	global_load_dwordx3 v[4:6], v[22:23], off
	global_load_dwordx4 v[8:11], v21, s[4:5]
	s_waitcnt vmcnt(1)
	v_add_f32 v0, v0, v4
	s_endpgm
"""
    got = _walk(asm)
    assert got == [(1, ("dwordx3/vaddr64", "dwordx4/saddr"))], got
    assert not A.form_only(got[0][1])


def test_one_class_and_full_waits_are_clean():
    asm = """
	global_load_dwordx4 v[0:3], v20, s[4:5]
	global_load_dwordx4 v[4:7], v21, s[4:5]
	s_waitcnt vmcnt(1)
	v_add_f32 v0, v0, v1
	global_load_dword v8, v[22:23], off
	s_waitcnt vmcnt(0)
	s_endpgm
"""
    assert _walk(asm) == []


def test_stores_count_but_do_not_make_a_mix():
    asm = """
	global_load_dword v0, v20, s[4:5]
	global_store_dwordx4 v21, v[4:7], s[6:7]
	global_load_dword v1, v22, s[4:5]
	s_waitcnt vmcnt(1)
	s_endpgm
"""
    assert _walk(asm) == []


def test_loads_left_in_flight_by_a_skipped_branch_reach_the_next_loop_iteration():
    # the epilogue shape round 4 found: the waits sit inside a predicated region; with the region skipped the dword loads
    # are still outstanding at the LDS-DMA ring's counted wait of the next work item
    asm = """
.LBB0_1:
	global_load_lds_dwordx4 v[6:7], off
	global_load_lds_dwordx4 v[6:7], off
	s_waitcnt vmcnt(1)
	s_barrier
	global_load_dword v0, v20, s[4:5]
	s_cbranch_execz .LBB0_3
	s_waitcnt vmcnt(0)
	global_store_dword v21, v0, s[6:7]
.LBB0_3:
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""
    got = _walk(asm)
    assert (1, ("dword/saddr", "dwordx4/vaddr64/lds")) in got, got
    # ... and with the wait in front of the branch nothing outlives the region
    fixed = asm.replace("\ts_cbranch_execz .LBB0_3\n\ts_waitcnt vmcnt(0)\n", "\ts_waitcnt vmcnt(0)\n\ts_cbranch_execz .LBB0_3\n")
    assert _walk(fixed) == []


def test_address_form_only_mixes_are_listed_as_such():
    asm = """
	global_load_dword v0, v20, s[4:5]
	global_load_dword v1, v[22:23], off
	global_load_dword v2, v21, s[4:5]
	s_waitcnt vmcnt(1)
	s_endpgm
"""
    got = _walk(asm)
    assert got and all(A.form_only(loads) for _, loads in got), got


def test_committed_audit_has_no_open_kernel():
    txt = open(os.path.join(ROOT, "profiles", "r04_vmcnt_audit.txt")).read()
    assert "# kernels with unjustified mixed counted waits: 0" in txt
    assert "  MIXED" not in txt
