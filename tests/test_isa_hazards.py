"""The coneqp kernels issue their operand loads by hand (inline asm `global_load_dwordx4 v, v, s[base]`), and hipcc's hazard
recognizer does not look inside inline asm: a scalar base that was reloaded from a VGPR lane (SGPR spill) or produced by
v_readfirstlane fewer than 5 wait states before such a load is read stale by the hardware (gfx9: "VALU writes SGPR -> VMEM reads
that SGPR"), which showed up on the device as memory faults on address 0.  Compile the kernels to assembly here (no GPU needed)
and check that no hand-issued load sits in that shadow (tools/sgpr_hazard.py)."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _scanner():
    spec = importlib.util.spec_from_file_location("sgpr_hazard", os.path.join(ROOT, "tools", "sgpr_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_sees_a_planted_hazard():
    mod = _scanner()
    planted = """_ZN6hipdrt4testEv:
	v_readlane_b32 s2, v248, 59
	v_readlane_b32 s3, v248, 60
	s_add_u32 s0, s2, 0x400
	global_load_dwordx4 v[4:7], v88, s[2:3]
	s_nop 4
	global_load_dwordx4 v[8:11], v88, s[2:3]
.Lfunc_end0:
"""
    found = mod.scan(planted)
    assert len(found) == 2 and all(f[1].startswith("global_load_dwordx4 v[4:7]") for f in found)


def test_scalar_overwrite_ends_the_hazard_and_a_scalar_copy_does_not_start_one():
    """The hazard is on the REGISTER a VALU instruction wrote: once a scalar instruction has overwritten it the load reads a
    scalar result (no wait states needed); a base computed by scalar arithmetic FROM a v_readlane result is clean as well,
    the VALU-written register itself stays dirty."""
    mod = _scanner()
    txt = """_ZN6hipdrt4testEv:
	v_readlane_b32 s22, v248, 0
	v_readlane_b32 s1, v248, 1
	s_add_u32 s22, s22, s5
	s_addc_u32 s23, s1, s0
	s_add_u32 s0, s22, 16
	global_load_dwordx4 v[122:125], v191, s[22:23] sc1
	s_addc_u32 s1, s23, 0
	global_load_dwordx4 v[126:129], v191, s[0:1] sc1
	v_readlane_b32 s7, v248, 2
	s_mov_b32 s6, s22
	global_load_dwordx4 v[130:133], v191, s[6:7]
.Lfunc_end0:
"""
    found = mod.scan(txt)
    assert [(f[1].split()[1], f[2]) for f in found] == [("v[130:133],", 7)]


def test_scanner_follows_fall_through_and_back_edges():
    mod = _scanner()
    planted = """_ZN6hipdrt5test2Ev:
	v_readlane_b32 s2, v248, 59
	v_readlane_b32 s3, v248, 60
.LBB0_1:
	global_load_dwordx4 v[4:7], v88, s[2:3]
	s_nop 7
	v_readlane_b32 s6, v248, 61
	v_readlane_b32 s7, v248, 62
	s_cbranch_scc1 .LBB0_3
	s_branch .LBB0_4
.LBB0_3:
	global_load_dwordx4 v[8:11], v88, s[6:7]
.LBB0_4:
	s_nop 7
	global_load_dwordx4 v[12:15], v88, s[6:7]
.Lfunc_end0:
"""
    found = mod.scan(planted)
    hit = sorted({f[1].split(',')[0] for f in found})
    # the loop header load sees the reloads of the block in front of it; the branch target sees the reloads at the branch;
    # the load behind eight wait states is clean
    assert hit == ["global_load_dwordx4 v[4:7]", "global_load_dwordx4 v[8:11]"], found


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("extra", [[], ["-DHIPDRT_QP_PROFILE"]], ids=["release", "profile"])
def test_hand_issued_loads_are_outside_the_sgpr_hazard_shadow(tmp_path, extra):
    mod = _scanner()
    src = os.path.join(ROOT, "hybrid-drt_amd", "csrc", "qp.hip")
    out = tmp_path / "qp.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only"] + extra + [src, "-o", str(out)], check=True, cwd=os.path.dirname(src),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    txt = out.read_text()
    assert "global_load_dwordx4" in txt and "qp_kernel_resident" in txt
    found = mod.scan(txt)
    assert not found, "hand-issued loads behind a VALU-written scalar base:\n" + "\n".join(
        "%s | %s | s%d %d wait states" % (n[:50], l, s_, age) for n, l, s_, age in found[:10])


def _instructions(body):
    return [ln.strip() for ln in body.split("\n")
            if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.strip().endswith(":")]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_group_barrier_drains_vector_stores_before_the_arrival(tmp_path):
    """qp_group.hpp group_sync(): the members of a group sit on different CUs, so every wavefront has to wait for its own vector
    stores (`s_waitcnt vmcnt(0)`) in front of the workgroup barrier that precedes the member's arrival on the group counter --
    a workgroup-scope `__syncthreads()` alone is `s_waitcnt lgkmcnt(0); s_barrier` on gfx950.  Checked on the assembly: every
    arrival (the agent-scope add on gsync word 2) has a barrier in front of it, and between that barrier and the `vmcnt(0)` wait before
    it there is neither a vector memory instruction nor a branch."""
    src = os.path.join(ROOT, "hybrid-drt_amd", "csrc", "qp.hip")
    out = tmp_path / "qp.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", str(out)], check=True, cwd=os.path.dirname(src),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    txt = out.read_text()
    start = txt.index("_ZN6hipdrt15qp_kernel_groupENS_6QpArgsEii:")
    ins = _instructions(txt[start:txt.index(".Lfunc_end", start)])
    arrivals = [i for i, s_ in enumerate(ins) if s_.startswith("global_atomic_add") and s_.endswith("offset:8")]
    assert len(arrivals) >= 2, "the group arrival was not found in the assembly (gsync word 2 = byte offset 8)"
    for i in arrivals:
        b = max(j for j in range(i) if ins[j].startswith("s_barrier"))
        # back from the barrier to the wait: nothing in between may touch vector memory (hipcc moves scalar bookkeeping there)
        j = b - 1
        while j >= 0 and not (ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]):
            assert not ins[j].startswith(("global_", "buffer_", "flat_", "scratch_")), (ins[j], "between the wait and the barrier")
            assert not ins[j].startswith(("s_barrier", "s_cbranch", "s_branch")), ("no vmcnt(0) in front of the group barrier", ins[j - 3:b + 1])
            j -= 1
        assert j >= 0
