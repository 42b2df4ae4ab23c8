"""Build-time check of the hand-scheduled kernels (the round-6 fault): a register load written as inline asm is
waited for by a hand-written ``s_waitcnt vmcnt(n)``; the compiler believes its destination registers are defined (or, if nobody
reads them, FREE) the moment the statement is passed.  ``vamb_amd/csrc/isa_pending_loads.py`` walks the ISA of the objects that
ship and reports every instruction that touches a register a load is still writing.  Round 6: the split-K instantiations of the
deep-prefetch fp32 GEMM computed an output row ABOVE their final wait in a register of a stage still in flight -- a wild store,
a memory fault in ~2 % of 500-step fp32 runs at the C2 shape.  No GPU needed: this reads the compiler's output."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vamb_amd", "csrc"))
import build as pb  # noqa: E402
import isa_pending_loads as ipl  # noqa: E402

GOOD = """
_Zgood:
	;;#ASMSTART
	global_load_dwordx4 v[2:5], v[8:9], off
	;;#ASMEND
	v_add_u32_e32 v10, s3, v11
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_u32_e32 v2, s3, v10
	global_store_dword v[6:7], v2, off
	s_endpgm
"""
# the pattern of the round-6 fault: the value of the load is dead, the compiler reuses v2 above the wait
BAD = GOOD.replace("\tv_add_u32_e32 v10, s3, v11\n", "\tv_add_u32_e32 v2, s3, v11\n")
# loads retire in order: vmcnt(1) retires the first of two, not the second
TWO = """
_Ztwo:
	;;#ASMSTART
	global_load_dwordx4 v[2:5], v[8:9], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[12:15], v[8:9], off
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_mov_b32_e32 v20, v2
	v_mov_b32_e32 v21, REG
	s_endpgm
"""


def _hazards(text):
    lines = text.split("\n")
    (_, body), = list(ipl.kernels(lines))
    msgs = []
    return ipl.check_kernel(body, out=msgs.append), msgs


def test_checker_accepts_a_register_used_after_the_wait():
    assert _hazards(GOOD)[0] == 0


def test_checker_flags_a_dead_destination_reused_above_the_wait():
    n, msgs = _hazards(BAD)
    assert n == 1 and "WRITES ['v2']" in msgs[0]


def test_checker_retires_loads_in_order():
    assert _hazards(TWO.replace("REG", "v3"))[0] == 0
    n, msgs = _hazards(TWO.replace("REG", "v13"))
    assert n == 1 and "READS ['v13']" in msgs[0]


def test_compiler_managed_loads_are_not_tracked():
    text = GOOD.replace(";;#ASMSTART\n\tglobal_load", "\tglobal_load").replace("off\n\t;;#ASMEND", "off")
    assert "ASMSTART\n\tglobal_load" not in text
    assert _hazards(BAD.replace(";;#ASMSTART\n\tglobal_load_dwordx4 v[2:5], v[8:9], off\n\t;;#ASMEND", "\tglobal_load_dwordx4 v[2:5], v[8:9], off"))[0] == 0


def test_shipped_objects_have_no_pending_load_hazard():
    """the ISA of exactly the objects in libvambhip.so (build.py compiles the two sources with -save-temps and keeps the device .s)"""
    def stale(src):
        path = pb.isa_path(src)
        return not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(pb.HERE, src)) - 1

    if any(stale(s) for s in pb.ISA_CHECKED):   # (no dump yet, or sources touched since: the incremental build regenerates them)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "vamb_amd", "csrc", "build.py")])
    seen = 0
    for src in pb.ISA_CHECKED:
        path = pb.isa_path(src)
        assert not stale(src), f"{path} is older than {src} after a build"
        msgs = []
        n_k, n_asm, hazards = ipl.check_file(path, out=msgs.append)
        assert hazards == 0, "\n".join(msgs)
        seen += n_asm
    assert seen >= 20   # the deep-prefetch / K-group GEMM instantiations and the two row-major scan kernels were really looked at
