"""Build-time check of the kernels that keep loads in flight across barriers in NAMED physical registers (csrc/inflight_regs.h,
conv1d_bsplit.hip, conv1d_bsplit2.hip): no compiler-generated instruction of those kernels may touch the reserved registers --
otherwise a value would be overwritten by, or read before, a load that has not landed.  Runs hipcc (gfx950 cross-compile), no
GPU needed."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("src,n_kernels", [("conv1d_bsplit.hip", 3), ("conv1d_bsplit2.hip", 3)])
def test_named_landing_registers_are_left_alone(src, n_kernels):
    import check_inflight_regs as C
    asm = C.compile_to_asm(os.path.join(REPO, "facodec_amd", "csrc", src))
    res = C.reserved_violations(asm)
    assert len(res) == n_kernels, sorted(res)          # every wide k = 7 / 5 / 3 kernel, every (K1, stride) of the 32-row kernel
    for name, (lo, bad) in res.items():
        assert lo in (176, 208), (name, lo)
    # Exact form: between the inline-asm load that writes a landing register and the inline-asm v_cndmask that takes the value out,
    # nothing else may touch it -- on any path of the control-flow graph.  (Since the k = 7 kernel walks several tiles per workgroup
    # with the next tile's inputs in flight across the epilogue, the MFMA waves' and the P8 staging waves' instantiations of the tile
    # walk do use v208.. for their own values; they are not reachable from a load site.)
    live = C.named_lifetime_violations(asm)
    assert set(live) == set(res)
    for name, bad in live.items():
        assert not bad, (name, bad[:5])
    if src == "conv1d_bsplit2.hip":                    # one role per kernel there: the whole reserved range stays untouched
        for name, (lo, bad) in res.items():
            assert not bad, (name, bad[:5])
    # no kernel of the file may spill more than a handful of registers to scratch (round 4: an innocent-looking unrolled DMA loop made
    # the 32 / 48-channel shape of the k = 7 kernel spill 443)
    spills = C.spill_counts(asm)
    assert spills and max(spills.values()) <= 32, spills


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_split_resident_lstm_keeps_its_state_pieces_in_flight_safely():
    """lstm_fwd_persist_split_kernel requests the h pieces of four MFMA steps ahead with inline-asm loads into compiler-allocated
    registers.  The code between request and `s_waitcnt` is straight-line, so the dataflow replay of the vmcnt queue is exact there:
    no instruction may touch a destination register before the wait that covers it; and the 12-step kernel (144 weight registers)
    must not spill."""
    import check_inflight_regs as C
    asm = C.compile_to_asm(os.path.join(REPO, "facodec_amd", "csrc", "lstm_persist.hip"))
    report, bad = C.check(asm)
    split = {k: v for k, v in report.items() if "lstm_fwd_persist_split_kernel" in k}
    assert len(split) == 3, sorted(report)
    assert sorted(v["asm_loads"] >= n for v, n in zip(sorted(split.values(), key=lambda v: v["asm_loads"]), (12, 24, 36))) == [True] * 3   # 3 planes x NK steps
    assert not [b for b in bad if "lstm_fwd_persist_split_kernel" in b[0]], bad[:5]
    spills = C.spill_counts(asm)
    assert all(n == 0 for k, n in spills.items() if "lstm_fwd_persist_split_kernel" in k), spills
