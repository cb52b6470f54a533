"""Static checks of the BUILT product library (no GPU): register allocation of every gfx950 kernel in kirag_amd/libkirag_amd.so.

A spilled vector register is a scratch store / reload — a VMEM operation in the same in-order queue as the LDS-DMA rings, which hipcc follows with
`s_waitcnt vmcnt(0)`: in round 3's library that drained the ring in the V^T tiles of the QKV projection and in front of the DMA of k_attn_dma's
partial chunk.  The library must carry none."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
LIB = os.path.join(REPO, "kirag_amd", "libkirag_amd.so")


@pytest.fixture(scope="module")
def notes():
    import kernel_notes
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    if not os.path.exists(kernel_notes.READELF):
        pytest.skip("llvm-readelf not installed")
    ks = kernel_notes.kernels(LIB)
    assert len(ks) > 100, "expected the search + two encoder code objects"
    return ks


def test_no_product_kernel_spills_vector_registers(notes):
    bad = [(k["name"], k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0)) for k in notes
           if k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)]
    assert not bad, "kernels with scratch memory: %r" % bad


def test_hot_kernels_keep_their_occupancy(notes):
    by = {k["name"]: k for k in notes}
    def find(*parts):
        hits = [k for n, k in by.items() if all(p in n for p in parts)]
        assert hits, parts
        return hits
    # ping-pong GEMMs and the coarse scan: two waves per SIMD (<= 256 registers each); the short-sequence attention: three blocks per CU (<= 168)
    for k in find("k_projILi", "GemmShapeILi256ELi256") + find("k_coarseINS", "ELb0EEEv"):
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256, k
    for k in find("k_attn_ldsILi"):
        assert k["vgpr_count"] <= 168, k
    for k in find("k_attn_dma"):
        assert k["vgpr_count"] <= 256, k


def test_no_kernel_uses_v_ashr_pk_u8_i32():
    """`v_ashr_pk_u8_i32` (shift two ints right, saturate each to a byte, pack) writes only bits 15:0 of its destination on gfx950, while hipcc's
    selection pattern (shift -> clamp to [0, 255] -> pack two neighbours) assumes a zero upper half and ORs the stale bits into whatever is packed
    next to it (tools/ashr_pk_check.hip reproduces it: every result differs in bits 31:16).  Round 4 met it in the residual stream's byte codec
    (encoder.hip: lo_encode clamps BEFORE the shift for that reason); no product kernel may contain the instruction."""
    import kernel_notes
    if not os.path.exists(kernel_notes.OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    assert kernel_notes.count_instruction(LIB, "v_ashr_pk_u8_i32") == 0
