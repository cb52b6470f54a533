"""Kernel-independent membership check for full-size searches (test infrastructure).

The library's answer is compared with a formulation that shares NOTHING with it: chunked fp32 ``q @ x.T`` by torch on the device
(rocBLAS / hipBLASLt sgemm — the arithmetic faiss.IndexFlatIP itself uses) and ``torch.topk`` with a running merge over row chunks.
fp32 sgemm differs from the canonical score (exact inner product rounded once) by up to ~d * 2^-24 * sum|q_i x_i| ~ 2e-7 on unit vectors,
so the comparison is:
  * sorted scores agree position by position within ``tol`` (2e-6)
  * every reference row that is clearly above the cut (ref score > our k-th score + tol) is in our result          -> no MISSED row
  * every row we return is in the reference's top (k + margin) list, or the reference's margin is exhausted by
    near-ties at the cut (then the row's own sgemm score, recomputed, must be >= our k-th score - tol)             -> no WRONG row
"""
import numpy as np
import torch


@torch.no_grad()
def torch_topk_fp32(q: torch.Tensor, row_chunks, k: int):
    """row_chunks: iterable of (first_row, fp32 device tensor [m, d]).  -> (scores [nq, k] desc, rows int64 [nq, k]) on the device."""
    best_s = best_i = None
    for r0, x in row_chunks:
        s = q @ x.T                                              # [nq, m] fp32 sgemm
        kk = min(k, s.shape[1])
        cs, ci = torch.topk(s, kk, dim=1)
        ci = ci + r0
        del s
        if best_s is None:
            best_s, best_i = cs, ci
        else:
            ms = torch.cat([best_s, cs], 1); mi = torch.cat([best_i, ci], 1)
            kk = min(k, ms.shape[1])
            best_s, sel = torch.topk(ms, kk, dim=1)
            best_i = torch.gather(mi, 1, sel)
    return best_s, best_i


def check_membership(s, i, ref_s, ref_i, k: int, tol: float = 2e-6, rescore=None):
    """s, i: library result [nq, k] (numpy).  ref_s, ref_i: reference top-(k + margin) (numpy, desc).
    rescore(qi, rows) -> fp32 sgemm-class scores of explicit rows, used only when the margin is exhausted by near-ties.
    Returns a dict of counts; raises AssertionError with the first offending query on a real mismatch."""
    nq = s.shape[0]
    assert ref_s.shape[1] >= k
    np.testing.assert_allclose(s, ref_s[:, :k], rtol=0, atol=tol)             # sorted scores, position by position
    swapped = 0
    for qi in range(nq):
        ours = set(i[qi].tolist())
        assert len(ours) == k, (qi, "duplicate rows in the result")
        cut = float(s[qi, k - 1])
        clear = ref_i[qi][ref_s[qi] > cut + tol]                              # rows clearly above the cut
        missed = [int(r) for r in clear.tolist() if int(r) not in ours]
        assert not missed, (qi, "missed rows", missed[:5], cut)
        ref_all = set(ref_i[qi].tolist())
        extra = [r for r in ours if r not in ref_all]
        if extra:
            # allowed only if the reference's margin ends inside the near-tie band of the cut
            assert float(ref_s[qi, -1]) >= cut - tol, (qi, "rows outside the reference's top list", extra[:5], cut, float(ref_s[qi, -1]))
            if rescore is not None:
                rs = rescore(qi, np.array(extra, np.int64))
                assert (rs >= cut - tol).all(), (qi, "returned rows score below the cut", extra[:5])
        swapped += int(not np.array_equal(i[qi], ref_i[qi, :k]))
    return {"queries": nq, "order_differs_in_near_ties": swapped}
