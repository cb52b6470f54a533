"""Known-answer tests for the search oracle.  faiss is not available: these pin the oracle's OWN stated
semantics (exact IP, (score desc, row asc) tie rule, canonical score = the exact inner product rounded once
to fp32), check its formulations against each other, against Python rationals and the committed vectors
(tests/golden/g9_exact_dot.npz), measure the distance from plain fp32 arithmetic / sgemm (what IndexFlatIP
computes), and cover the edge cases of retriever/index.py (1024-query block boundary, id mapping, duplicate
rows, k == n, k > n)."""
import numpy as np
import pytest

from oracle import search_np as S


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _g9(golden):
    g = golden("g9_exact_dot.npz")
    off = g["offsets"]
    for c in range(len(off) - 1):
        yield (g["q_bits"][off[c]:off[c + 1]].view(np.float32), g["x_bits"][off[c]:off[c + 1]].view(np.float32), g["expected_bits"][c])


def test_canonical_score_golden_vectors(golden):
    """351 known answers computed with Python rationals (tests/golden/make_exact_dot_golden.py): exact midpoints, sticky bits, subnormal
    and vanishing results, overflow, cancellation, 120-bit exponent spreads.  Both C formulations must reproduce every bit."""
    n = 0
    for q, x, want in _g9(golden):
        for which in ("canonical", "exact"):
            got = np.float32(S.dot(q, x, which)).view(np.uint32)
            assert got == want, (n, which, len(q), hex(int(got)), hex(int(want)))
        n += 1
    assert n == 351


def test_canonical_equals_rationals_on_fresh_random_cases():
    rng = np.random.default_rng(0)
    for t in range(200):
        d = int(rng.choice([1, 2, 3, 5, 64, 257, 1024]))
        q = rng.standard_normal(d).astype(np.float32); x = rng.standard_normal(d).astype(np.float32)
        if t % 3 == 0:
            q *= np.float32(2.0) ** rng.integers(-50, 50, d).astype(np.float32)
        want = S.dot_fraction(q, x).view(np.uint32)
        assert np.float32(S.dot(q, x)).view(np.uint32) == want and np.float32(S.dot(q, x, "exact")).view(np.uint32) == want


def test_sequential_fp64_path_equals_super_accumulator_and_distance_from_fp32():
    """The plain sequential-fp64 formulation (with its certified rounding) against the integer super-accumulator on 1.2M unit-vector pairs
    at the metric's dimension, and the distance of the canonical score from fp32 arithmetic: an fp32 accumulation in index order and numpy's
    sgemm (the BLAS formulation faiss.IndexFlatIP uses) stay within a few fp32 ulps of it — the oracle is not an artefact of one summation order."""
    rng = np.random.default_rng(123)
    x = _unit(rng, 30000, 1024); q = _unit(rng, 40, 1024)
    can = S.scores_all(q, x, "canonical"); exa = S.scores_all(q, x, "exact")
    assert np.array_equal(can.view(np.uint32), exa.view(np.uint32))
    f32 = S.scores_all(q, x, "f32").astype(np.float64)
    sg = (q @ x.T).astype(np.float64)
    d_loop = np.abs(f32 - can).max(); d_sgemm = np.abs(sg - can).max()
    absd = max(d_loop, d_sgemm)
    ulp1 = 2.0 ** -24                                     # fp32 spacing just below 1.0 (cosine scores of real neighbours sit in [0.5, 1))
    print(f"canonical vs fp32 loop: max |diff| {d_loop:.2e} = {d_loop / ulp1:.1f} ulp(1-); vs sgemm: {d_sgemm:.2e} = {d_sgemm / ulp1:.1f} ulp(1-)")
    assert absd < 2e-6                                    # d * 2^-24 * sum|q_i x_i| ~ 1024 * 6e-8 * 0.03
    f64 = (q.astype(np.float64) @ x.astype(np.float64).T).astype(np.float32)
    assert (f64.view(np.uint32) != can.view(np.uint32)).mean() < 1e-4      # an unordered fp64 sum rounds the same except at fp32 boundaries


def test_three_scorers_agree_and_match_sgemm():
    rng = np.random.default_rng(1)
    x = _unit(rng, 5000, 128); q = _unit(rng, 37, 128)
    sc, ic = S.search_canonical(q, x, 10)
    sf, i_f = S.search_f64(q, x, 10)
    assert np.array_equal(ic, i_f) and np.array_equal(sc, sf)
    ss, is_ = S.search_sgemm(q, x, 10)
    np.testing.assert_allclose(ss, sc, atol=2e-6)
    assert (is_ == ic).mean() > 0.995          # sgemm rounding may swap near-ties only
    assert np.all(np.diff(sc, axis=1) <= 0)


def test_self_retrieval_and_planted_neighbours():
    rng = np.random.default_rng(2)
    x = _unit(rng, 2000, 64)
    q = x[[5, 77, 1999]] + 0.05 * rng.standard_normal((3, 64)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s, i = S.search_canonical(q, x, 5)
    assert list(i[:, 0]) == [5, 77, 1999]
    assert np.all(s[:, 0] > 0.9)


def test_duplicate_rows_tie_rule_and_k_bounds():
    rng = np.random.default_rng(3)
    x = _unit(rng, 50, 32)
    x[10] = x[3]; x[40] = x[3]
    q = x[[3]]
    s, i = S.search_canonical(q, x, 4)
    assert list(i[0, :3]) == [3, 10, 40] and s[0, 0] == s[0, 1] == s[0, 2]
    s_all, i_all = S.search_canonical(q, x, 50)
    assert sorted(i_all[0]) == list(range(50))
    with pytest.raises(ValueError):
        S.search_canonical(q, x, 51)
    with pytest.raises(ValueError):
        S.search_f64(q, x, 0)


def test_shard_merge_equals_unsharded():
    rng = np.random.default_rng(4)
    x = _unit(rng, 3000, 64); q = _unit(rng, 9, 64)
    x[2500] = x[100]                                   # tie across shards
    s, i = S.search_canonical(q, x, 20)
    bounds = [0, 700, 1500, 3000]
    ss, ii = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s_, i_ = S.search_canonical(q, x[a:b], 20)
        ss.append(s_); ii.append(i_ + a)
    ms, mi = S.merge_shards(ss, ii, 20)
    assert np.array_equal(mi, i) and np.array_equal(ms, s)


def test_indexer_id_mapping_and_block_boundary():
    rng = np.random.default_rng(5)
    x = _unit(rng, 300, 16)
    ids = [str(1000 + 7 * j) for j in range(300)]
    ix = S.OracleIndexer(16)
    ix.index_data(ids[:100], x[:100]); ix.index_data(ids[100:], x[100:].astype(np.float64))
    assert ix.index_id_to_db_id.dtype == np.int64 and len(ix.index_id_to_db_id) == 300
    q = _unit(rng, 1030, 16)                             # crosses the 1024-query block of index.py:39-46
    res = ix.search_knn(q, 3, verbose=False, exact_c=False)
    assert len(res) == 1030
    s, i = S.search_f64(q, x, 3)
    for r in (0, 1023, 1024, 1029):
        assert res[r][0] == [ids[j] for j in i[r]] and isinstance(res[r][0][0], str)
        assert res[r][1].dtype == np.float32 and np.array_equal(res[r][1], s[r])
    with pytest.raises(NotImplementedError):
        S.OracleIndexer(16, metric="l2")


def test_bf16_rounding_twins():
    rng = np.random.default_rng(6)
    a = np.concatenate([rng.standard_normal(1000).astype(np.float32), np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-40, 3.3895314e38], np.float32)])
    bits = S.f32_to_bf16_bits(a)
    for v, b in zip(a, bits):
        assert S.lib().kr_oracle_f32_to_bf16(float(v)) == int(b) or np.isnan(v)
    back = S.bf16_bits_to_f32(bits)
    fin = np.isfinite(a) & (np.abs(a) > 1e-30) & (np.abs(a) < 1e38)
    assert np.all(np.abs(back[fin] - a[fin]) <= np.abs(a[fin]) * 2.0 ** -8)
    assert np.isnan(back[np.isnan(a)]).all()
