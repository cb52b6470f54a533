"""CPU oracle for the KiRAG dense-retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``kirag_amd``) never imports this package and fails loudly
when its HIP library is missing.

Pinning status
--------------
* encoder half (``encoder_np``): pinned against outputs of the reference's own
  ``retriever/encoders.py`` (``E5Encoder`` / ``BGEEncoder`` / ``average_pool``)
  imported in the build container; the vectors live in ``tests/golden/`` and
  were produced by ``tests/golden/make_golden.py``.
* search half (``search_np`` / ``search_c.c``): the reference delegates to
  faiss-cpu==1.8.0.post1 ``IndexFlatIP`` (``retriever/index.py:13,47``), which is
  neither vendored in the reference nor installed here, and the reference has
  no tests.  **Parity with faiss itself is unpinned** (no faiss output exists to
  compare with): the oracle restates the published semantics of
  ``IndexFlatIP.search`` (exact inner product, k best per query, scores
  descending) and fixes what faiss leaves to its BLAS — the rounding of the sum
  and the tie rule — by a definition that depends on no summation order: the
  score is the EXACT inner product of the fp32 inputs rounded once to fp32,
  ties go to the lower row.  That definition is pinned by three independent
  formulations (integer super-accumulator and certified sequential fp64 in
  ``search_c.c``, Python rationals in ``search_np.dot_fraction``) and by the
  committed known-answer vectors ``tests/golden/g9_exact_dot.npz``
  (``tests/golden/make_exact_dot_golden.py``).
"""
