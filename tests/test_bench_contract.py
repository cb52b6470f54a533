"""The committed bench line (profiles/rNN/bench_plain.json of the NEWEST round that has one, produced by `python bench.py` on an MI355X) carries every
field of the driver's contract, with consistent arithmetic.  bench.py itself needs a GPU; this checks the artefact the round ships."""
import glob
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench_line():
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]", "bench_plain.json")))
    assert cands, "no committed bench line under profiles/rNN/"
    return cands[-1]


def test_committed_bench_line_has_the_contract_fields():
    path = _newest_bench_line()
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["queries"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6      # value = units per step / time per step
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    flops = 2.0 * d["config"]["queries"] * r["rows_scanned"] * d["config"]["dim"]
    assert abs(r["achieved"] - flops / (r["launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-6           # algorithmic FLOPs / measured launch time
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_gb"] * 0.9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1 and d["value"] > 10 * c["value"]
    if os.path.basename(os.path.dirname(path)) >= "r05":     # round 5 on: the latency block and the reference-surface search ride in the same line
        lat = d["latency"]
        assert set(lat["forwards"]) == {"1x32", "1x256", "8x128", "125x32"} and all(0 < v < 50 for v in lat["forwards"].values()) and 0 < lat["kirag_hop_nq1"]["ms"] < 50
        sf = d["surface"]
        assert sf["lists_match_c_abi"] is True and abs(sf["ratio"] - sf["search_knn_queries_per_s"] / sf["c_abi_queries_per_s"]) < 1e-6


    if os.path.basename(os.path.dirname(path)) >= "r06":     # round 6 on: passages-encoded/s at the reference's ENTRY POINT (cal_doc_embeddings from text) and the hop at its surface
        ep = d["encode"]["entry_point"]
        assert ep["passages"] >= 32768 and abs(ep["ratio"] - ep["passages_per_s"] / ep["pretokenised_resident_passages_per_s"]) < 1e-9
        assert ep["ratio"] >= 0.95 and ep["feed"]["packed_forward"] is True, ep          # VERDICT r05 item 1's bar, in the line the round ships
        hs = d["latency"]["kirag_hop_nq1_surface"]
        assert hs["same_ids_as_c_abi"] is True and abs(hs["ratio_to_c_abi"] - hs["ms"] / hs["c_abi_same_tokens_ms"]) < 1e-9 and hs["ms"] < 2 * hs["c_abi_same_tokens_ms"]
        assert r.get("traffic_range_over_boxes") is None or r["traffic_range_over_boxes"][0] <= r["traffic"] <= r["traffic_range_over_boxes"][1]


def test_bench_defaults_are_the_metric_configuration():
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("kr_bench", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = m.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.total_rows, a.queries, a.topk, a.dim, a.query_tokens, a.passage_tokens) == (1, 5_000_000, 1000, 100, 1024, 32, 128)
    assert a.steps * 35e-3 < 60 and a.coarse_dtype == "bf16"        # the default run finishes within minutes


def _run_bench(extra, env_extra=None, timeout=600):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must itself start 2 ranks and print ONE line with n_gpus == 2
    (round-1 finding: the flag was parsed and ignored).  --plumbing-only = everything but the GPU work, gloo on CPU."""
    p, lines = _run_bench(["--gpus", "2", "--plumbing-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None


def test_gpus_flag_must_match_world_size():
    p, lines = _run_bench(["--gpus", "2", "--plumbing-only"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and not lines and "WORLD_SIZE=3" in p.stderr


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("split,steps", [("batch", "3"), ("queries", "2")])
def test_bench_gpus_2_real_step_on_one_box(split, steps):
    """The real N = 2 step (row shards, all-gather of query vectors and of per-shard top-k, device merge) started by `python bench.py --gpus 2` itself, under
    both encode schedules: "batch" (rank r encodes the whole batch of every 2nd step; 3 steps = one full and one partial block) and "queries" (every rank
    encodes half of every batch).  The two ranks share the box's one GPU, so the collective backend is gloo (RCCL wants one device per rank)."""
    p, lines = _run_bench(["--gpus", "2", "--total-rows", "300000", "--queries", "256", "--steps", steps, "--warmup", "1", "--passages", "64",
                           "--no-cpu-baseline", "--encode-split", split], {"KIRAG_BENCH_BACKEND": "gloo"}, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["rows_per_gpu"] == 150000
    st = d["search_stats"]
    assert st["certified"] + st["fallback"] == st["queries"]
