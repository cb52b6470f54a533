"""The committed bench line (profiles/r01/bench_plain.json, produced by `python bench.py` on an MI355X) carries every field of the driver's
contract, with consistent arithmetic.  bench.py itself needs a GPU; this checks the artefact the round ships."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(REPO, "profiles", "r01", "bench_plain.json")))
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
