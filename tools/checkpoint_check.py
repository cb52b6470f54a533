#!/usr/bin/env python3
"""Precision self-check for a REAL checkpoint: does the HIP encoder's default operand precision hold north_star's 1e-3 score
tolerance on YOUR weights, and could a cheaper mode be used?

    python tools/checkpoint_check.py <hf_dir> [--texts file.txt] [--n 16] [--max-length 128] [--pool mean|cls] [--emulate] [--json out.json]

<hf_dir> is any HF BERT-family checkpoint directory (config.json + weights; e5-large-v2 / bge-large-en: retriever/e5.py:18-19 of the reference).  No
checkpoint exists in the build container (no network), so the defaults of this library — f16 MFMA operands + the residual stream's low half — were chosen on
SYNTHETIC outlier recipes (golden set G10: "out3" models the two orders of magnitude between outlier and median channels that real
BERT-family checkpoints show).  This tool replaces that assertion by a measurement the day weights exist.  It reports

 1. activation statistics of the module's own fp32 forward, per layer: the residual stream after each LayerNorm (max, p99.9, median of |x| — the quantity
    the out3 / out16 recipes model) and the largest value any 16-bit-stored tensor takes (q / k / v, context, dense outputs, GELU outputs) against the f16
    limit 65504 that KR_ERANGE guards;
 2. the worst |q.d - reference q.d| over all (query text, passage text) pairs for the four precision modes — f16 + low half (default), f16, bf16 + low half,
    bf16 — where the reference is the module's own fp32 forward + the encoder's pooling (encoders.py:67-77 / 106-118) and the tested path is the HIP
    encoder (GPU present) or, with --emulate or without a GPU, the torch emulation of its rounding points (tools/precision_emulation.py);
 3. a recommendation: the cheapest mode whose error stays below HALF the 1e-3 tolerance, else the default.

Texts: --texts (one per line; the first half are used as queries, the rest as passages, E5 prefixes added) needs the checkpoint's tokenizer; without a
tokenizer in the directory (or with --random-tokens) seeded random token ids of ragged lengths are used."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch

F16_MAX = 65504.0
MODES = [("f16", True), ("f16", False), ("bf16", True), ("bf16", False)]      # default first; cost order: last is cheapest
MODE_COST = {("f16", True): "default: 29.5 ms per 1000-query step", ("f16", False): "-3 %", ("bf16", True): "-3.5 %", ("bf16", False): "-7 %"}


def load_inputs(args, hf_dir, vocab):
    """-> list of (input_ids, attention_mask) int64 numpy batches: [queries, passages]"""
    tok = None
    if not args.random_tokens and os.path.isdir(hf_dir):
        try:
            from transformers import AutoTokenizer
            tok = AutoTokenizer.from_pretrained(hf_dir)
        except Exception as e:      # noqa: BLE001 — a directory without tokenizer files
            print(f"# no tokenizer in {hf_dir} ({type(e).__name__}): seeded random token ids instead", file=sys.stderr)
    if tok is not None:
        if args.texts:
            texts = [ln.strip() for ln in open(args.texts) if ln.strip()][: 2 * args.n]
        else:
            base = ["who discovered penicillin", "capital of the country that hosted the 1992 olympics", "when was the eiffel tower completed",
                    "what is the boiling point of water at altitude", "Alexander Fleming discovered penicillin in 1928 at St Mary's Hospital in London.",
                    "Barcelona hosted the 1992 Summer Olympics; the capital of Spain is Madrid.", "The Eiffel Tower was completed in March 1889 for the World's Fair.",
                    "Water boils at lower temperatures at higher altitudes because the air pressure is lower."]
            texts = (base * ((2 * args.n + len(base) - 1) // len(base)))[: 2 * args.n]
        half = max(1, len(texts) // 2)
        out = []
        for prefix, part in (("query: ", texts[:half]), ("passage: ", texts[half:])):
            enc = tok([prefix + t for t in part], padding=True, truncation=True, max_length=args.max_length, return_tensors="np")
            out.append((enc["input_ids"].astype(np.int64), enc["attention_mask"].astype(np.int64)))
        return out
    rng = np.random.default_rng(args.seed)
    out = []
    for S in (min(32, args.max_length), args.max_length):
        lens = np.clip(rng.normal(0.8 * S, 0.2 * S, size=args.n).astype(np.int64), 3, S)
        ids = rng.integers(1000 if vocab > 2000 else 5, vocab, size=(args.n, S), dtype=np.int64)
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
        ids[:, 0] = 101 if vocab > 2000 else 1
        ids = ids * mask
        out.append((ids, mask))
    return out


def reference_and_stats(model, batches, pool, dev):
    """the module's own fp32 forward with hooks -> (embeddings per batch, per-layer statistics)"""
    L = model.config.num_hidden_layers
    stats = [{"ln1": [], "ln2": [], "max16": 0.0} for _ in range(L)]
    hooks = []
    cur = {}

    def grab(li, key):
        def fn(_m, _inp, out):
            t = out[0] if isinstance(out, tuple) else out
            if key in ("ln1", "ln2"):
                stats[li][key].append(t.detach()[cur["mask"]].abs().float().flatten().cpu())
            stats[li]["max16"] = max(stats[li]["max16"], float(t.detach()[cur["mask"]].abs().max()))
        return fn
    for li, layer in enumerate(model.encoder.layer):
        att = layer.attention
        for key, mod in (("q", att.self.query), ("k", att.self.key), ("v", att.self.value), ("ctx", att.self), ("y1", att.output.dense), ("ln1", att.output.LayerNorm),
                         ("h", layer.intermediate), ("y2", layer.output.dense), ("ln2", layer.output.LayerNorm)):
            hooks.append(mod.register_forward_hook(grab(li, key)))
    embs = []
    with torch.no_grad():
        for ids, mask in batches:
            cur["mask"] = torch.from_numpy(mask).bool().to(dev)
            out = model(input_ids=torch.from_numpy(ids).to(dev), attention_mask=torch.from_numpy(mask).to(dev)).last_hidden_state
            if pool == "mean":
                m = torch.from_numpy(mask).to(dev)[..., None].bool()
                e = out.masked_fill(~m, 0.0).sum(1) / m.sum(1)
            else:
                e = out[:, 0]
            embs.append(torch.nn.functional.normalize(e, dim=-1).cpu().numpy())
    for h in hooks:
        h.remove()
    table = []
    for li, st in enumerate(stats):
        row = {"layer": li, "max_16bit_tensor": st["max16"], "f16_headroom": F16_MAX / max(st["max16"], 1e-30)}
        for key in ("ln1", "ln2"):
            v = torch.cat(st[key]) if st[key] else torch.zeros(1)
            row[key] = {"max": float(v.max()), "p99.9": float(torch.quantile(v[:: max(1, v.numel() // 2_000_000)], 0.999)), "median": float(v.median())}
        table.append(row)
    return embs, table


def run_mode_hip(model, batches, pool, dtype, lo):
    from kirag_amd.retriever.encoders import HipBertForward
    h = HipBertForward(model.config, 0, operand_dtype=dtype, residual_lo=lo)
    h.sync(model)
    return [h.forward_np(ids, mask, 1 if pool == "cls" else 0) for ids, mask in batches]


def run_mode_emulated(model, batches, pool, dtype, lo, dev):
    from precision_emulation import forward
    W = {k: v.detach().float().to(dev) for k, v in model.state_dict().items()}
    outs = []
    with torch.no_grad():
        for ids, mask in batches:
            outs.append(forward(W, torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), model.config.num_attention_heads, dtype, 4 if lo else 0, True,
                                "mean" if pool == "mean" else "cls").cpu().numpy())
    return outs


def check(hf_dir, args):
    from transformers import AutoConfig, AutoModel
    cfg = AutoConfig.from_pretrained(hf_dir)
    if getattr(cfg, "model_type", "bert") != "bert":
        raise SystemExit(f"{hf_dir}: model_type {cfg.model_type!r}; the HIP encoder implements the BERT architecture (e5 / bge)")
    model = AutoModel.from_pretrained(hf_dir, add_pooling_layer=False)
    return check_model(model, args, hf_dir)


def check_model(model, args, hf_dir="(in-memory model)"):
    """the same check on an already constructed HF BertModel (tests build one from the synthetic outlier recipes)"""
    cfg = model.config
    use_hip = torch.cuda.is_available() and not args.emulate
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    model = model.float().eval().to(dev)
    batches = load_inputs(args, hf_dir, cfg.vocab_size)
    ref, table = reference_and_stats(model, batches, args.pool, dev)
    print(f"# {hf_dir}: {cfg.num_hidden_layers} layers, hidden {cfg.hidden_size}; {sum(len(b[0]) for b in batches)} sequences, pooling {args.pool}; "
          f"tested path: {'HIP encoder (libkirag_amd)' if use_hip else 'torch emulation of the rounding points'}")
    print("# residual stream after each LayerNorm (|x|: max / p99.9 / median) and the largest 16-bit-stored activation of the layer (f16 limit 65504)")
    for r in table:
        print(f"  layer {r['layer']:2d}  LN1 {r['ln1']['max']:9.2f} / {r['ln1']['p99.9']:7.2f} / {r['ln1']['median']:6.3f}   LN2 {r['ln2']['max']:9.2f} / {r['ln2']['p99.9']:7.2f} / "
              f"{r['ln2']['median']:6.3f}   max stored {r['max_16bit_tensor']:10.2f}  (f16 headroom x{r['f16_headroom']:.0f})")
    ratio = max(max(r["ln1"]["max"] / max(r["ln1"]["median"], 1e-9), r["ln2"]["max"] / max(r["ln2"]["median"], 1e-9)) for r in table)
    headroom = min(r["f16_headroom"] for r in table)
    print(f"# outlier ratio max|x| / median|x| of the residual stream: {ratio:.0f}  (the synthetic recipes the defaults were chosen on: out3 ~300, out16 ~1200, out60 ~4700); "
          f"smallest f16 headroom x{headroom:.0f}" + ("  ** below 4: f16 operands may overflow (KR_ERANGE) — use bf16 **" if headroom < 4 else ""))
    s_ref = ref[0] @ ref[1].T
    results = []
    for dtype, lo in MODES:
        outs = run_mode_hip(model, batches, args.pool, dtype, lo) if use_hip else run_mode_emulated(model, batches, args.pool, dtype, lo, dev)
        finite = all(np.isfinite(o).all() for o in outs)
        err = float(np.abs(outs[0] @ outs[1].T - s_ref).max()) if finite else float("inf")
        cos = float(max((1 - (o * r).sum(1)).max() for o, r in zip(outs, ref))) if finite else float("inf")
        results.append({"operand_dtype": dtype, "residual_lo": lo, "worst_score_error": err, "worst_1_minus_cos": cos, "finite": finite})
        print(f"  {dtype:4s} {'+ low half' if lo else '          '}  worst |q.d - ref| = {err:.2e}   worst 1 - cos = {cos:.2e}   ({MODE_COST[(dtype, lo)]})"
              + ("" if finite else "   NON-FINITE OUTPUT"))
    nseq = sum(len(b[0]) for b in batches)
    if nseq < 64:
        print(f"# note: the worst case over {len(batches[0][0])} x {len(batches[1][0])} pairs grows with the sample — golden set G10 measured 1.4e-4 / 7.2e-4 / 1.0e-3 / 5.9e-3 for the four "
              "modes on the out3 recipe with 30 sequences of up to 512 tokens; use --n 32 or more texts of production length before switching modes")
    ok = [r for r in results if r["finite"] and r["worst_score_error"] <= 5e-4]
    pick = ok[-1] if ok else results[0]            # MODES is ordered dearest -> cheapest
    rec = {"operand_dtype": pick["operand_dtype"], "residual_lo": pick["residual_lo"], "within_half_tolerance": bool(ok)}
    if not ok:
        print("# recommendation: keep the default (f16 + low half); NO mode stays below 5e-4 on these inputs" +
              (" — the default itself exceeds north_star's 1e-3: report this checkpoint" if results[0]["worst_score_error"] > 1e-3 else ""))
    else:
        env = f"KIRAG_AMD_ENCODER_DTYPE={pick['operand_dtype']} KIRAG_AMD_RESIDUAL_LO={int(pick['residual_lo'])}"
        print(f"# recommendation: {pick['operand_dtype']}{' + low half' if pick['residual_lo'] else ''} ({env}): the cheapest mode within half the 1e-3 tolerance")
    out = {"checkpoint": hf_dir, "tested_path": "hip" if use_hip else "emulation", "outlier_ratio": ratio, "f16_headroom": headroom, "layers": table, "modes": results,
           "recommendation": rec}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("hf_dir")
    ap.add_argument("--texts", default=None)
    ap.add_argument("--n", type=int, default=32, help="sequences per side (queries / passages)")
    ap.add_argument("--max-length", type=int, default=128)
    ap.add_argument("--pool", default="mean", choices=["mean", "cls"], help="mean = E5Encoder, cls = BGEEncoder")
    ap.add_argument("--emulate", action="store_true", help="torch emulation of the rounding points even when a GPU is present")
    ap.add_argument("--random-tokens", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default=None)
    return ap.parse_args(argv)


if __name__ == "__main__":
    a = parse()
    check(a.hf_dir, a)
