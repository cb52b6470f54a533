#!/usr/bin/env python3
"""Design tool: which operand precision does the encoder need?  A torch emulation of the HIP forward's ROUNDING POINTS (16-bit MFMA operands, fp32
accumulation, 16-bit stored activations, optional (hi, lo) residual stream) run against the reference-generated goldens G10 — so that a precision
decision (bf16 vs f16 operands, residual low half) can be taken from numbers before any kernel is touched.  Runs on CPU or GPU.
Usage: python tools/precision_probe.py [weights: benign|out16|out60|...] [cases e.g. e5.c0,e5.c3]"""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np, torch
import g10_spec

dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")


from precision_emulation import forward, rnd  # noqa: E402,F401


if __name__ == "__main__":
    wname = sys.argv[1] if len(sys.argv) > 1 else "out16"
    cases = sys.argv[2].split(",") if len(sys.argv) > 2 else ["e5.c0", "e5.c3"]
    g = np.load(os.path.join(REPO, "tests", "golden", "g10_encoder_large_ext.npz"))
    W = {k: torch.from_numpy(v).to(dev) for k, v in g10_spec.weights(wname).items()}
    outs = {}
    modes = [("f16", 1, True), ("f16", 4, True)]
    SITES = {"ff2": {"h": "bf16", "w2": "bf16"}, "ff2out": {"h": "bf16", "w2": "bf16", "ctx": "bf16", "wo": "bf16"},
             "allw": {"h": "bf16", "w2": "bf16", "ctx": "bf16", "wo": "bf16", "w1": "bf16", "wqkv": "bf16"}, "w8": {"w2": "bf16", "wo": "bf16", "w1": "bf16", "wqkv": "bf16"}}
    if os.environ.get("PROBE_SITES"):      # e.g. PROBE_SITES=ff2,ff2out : f16 + 8-bit low half with single GEMMs / weights at bf16 precision
        modes = [("f16", 4, True)] + [("f16", 4, nm) for nm in os.environ["PROBE_SITES"].split(",")]
    for case in cases:
        tag, ci = case.split(".c"); ci = int(ci)
        B, S, layout, seed = g10_spec.CASES[wname][tag][ci]
        ids, mask = g10_spec.tokens(B, S, layout, seed)
        ref = g[f"{wname}.{case}.out"]
        for op, lo, y16 in modes:
            with torch.no_grad():
                out = forward(W, torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), 16, op, lo, True, "mean" if tag == "e5" else "cls",
                              site=SITES.get(y16) if isinstance(y16, str) else None).cpu().numpy()
            cos = (out * ref).sum(1)
            outs[(case, op, lo, y16)] = out
            print(f"{wname}.{case} B{B} S{S}  operands {op:4s} resid_lo={int(lo)} y16={y16}: max abs err {np.abs(out - ref).max():.2e}  1-cos {float((1 - cos).max()):.2e}", flush=True)
    if len(cases) >= 2:
        a, b = cases[0], cases[1]
        ra, rb = g[f"{wname}.{a}.out"], g[f"{wname}.{b}.out"]
        for op, lo, y16 in modes:
            print(f"scores {a} x {b}  operands {op:4s} resid_lo={int(lo)} y16={y16}: max |q.d - ref| = {np.abs(outs[(a, op, lo, y16)] @ outs[(b, op, lo, y16)].T - ra @ rb.T).max():.2e}")
