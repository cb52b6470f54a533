#!/usr/bin/env python3
"""Per-kernel HBM-side bytes of one encoder shape from two rocprofv3 --pmc passes (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction,
WRITE_SIZE as reported; counted on the fabric side of L2: Infinity-Cache hits included) next to the ALGORITHMIC bytes of the kernel.
Usage: pmc_encoder_traffic.py <dir> B S forwards"""
import collections, csv, glob, json, sys
d, B, S, fw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
T, H, FF, L = B * S, 1024, 4096, 24
agg = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/enc_{ctr}_{B}_{S}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "kr::enc" not in name:
                continue
            a = agg.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": collections.Counter()})
            a[ctr] += float(r["Counter_Value"]) * 1024.0          # the counters are in KiB
            a["n"][ctr] += 1
# algorithmic bytes per dispatch: activations once in / once out, weights once (16-bit operands; residual low half = 1 byte per element)
alg = {
    "k_ln16": T * H * (2 + 2 + 1 + 2 + 1),                       # y, hi, lo in; hi, lo out
    "k_attn": T * H * 2 * 4,                                      # q, k, v^T in; ctx out
    "k_proj<0": T * H * 2 + 3 * H * H * 2 + T * 3 * H * 2,        # QKV
    "k_proj<2": T * H * 2 + FF * H * 2 + T * FF * 2,              # FF1 + GELU
}
out = collections.OrderedDict()
for name, a in sorted(agg.items()):
    n = max(a["n"].values())
    if n < fw * L // 2:
        continue
    per = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) / n
    e = {"dispatches": n, "fetch_x2_gb_per_dispatch": 2.0 * a["FETCH_SIZE"] / n / 1e9, "write_gb_per_dispatch": a["WRITE_SIZE"] / n / 1e9, "hbm_gb_per_dispatch": per / 1e9}
    for key, v in alg.items():
        if key in name:
            e["algorithmic_gb_per_dispatch"] = v / 1e9
            e["traffic_over_algorithmic"] = per / v
    out[name[:90]] = e
print(json.dumps({"shape": [B, S], "tokens": T, "kernels": out}, indent=1))
