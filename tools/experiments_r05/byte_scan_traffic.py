#!/usr/bin/env python3
"""HBM-side bytes per launch of the kernels of the one-query search (byte pre-scan on) from two rocprofv3 --pmc passes of byte_scan_profile.py (FETCH_SIZE doubled per
MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as reported; counters in KiB) next to the algorithmic bytes.  Usage: byte_scan_traffic.py <dir> rows"""
import collections, csv, glob, json, sys
d, n = sys.argv[1], int(sys.argv[2])
agg = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "kr::" not in name:
                continue
            a = agg.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": collections.Counter()})
            a[ctr] += float(r["Counter_Value"]) * 1024.0
            a["n"][ctr] += 1
sample = 4096 + 31 * 4096                                   # rows of the direct round and of the one 16-bit round behind it
alg = {"k_coarse_q32<kr::BF16, 3, 8>": (n - sample) * (1024 + 4), "k_coarse_q32<kr::BF16, 0, 16>": 31 * 4096 * 2048, "k_coarse_q32<kr::BF16, 1, 16>": 4096 * 2048}
out = collections.OrderedDict()
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["FETCH_SIZE"]):
    c = max(a["n"].values())
    if c < 100:
        continue
    per = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) / c
    e = {"launches": c, "fetch_x2_mb_per_launch": 2.0 * a["FETCH_SIZE"] / c / 1e6, "write_mb_per_launch": a["WRITE_SIZE"] / c / 1e6, "hbm_mb_per_launch": per / 1e6}
    for key, v in alg.items():
        if key in name:
            e["algorithmic_mb_per_launch"] = v / 1e6
            e["traffic_over_algorithmic"] = per / v
    out[name[:80]] = e
print(json.dumps({"rows": n, "kernels": out}, indent=1))
