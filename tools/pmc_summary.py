#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (pmc_fetch/, pmc_write/ under the given dir) per kernel name -> JSON on stdout."""
import csv, glob, json, sys, collections
d = sys.argv[1]
out = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)
    agg = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"].split("(")[0]
            if not name.startswith(("kr::", "void kr::")):
                continue
            a = agg.setdefault(name, {"dispatches": 0, "sum_kb": 0.0})
            a["dispatches"] += 1
            a["sum_kb"] += float(r["Counter_Value"])
    out[ctr + "_KB"] = agg
print(json.dumps(out, indent=1))
