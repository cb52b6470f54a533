#!/usr/bin/env python3
"""MFMA-pipe utilisation from rocprofv3 --pmc passes (north_star: "rocprof ... MFMA utilisation").

Input: a directory holding rocprofv3 counter_collection CSVs of `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [SQ_BUSY_CYCLES SQ_WAVE_CYCLES ...]`.
Per kernel: dispatches, counter sums, and
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
(MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles, 32 per v_mfma_f32_32x32x16_bf16, summed over the 1024 SIMDs of the chip;
GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs, so / 8 = the kernel's duration in shader cycles).  For kernels whose MFMA count is known
algorithmically (`name=flops executed by all its dispatches in the profiled run`) the expected busy cycles = flops / (32*32*16*2) * 32 are printed next to the counter, which
calibrates the counter's unit on this stack instead of trusting it."""
import collections, csv, glob, json, sys

d = sys.argv[1]
expect = {}
for a in sys.argv[2:]:
    if a.startswith("--expect"):
        continue
    if "=" in a:
        k, v = a.split("=", 1); expect[k] = float(v)
agg = collections.OrderedDict()
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("kr::"):
            continue
        a = agg.setdefault(name, {"dispatch_ids": set(), "counters": collections.defaultdict(float)})
        a["dispatch_ids"].add((f, r.get("Dispatch_Id")))
        a["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
out = collections.OrderedDict()
for name, a in agg.items():
    c = dict(a["counters"]); n = len(a["dispatch_ids"])
    e = {"dispatches": n, **{k: v for k, v in c.items()}}
    if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        e["kernel_cycles_per_dispatch"] = cyc / n
        e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    for key, fl in expect.items():
        if key in name and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["expected_busy_cycles_total_from_flops"] = fl / (32 * 32 * 16 * 2) * 32      # fl = MFMA flops executed by ALL its dispatches of the run
            e["counter_busy_cycles_total"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    out[name] = e
print(json.dumps(out, indent=1))
