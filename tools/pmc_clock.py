#!/usr/bin/env python3
"""Sustained shader clock of the coarse scan from a counter pass (tools/profile_round.sh): python tools/pmc_clock.py <pmc_mfma dir> <bench json of the SAME run>.
GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3, so / 8 = shader cycles of the dispatch) of every k_coarse dispatch of the run / its scans = cycles per
scan; the bench line of the same process holds the HIP-event time per scan (roofline.launch_ms).  cycles / time = the clock the kernel really ran at; the
2.5 PFLOP/s nameplate assumes 2.4 GHz, so frac x 2.4 / clock = the fraction of the peak the chip offers AT THAT CLOCK."""
import collections, csv, glob, json, sys
d, bench = sys.argv[1], json.load(open(sys.argv[2]))
cyc = 0.0; ids = set()
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "k_coarse<" in r["Kernel_Name"] and "false, false" in r["Kernel_Name"]:
            cyc += float(r["Counter_Value"]) / 8.0; ids.add((f, r.get("Dispatch_Id")))
scans = bench["steps"] + bench["warmup"]
ms = bench["roofline"]["launch_ms"]
ghz = cyc / scans / (ms * 1e-3) / 1e9
print(json.dumps({"kernel": "k_coarse", "rows": bench["config"]["rows_per_gpu"], "queries": bench["config"]["queries"], "scans": scans, "dispatches": len(ids),
                  "shader_cycles_per_scan": cyc / scans, "ms_per_scan_same_run": ms, "sustained_clock_ghz": ghz, "nameplate_clock_ghz": 2.4,
                  "frac_same_run": bench["roofline"]["frac"], "frac_of_clock_adjusted_peak": bench["roofline"]["frac"] * 2.4 / ghz,
                  "note": "threshold launches only (the DIRECT round-0 launch is not in the sum: its cycles are a few per cent of a scan, so the clock is a slight under-estimate)"}, indent=1))
