#!/usr/bin/env python3
"""Repro of the ROCm 7.2 / gfx950 virtual-memory hazard the index's growth path has to avoid (kirag_amd/csrc/search.hip, VBuf::release):
an address range that is unmapped, freed with hipMemAddressFree, reserved again (same addresses) and mapped to new physical chunks is read by kernels
through stale translations.  `KIRAG_AMD_DEBUG_FREE_VA=1 python tools/vmm_va_reuse_repro.py plain` shows it (self-matches of rows 0..4 are not found: the scan kernels
do not see the rows a copy engine reads back correctly); without the variable the library retires the range instead and the search is right."""
import os, sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from kirag_amd.retriever.index import FlatIPIndex
def unit(n, d): return torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
n, d = 120018, 1024
def chk(tag, ix, x):
    s, i = ix.search(x[:5].clone(), 1)
    st = ix.stats()
    print(tag, "search", i[:, 0].tolist(), [round(float(v), 4) for v in s[:, 0]], "certified", st["certified"], "fine", st["fine"], "exact", st["exact"], "ntotal", ix.ntotal, flush=True)
variant = sys.argv[1]
x = unit(n, d)
ix = FlatIPIndex(d, device=0)
ix.add(x[:70000]); ix.add(x[70000:])
if variant == "recon": ix.reconstruct_n(0, 4)
if variant == "search_first": chk("first", ix, x)
x2 = x.clone(); x2[n - 3] = x2[1]
if variant == "del_first":
    del ix
ix = FlatIPIndex(d, device=0); ix.add(x2); x = x2
chk(variant, ix, x)
chk(variant + " again", ix, x)
