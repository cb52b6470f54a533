#!/usr/bin/env python3
"""Per-kernel resource notes of the gfx950 code objects embedded in a built object / shared library.

    python tools/kernel_notes.py kirag_amd/libkirag_amd.so [--spills]

Reads the clang offload bundles out of the file (no GPU needed), runs `llvm-readelf --notes` on every gfx950 code object and prints
VGPR / AGPR / SGPR counts, spill counts, scratch and static LDS bytes per kernel.  `tests/test_build_quality.py` uses `kernels()` to
fail the CPU suite when a product kernel spills vector registers: a scratch reload is a VMEM operation queued in the same in-order
pipe as the LDS-DMA rings, and hipcc follows it with `s_waitcnt vmcnt(0)` (round 4 found exactly that in front of the DMA of
k_attn_dma's partial chunk and in the V^T epilogue of the QKV projection)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    """Every code object for `arch` bundled in `path` (bytes)."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if arch in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + 1


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def count_instruction(path, mnemonic, arch="gfx950"):
    """How often `mnemonic` occurs in the disassembly of the `arch` code objects of `path` (tests/test_build_quality.py: instructions this compiler
    mis-models on gfx950 must not be selected)."""
    n = 0
    for co in code_objects(path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            txt = subprocess.run([OBJDUMP, "-d", "--mcpu=" + arch, f.name], capture_output=True, text=True, check=True).stdout
            n += len(re.findall(r"\b" + re.escape(mnemonic) + r"\b", txt))
        finally:
            os.unlink(f.name)
    return n


def kernels(path, arch="gfx950"):
    """List of dicts (one per kernel): name, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size, group_segment_fixed_size (all ints except name)."""
    res = []
    for co in code_objects(path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        try:
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key == "agpr_count" and "name" in cur:      # first key of the next kernel's record
                res.append(cur)
                cur = {}
            cur[key] = int(val) if re.fullmatch(r"-?\d+", val) else val
            if key == "wavefront_size":                    # last key of a record
                res.append(cur)
                cur = {}
    return [r for r in res if "name" in r and "vgpr_count" in r]


if __name__ == "__main__":
    only_spills = "--spills" in sys.argv
    for r in kernels([a for a in sys.argv[1:] if not a.startswith("--")][0]):
        if only_spills and not (r.get("vgpr_spill_count") or r.get("private_segment_fixed_size")):
            continue
        print("%-92s vgpr %3d agpr %3d sgpr %3d | spill v %3d s %3d | scratch %4d B | lds %6d B" % (
            r["name"][:92], r.get("vgpr_count", 0), r.get("agpr_count", 0), r.get("sgpr_count", 0), r.get("vgpr_spill_count", 0),
            r.get("sgpr_spill_count", 0), r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0)))
