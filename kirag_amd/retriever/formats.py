"""On-disk formats of the index: the reference's ``index.faiss`` (``retriever/index.py:62,73``: ``faiss.write_index`` / ``read_index`` of an
``IndexFlatIP``, layout restated from faiss 1.8's writer) and the native ``KRSHARD1`` row-shard files ``ShardedIndexer`` writes."""
from __future__ import annotations

import os
import struct
from typing import Optional

import numpy as np

from .flat_index import FlatIPIndex

_IO_CHUNK_ROWS = 1 << 18


# ---------------------------------------------------------------------------------------------------------
# faiss flat-index file layout.  faiss is a third-party dependency of the reference (requirements.txt:10) and
# its source is not on disk here: the layout below restates faiss 1.8 `write_index` for IndexFlat from its
# published io code (impl/index_write.cpp: fourcc, write_index_header, WRITEXBVECTOR) and is UNVERIFIED
# against a real faiss build in this environment: tests/test_capi_and_host.py pins the writer to a hand-assembled
# byte string of that field list (header 4+4+8+8+8+1+4 = 37 bytes, then the WRITEXBVECTOR count = payload bytes / 4),
# which guards the layout against regressions but is NOT a round trip through faiss.
#   u32  fourcc "IxFI"
#   i32  d ; i64 ntotal ; i64 dummy (1<<20) ; i64 dummy (1<<20) ; u8 is_trained ; i32 metric_type (0 = IP)
#   u64  number of float32 values (= ntotal * d) ; float32[ntotal * d] row-major
# ---------------------------------------------------------------------------------------------------------
_FOURCC_IXFI = struct.unpack("<I", b"IxFI")[0]


def write_faiss_flat_ip(index: FlatIPIndex, path: str) -> None:
    n, d = index.ntotal, index.d
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _FOURCC_IXFI))
        f.write(struct.pack("<iqqqBi", d, n, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", n * d))
        for s in range(0, n, _IO_CHUNK_ROWS):
            m = min(_IO_CHUNK_ROWS, n - s)
            f.write(index.reconstruct_n(s, m).tobytes())


def read_faiss_flat_ip(path: str, device: Optional[int] = None, coarse_dtype: str = "bf16", row_range=None) -> FlatIPIndex:
    """``row_range = (rank, world)`` loads only that rank's contiguous share of the rows (``ShardedIndexer``); the returned index carries
    ``file_ntotal`` (rows in the file) and ``row_offset`` (first row held)."""
    with open(path, "rb") as f:
        (fourcc,) = struct.unpack("<I", f.read(4))
        if fourcc != _FOURCC_IXFI:
            raise ValueError(f"{path}: not a faiss IndexFlatIP file (fourcc {struct.pack('<I', fourcc)!r})")
        d, n, _, _, _trained, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        if metric != 0:
            raise ValueError(f"{path}: metric_type {metric} is not METRIC_INNER_PRODUCT")
        (nfloat,) = struct.unpack("<Q", f.read(8))
        if nfloat != n * d:
            raise ValueError(f"{path}: payload {nfloat} floats != ntotal*d = {n * d}")
        a, b = 0, n
        if row_range is not None:
            rank, world = row_range
            per = (n + world - 1) // world
            a, b = min(rank * per, n), min((rank + 1) * per, n)
            f.seek(a * d * 4, os.SEEK_CUR)
        index = FlatIPIndex(d, device=device, coarse_dtype=coarse_dtype)
        index.reserve(b - a)
        for s in range(a, b, _IO_CHUNK_ROWS):
            m = min(_IO_CHUNK_ROWS, b - s)
            buf = np.frombuffer(f.read(m * d * 4), dtype=np.float32).reshape(m, d)
            index.add(buf)
        index.file_ntotal, index.row_offset = n, a
    return index


# ---------------------------------------------------------------------------------------------------------
# Native shard files: what one rank's FlatIPIndex holds, byte for byte, so that a reload does not re-quantise.
#   bytes 0..7    magic  b"KRSHARD1"
#   i32 d ; i32 coarse_dim ; i32 coarse_dtype (0 bf16, 1 f16) ; i32 reserved = 0
#   i64 row0 (first global row) ; i64 rows ; i64 ntotal (rows of the whole corpus)
#   f32 bounds[2]   max |x - c(x)|_2 , max |c(x)|_2 over the rows of the index that wrote the file
#   f32 [rows, d]   master rows ; u16 [rows, coarse_dim]  scan copy
# ---------------------------------------------------------------------------------------------------------
SHARD_MANIFEST = "kirag_shards.json"


def _ids_crc32(ids) -> int:
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(np.asarray(ids, dtype=np.int64)).tobytes()) & 0xFFFFFFFF)


def _manifest_matches(manifest_path: str, id_map) -> bool:
    """The native shards belong to the ``index_meta.faiss`` next to them: same row count and (manifests written since round 3) the same CRC-32 of
    the id map.  A directory later rewritten with the reference-format files fails this and is loaded from ``index.faiss``."""
    import json
    try:
        with open(manifest_path) as f:
            man = json.load(f)
    except Exception:
        return False
    if int(man.get("ntotal", -1)) != len(id_map):
        return False
    crc = man.get("meta_crc32")
    return crc is None or int(crc) == _ids_crc32(id_map)
_SHARD_MAGIC = b"KRSHARD1"
_SHARD_HEADER = struct.Struct("<8siiiiqqqff")


def shard_file_name(rank: int, world: int) -> str:
    return f"index_shard_{rank:04d}_of_{world:04d}.krshard"


def write_native_shard(index: FlatIPIndex, path: str, row0: int, ntotal: int) -> None:
    n, d, dc = index.ntotal, index.d, index.coarse_dim
    b = index.bounds() if n else np.zeros(2, np.float32)
    with open(path, "wb") as f:
        f.write(_SHARD_HEADER.pack(_SHARD_MAGIC, d, dc, {"bf16": 0, "f16": 1}[index.coarse_dtype], 0, int(row0), n, int(ntotal), float(b[0]), float(b[1])))
        for s in range(0, n, _IO_CHUNK_ROWS):
            f.write(index.reconstruct_n(s, min(_IO_CHUNK_ROWS, n - s)).tobytes())
        for s in range(0, n, _IO_CHUNK_ROWS):
            f.write(index.coarse_rows(s, min(_IO_CHUNK_ROWS, n - s)).tobytes())


def read_native_shards(dir_path: str, device: Optional[int] = None, coarse_dtype: str = "bf16", row_range=None) -> FlatIPIndex:
    """Rows [a, b) of the corpus (``row_range = (rank, world)``: that rank's contiguous share; None: everything) from whichever shard files hold
    them — the loading world size need not be the saving one.  The 16-bit copy is taken from the files when their dtype matches ``coarse_dtype``
    (bounds = the maximum over the files read, which is valid for any subset of their rows), otherwise the rows are re-quantised."""
    import json
    with open(os.path.join(dir_path, SHARD_MANIFEST)) as f:
        man = json.load(f)
    if man.get("format") != "krshard-1":
        raise ValueError(f"{dir_path}: unknown shard format {man.get('format')!r}")
    n, d = int(man["ntotal"]), int(man["d"])
    a, b = 0, n
    if row_range is not None:
        rank, world = row_range
        per = (n + world - 1) // world
        a, b = min(rank * per, n), min((rank + 1) * per, n)
    index = FlatIPIndex(d, device=device, coarse_dtype=coarse_dtype)
    index.reserve(b - a)
    raw = man["coarse_dtype"] == coarse_dtype
    want_code = {"bf16": 0, "f16": 1}[coarse_dtype]
    covered = a
    for sh in man["shards"]:
        r0, rows = int(sh["row0"]), int(sh["rows"])
        lo, hi = max(a, r0), min(b, r0 + rows)
        if lo >= hi:
            continue
        if lo != covered:
            raise ValueError(f"{dir_path}: rows [{covered}, {lo}) are in no shard file")
        with open(os.path.join(dir_path, sh["file"]), "rb") as f:
            magic, fd, fdc, fct, _, fr0, frows, fnt, b0, b1 = _SHARD_HEADER.unpack(f.read(_SHARD_HEADER.size))
            if magic != _SHARD_MAGIC or fd != d or fr0 != r0 or frows != rows or fnt != n:
                raise ValueError(f"{sh['file']}: header does not match the manifest")
            base_f = _SHARD_HEADER.size
            base_c = base_f + rows * d * 4
            for s0 in range(lo, hi, _IO_CHUNK_ROWS):
                m = min(_IO_CHUNK_ROWS, hi - s0)
                f.seek(base_f + (s0 - r0) * d * 4)
                xf = np.frombuffer(f.read(m * d * 4), dtype=np.float32).reshape(m, d)
                if raw and fdc == index.coarse_dim and fct == want_code:     # the FILE's own dtype code, not only the manifest's word
                    f.seek(base_c + (s0 - r0) * fdc * 2)
                    xc = np.frombuffer(f.read(m * fdc * 2), dtype=np.uint16).reshape(m, fdc)
                    index.add_raw(xf, xc, np.array([b0, b1], np.float32))
                else:
                    index.add(xf)
        covered = hi
    if covered != b:
        raise ValueError(f"{dir_path}: rows [{covered}, {b}) are in no shard file")
    index.file_ntotal, index.row_offset = n, a
    return index
