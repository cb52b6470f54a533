"""ctypes binding of libkirag_amd.so (include/kirag_amd.h).

The library is built in-tree by ``kirag_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no
CPU fallback anywhere in this package: if the shared object is missing or a call fails, an exception is
raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KIRAG_AMD_LIB") or os.path.join(_HERE, "libkirag_amd.so")   # KIRAG_AMD_LIB: diagnostic builds (tools/stamp_build.sh)
ABI_VERSION = 9


class KiragAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libkirag_amd error {code}: {msg}")
        self.code = code


class SearchStats(C.Structure):
    _fields_ = [("queries", C.c_int64), ("certified", C.c_int64), ("fallback", C.c_int64), ("overflow", C.c_int64),
                ("reranked_rows", C.c_int64), ("coarse_rounds", C.c_int64), ("last_coarse_ms", C.c_double),
                ("last_total_ms", C.c_double), ("fine", C.c_int64), ("exact", C.c_int64), ("fine_rounds", C.c_int64),
                ("last_fine_ms", C.c_double), ("marked_passes", C.c_int64), ("marked_rows", C.c_int64),
                ("va_retired_bytes", C.c_int64), ("grow_mode", C.c_int64), ("byte_scans", C.c_int64), ("byte_marked_rows", C.c_int64), ("byte_rows", C.c_int64)]


class BertCfg(C.Structure):
    _fields_ = [("hidden", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("intermediate", C.c_int),
                ("vocab", C.c_int), ("max_pos", C.c_int), ("type_vocab", C.c_int), ("ln_eps", C.c_float)]


# name -> (restype, argtypes); every symbol include/kirag_amd.h declares
SIGNATURES = {
    "kr_abi_version": (C.c_int, []),
    "kr_last_error": (C.c_char_p, []),
    "kr_device_count": (C.c_int, []),
    "kr_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "kr_release_scratch": (None, []),
    "kr_index_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kr_index_destroy": (None, [C.c_void_p]),
    "kr_index_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "kr_index_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "kr_index_ntotal": (C.c_int64, [C.c_void_p]),
    "kr_index_dim": (C.c_int, [C.c_void_p]),
    "kr_index_get_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "kr_index_coarse_dim": (C.c_int, [C.c_void_p]),
    "kr_index_coarse_dtype": (C.c_int, [C.c_void_p]),
    "kr_index_get_coarse": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "kr_index_get_bounds": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kr_index_add_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "kr_index_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "kr_index_search_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kr_index_search_coarse_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_index_search_global_theta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_index_search_rerank_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kr_index_search_finish": (C.c_int, [C.c_void_p]),
    "kr_index_search_finish_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]),
    "kr_index_search_finish_one": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "kr_index_search_pending": (C.c_int, [C.c_void_p]),
    "kr_index_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "kr_index_stats": (C.c_int, [C.c_void_p, C.POINTER(SearchStats), C.c_int]),
    "kr_score_topk": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "kr_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_format_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_char, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "kr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kr_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kr_comm_destroy": (C.c_int, [C.c_void_p]),
    "kr_comm_rank": (C.c_int, [C.c_void_p]),
    "kr_comm_world": (C.c_int, [C.c_void_p]),
    "kr_shard_allgather_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kr_topk_merge_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "kr_encoder_create": (C.c_int, [C.POINTER(BertCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "kr_encoder_create_ex": (C.c_int, [C.POINTER(BertCfg), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kr_encoder_operand_dtype": (C.c_int, [C.c_void_p]),
    "kr_encoder_residual_lo": (C.c_int, [C.c_void_p]),
    "kr_encoder_destroy": (None, [C.c_void_p]),
    "kr_encoder_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "kr_encoder_finalize": (C.c_int, [C.c_void_p]),
    "kr_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_encoder_forward_tt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_encoder_forward_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "kr_encoder_last_hidden": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "kr_encoder_check": (C.c_int, [C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared object (once).  Raises if it is missing — the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own libamdhip64.so.7; whichever HIP runtime is loaded first serves the whole
    # process (same SONAME).  Import torch FIRST so that its runtime is the one: tensors handed to this library
    # and this library's own allocations then live in one runtime.  (Loading this library first made torch fail
    # with "no ROCm-capable device is detected".)  A process that never imports torch uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C kirag_amd/csrc` (or __graft_entry__.build()). "
            "kirag_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # KIRAG_AMD_LIB_OLDER=1 (tools' same-box A/Bs against an earlier round's library through KIRAG_AMD_LIB): symbols the older build lacks are left
    # unbound and its ABI number is accepted; never set for the product library
    older = bool(os.environ.get("KIRAG_AMD_LIB")) and os.environ.get("KIRAG_AMD_LIB_OLDER") == "1"
    for name, (res, args) in SIGNATURES.items():
        if older and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the build does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.kr_abi_version()
    if v != ABI_VERSION and not (older and v < ABI_VERSION):
        raise ImportError(f"libkirag_amd ABI {v} != expected {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().kr_last_error()
        raise KiragAmdError(rc, msg.decode("utf-8", "replace") if msg else "")


def current_stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (0 = default stream)."""
    import torch
    if not torch.cuda.is_available():
        return 0
    return int(torch.cuda.current_stream().cuda_stream)
