"""ctypes binding of the C-ABI library ``libkvzip_hip.so`` (declared in ``include/kvzip_hip.h``).

The product path has NO CPU fallback: if the HIP library is missing or a kernel reports an error the
call raises.  (The CPU oracle lives under ``oracle/`` and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  MUST precede loading libkvzip_hip.so: torch bundles its own libamdhip64.so.7 and both
# must share ONE HIP runtime (streams and device pointers cross the boundary).

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KVZIP_HIP_LIB", os.path.join(_HERE, "libkvzip_hip.so"))  # override: A/B builds

KVZ_F16, KVZ_BF16 = 0, 1
ABI_VERSION = 6

# name -> (restype, argtypes); mirrors include/kvzip_hip.h (+ the test hooks of include/kvzip_hip_debug.h) one to one
_vp, _i, _i64, _sz, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_double
SIGNATURES = {
    "kvz_abi_version": (_i, []),
    "kvz_last_error": (C.c_char_p, []),
    "kvz_prof_enable": (None, [_i]),
    "kvz_prof_reset": (None, []),
    "kvz_prof_read": (_i, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "kvz_score_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kvz_score_chunk": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "kvz_async_create": (_i, [_i]),
    "kvz_async_destroy": (_i, [_i]),
    "kvz_async_wait": (_i, [_i, _i, _vp]),
    "kvz_score_chunk_async": (_i, [_i, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _sz]),
    "kvz_score_chunk_log": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "kvz_score_chunk_async_log": (_i, [_i, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp,
                                       _sz]),
    "kvz_update_score_async_log": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i64, _i, _i, _i,
                                        _i, _i, _i, _i, _i, _vp, _i64, _vp, _sz]),
    "kvz_score_tail_flush": (_i, [_vp]),
    "kvz_score_tail_flush_async": (_i, [_i, _i, _vp, _vp]),
    "kvz_score_from_stats_log": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp]),
    "kvz_score_from_stats_async_log": (_i, [_i, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64]),
    "kvz_flash_fwd_window": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _i64, _i64, _i64, _i, _i, _i, _vp,
                                  _i64, _vp]),
    "kvz_score_log_fill": (_i, [_vp, _i64, _vp]),
    "kvz_score_finalize_log": (_i, [_vp, _i64, _vp, _i, _vp]),
    "kvz_score_finalize_log_hist": (_i, [_vp, _i64, _vp, _i, _vp, _sz, _vp]),
    "kvz_dense_append": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _vp]),
    "kvz_debug_round_chain": (_i, [_vp, _i, _i, _i, _i, _vp, C.POINTER(C.c_float), _vp]),
    "kvz_debug_fastdiv": (_i, [_i, _i, _vp, _vp]),
    "kvz_debug_set_tunable": (_i, [C.c_char_p, _i]),
    "kvz_debug_get_tunable": (_i, [C.c_char_p]),
    "kvz_debug_copy_kernel": (_i, [_vp, _vp, _sz, _i, _vp]),
    "kvz_debug_score_plan": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "kvz_select_workspace_bytes": (_sz, []),
    "kvz_select_threshold": (_i, [_vp, _i64, _d, _i, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kvz_select_threshold_prehist": (_i, [_vp, _i64, _d, _i, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kvz_select_topk_rows": (_i, [_vp, _i64, _i64, _i64, _i, _vp, _vp, _vp]),
    "kvz_select_heads": (_i, [_vp, _i, _i64, _d, _i, _vp, _vp, _vp, _vp, _vp]),
    "kvz_rowmax16": (_i, [_vp, _i64, _i64, _i, _vp, _vp]),
    "kvz_compact_plan_bytes": (_sz, [_i, _i, _i]),
    "kvz_compact_plan": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kvz_compact_layer": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "kvz_compact_layers": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "kvz_compact_plan_heads": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kvz_compact_layers_heads": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "kvz_update_flatten_view": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "kvz_append_inplace": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "kvz_varlen_attn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kvz_varlen_attn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _sz, _vp]),
    "kvz_varlen_attn_append": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp,
                                    _sz, _vp]),
    "kvz_add_i32": (_i, [_vp, _i, _vp]),
    "kvz_flash_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "kvz_flash_fwd": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _i64, _i64,
                           _i64, _vp, _vp, _sz, _vp]),
}

_lib = None


class KvzError(RuntimeError):
    """Raised when a C-ABI entry point returns a non-zero code (mirrors the reference's TORCH_CHECK
    RuntimeError, csrc/csrc/cuda_api.cu:73-74)."""


def load() -> C.CDLL:
    """Load the library once; fail loudly when it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KvzError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C kvzip_amd/csrc`). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    lib.kvz_abi_version.restype = C.c_int
    if lib.kvz_abi_version() != ABI_VERSION:   # (first: a stale library must fail with THIS message, not with a missing symbol)
        raise KvzError(f"ABI version mismatch: library {lib.kvz_abi_version()} != binding {ABI_VERSION} - rebuild {LIB_PATH}")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, who: str) -> None:
    if rc != 0:
        msg = load().kvz_last_error()
        raise KvzError(f"{who} failed (code {rc}): {msg.decode() if msg else ''}")
