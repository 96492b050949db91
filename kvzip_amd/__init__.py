"""kvzip_amd — MI355X (gfx950) native implementation of KVzip's KV-eviction hot path.

score (attention/score.py) -> select -> compact -> append -> variable-length attention
(attention/kvcache.py, csrc/) behind the reference's ``ModelKVzip / kv.prune(ratio) / generate(kv=...)`` surface.
All compute runs in hand-written HIP kernels (``kvzip_amd/csrc``) reached through the C ABI in
``include/kvzip_hip.h``; there is no CPU fallback.
"""
from ._lib import KvzError, LIB_PATH  # noqa: F401

__all__ = ["KvzError", "LIB_PATH"]
