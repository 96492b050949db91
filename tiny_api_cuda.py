"""Drop-in alias for the reference's native module: ``from tiny_api_cuda import update_flatten_view``
(reference attention/kvcache.py:10; pybind11 module built from csrc/csrc/cuda_api.cu:113-116).  Same function name,
arguments, dtype errors (RuntimeError) and return value; the work is done by ``kvz_update_flatten_view`` of
``libkvzip_hip.so`` (include/kvzip_hip.h).  With the repo root on ``sys.path`` the reference's import line works unchanged."""
from kvzip_amd.ops import update_flatten_view  # noqa: F401

__all__ = ["update_flatten_view"]
