"""CPU: the C-ABI library builds, loads and exports every symbol include/kvzip_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """Build the HIP library on demand (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(ROOT, "kvzip_amd", "libkvzip_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "kvzip_amd", "csrc"), "-j8"])
    assert os.path.exists(lib)


def _declared_symbols(headers=("kvzip_hip.h", "kvzip_hip_debug.h")):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(kvz_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_debug_hooks_live_in_their_own_header():
    assert not [s for s in _declared_symbols(("kvzip_hip.h",)) if s.startswith("kvz_debug_")]
    assert all(s.startswith("kvz_debug_") for s in _declared_symbols(("kvzip_hip_debug.h",)))


def test_header_declares_the_whole_path():
    syms = _declared_symbols()
    for need in ("kvz_score_chunk", "kvz_select_threshold", "kvz_select_topk_rows", "kvz_compact_plan",
                 "kvz_compact_layer", "kvz_compact_layers", "kvz_update_flatten_view", "kvz_append_inplace",
                 "kvz_varlen_attn", "kvz_last_error", "kvz_abi_version"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from kvzip_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/kvzip_hip.h but not exported"
    # and the ctypes binding covers exactly the declared set
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_host_only_entry_points():
    from kvzip_amd import _lib
    lib = _lib.load()
    assert lib.kvz_abi_version() == _lib.ABI_VERSION == 6
    assert lib.kvz_select_workspace_bytes() >= (2048 + 32) * 4
    assert lib.kvz_compact_plan_bytes(28, 4, 131104) == 28 * 4 * 129 * 4
    assert lib.kvz_score_workspace_bytes(4, 7, 2026, 2000, 32) >= 4 * 7 * 2026 * 8 + 4 * 2000 * 4
    assert lib.kvz_varlen_attn_workspace_bytes(4, 7, 1, 128, 131104) > 0
    assert lib.kvz_score_workspace_bytes(0, 7, 1, 1, 0) == 0
    # dense forward: from 16 (head, 256-row tile) units on the 32-row kernel takes the call when it gets the workspace of its split last
    # round - two partial slots (256 rows x (128 + 2) floats) per block of a round; below that the 16-row kernel's key splits
    slot = 256 * (128 + 2) * 4
    assert lib.kvz_flash_workspace_bytes(4, 7, 2026, 128) >= 2 * 256 * slot          # 224 units
    assert lib.kvz_flash_workspace_bytes(4, 7, 128, 128) >= 2 * 256 * slot           # 16 units
    assert 0 < lib.kvz_flash_workspace_bytes(4, 7, 100, 128) < 2 * 256 * slot        # 12 units: key splits of the 16-row kernel
    prev = lib.kvz_debug_set_tunable(b"flash2_split", 0)
    try:
        assert prev == 1 and lib.kvz_flash_workspace_bytes(4, 7, 2026, 128) == 0     # enough units by themselves, nothing to merge
    finally:
        lib.kvz_debug_set_tunable(b"flash2_split", prev)
    # asynchronous-scoring contexts are host objects (events): creating / destroying them needs no kernel
    assert lib.kvz_async_create(0) < 0 and b"slot" in lib.kvz_last_error()
    assert lib.kvz_async_wait(12345, -1, None) < 0 and lib.kvz_async_destroy(12345) == 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (error code + message)."""
    from kvzip_amd import _lib
    lib = _lib.load()
    rc = lib.kvz_select_threshold(None, 10, 0.3, 0, None, 10, None, None, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.kvz_last_error()
    rc = lib.kvz_varlen_attn(16, 16, 16, 16, 16, 0, None, None, 2, 7, 1, 96, 10, 0.1, 1, 0, 16, 16, 1 << 30, None)
    assert rc == -4 and b"head_dim" in lib.kvz_last_error()
    with pytest.raises(_lib.KvzError):
        _lib.check(rc, "kvz_varlen_attn")


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under kvzip_amd/ may import it."""
    pkg = os.path.join(ROOT, "kvzip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "kvzip_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f
