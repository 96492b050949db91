"""Tensor-level wrappers over the C ABI: torch is used for device memory and streams only.

Every function takes CUDA(=HIP) tensors, passes raw ``data_ptr()`` values plus the current HIP stream to
``libkvzip_hip.so`` and returns torch tensors that it allocated.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import math
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import KVZ_BF16, KVZ_F16, KvzError  # noqa: F401
from ._lib import check as _check

COMPACT_TILE = 1024


def _dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return KVZ_F16
    if dtype == torch.bfloat16:
        return KVZ_BF16
    raise KvzError(f"unsupported dtype {dtype}: the path computes in fp16 or bf16 "
                   "(reference csrc/csrc/static_switch.h:3-12)")


_tls = threading.local()  # .restore: device that was current before a call had to switch (given back by check() after the launch)


def _give_back_device() -> None:
    prev = getattr(_tls, "restore", None)
    if prev is not None:
        _tls.restore = None
        torch.cuda.set_device(prev)


def check(rc: int, who: str) -> None:
    """Raise on a non-zero return code; give the caller back the current device it had before the call (see ``_stream``)."""
    _give_back_device()
    _check(rc, who)


def _stream(t: torch.Tensor) -> int:
    """HIP stream handle for a launch on ``t``'s device.  The library launches on the CURRENT HIP device, so a tensor that
    lives on another GPU (a cache on cuda:1 while cuda:0 is current) makes its device current for the call; ``check`` - which
    follows every launch - switches back, so the caller's current device is what it was.  The pending switch is per host thread,
    and a switch that a failed call left behind (an exception between ``_stream`` and ``check``) is undone by the next call
    before it looks at the current device.  Every cache object and every call of this module works on ONE device (all tensors
    of a call must share it)."""
    if not t.is_cuda:
        raise KvzError("the HIP path needs device tensors (no CPU fallback)")
    _give_back_device()
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        _tls.restore = cur
        torch.cuda.set_device(t.device)
    return raw_stream(t.device.index)


try:  # the raw handle of the current stream without building a torch.cuda.Stream object (4 us -> 0.3 us per launch)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def raw_stream(device_index: int) -> int:
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _same_device(*tensors) -> None:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise KvzError(f"all tensors of a call must live on one device (got {dev} and {t.device})")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------------
# a1  scoring
# --------------------------------------------------------------------------------------------------
def score_chunk(query_states: torch.Tensor, key_states: torch.Tensor, sink: int, start: int, end: int,
                out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                stream: Optional["torch.cuda.Stream"] = None) -> torch.Tensor:
    """KV importance score of one layer / one chunk (reference attention/score.py:36-65).

    query_states ``[1, H, q, D]``, key_states ``[1, Hkv, klen, D]`` (rows contiguous; the head stride may be
    larger than ``klen*D`` — views into a cache with slack are fine).  Returns ``[1, Hkv, end-start]``.
    ``stream``: launch there instead of on the current stream (``out`` and ``workspace`` must then be given: nothing is
    allocated on a foreign stream).
    """
    lib = _lib.load()
    bsz, H, q_len, D = query_states.shape
    _, Hkv, klen, _ = key_states.shape
    assert bsz == 1, "batch size is 1 on this path (reference attention/score.py:29)"
    assert H % Hkv == 0
    G = H // Hkv
    m = end - start
    assert m > 0 and sink >= 0 and start >= sink and end <= klen - q_len, (sink, start, end, klen, q_len)
    dt = _dtype_code(query_states.dtype)
    assert key_states.dtype == query_states.dtype
    assert query_states.stride(-1) == 1 and query_states.stride(-2) == D
    assert key_states.stride(-1) == 1 and key_states.stride(-2) == D
    assert stream is None or (out is not None and workspace is not None)
    _same_device(query_states, key_states, out, workspace)
    if out is None:
        out = torch.empty((1, Hkv, m), dtype=query_states.dtype, device=query_states.device)
    assert out.stride(-1) == 1 and out.shape[-1] == m
    need = lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=query_states.device)
    rc = lib.kvz_score_chunk(query_states.data_ptr(), query_states.stride(1), key_states.data_ptr(),
                             key_states.stride(1), klen, sink, start, end, q_len, Hkv, G, D, dt,
                             out.data_ptr(), out.stride(1), workspace.data_ptr(), workspace.numel(),
                             _stream(query_states) if stream is None else stream.cuda_stream)
    check(rc, "kvz_score_chunk")
    return out


# --------------------------------------------------------------------------------------------------
# a4 / a5  selection
# --------------------------------------------------------------------------------------------------
def select_workspace(device) -> torch.Tensor:
    """Scratch of the global-threshold selection (two radix histograms); cleared by the library call that fills it."""
    return torch.empty(_lib.load().kvz_select_workspace_bytes(), dtype=torch.uint8, device=device)


def select_threshold(score: torch.Tensor, ratio: float, row_len: Optional[int] = None, prehist: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Global-threshold selection (reference attention/score.py:88-102) — device side only.

    Returns ``(valid bool like score, thres f32[1] (device), kept i64[1] (device), row_counts i32[rows] or None)``.
    No host synchronisation happens here.  ``prehist``: a selection workspace that already holds the first histogram of exactly
    these scores (``kvz_score_finalize_log_hist``, see ``KVScore._finalize_log``): one streaming pass less.
    """
    lib = _lib.load()
    assert prehist is None or score.is_contiguous()
    score = score.contiguous()
    n = score.numel()
    dt = _dtype_code(score.dtype)
    dev = score.device
    valid = torch.empty(score.shape, dtype=torch.bool, device=dev)
    thres = torch.empty(1, dtype=torch.float32, device=dev)
    kept = torch.empty(1, dtype=torch.int64, device=dev)
    rows = None
    if row_len is not None:
        rows = torch.empty(n // row_len, dtype=torch.int32, device=dev)
    ws = prehist if prehist is not None else select_workspace(dev)
    fn = lib.kvz_select_threshold_prehist if (prehist is not None and ratio < 1) else lib.kvz_select_threshold
    rc = fn(score.data_ptr(), n, float(ratio), dt, valid.data_ptr(), row_len if row_len is not None else n, _ptr(rows),
            thres.data_ptr(), kept.data_ptr(), ws.data_ptr(), ws.numel(), _stream(score))
    check(rc, "kvz_select_threshold")
    return valid, thres, kept, rows


def select_heads(head_scores: torch.Tensor, N: int, ratio: float
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Head-level selection (reference model/wrapper.py:40-58 + attention/score.py:88-102) on the ``[L, Hkv]`` head scores
    themselves: the threshold of the N-fold expanded tensor is the head value of rank ``idx // N``.
    Returns ``(valid_heads bool [L, Hkv], thres f32[1], kept i64[1] (= N * kept heads), row_counts i32[L*Hkv])``."""
    lib = _lib.load()
    hs = head_scores.contiguous()
    rows = hs.numel()
    dev = hs.device
    valid = torch.empty(hs.shape, dtype=torch.bool, device=dev)
    thres = torch.empty(1, dtype=torch.float32, device=dev)
    kept = torch.empty(1, dtype=torch.int64, device=dev)
    counts = torch.empty(rows, dtype=torch.int32, device=dev)
    rc = lib.kvz_select_heads(hs.data_ptr(), rows, int(N), float(ratio), _dtype_code(hs.dtype), valid.data_ptr(),
                              counts.data_ptr(), thres.data_ptr(), kept.data_ptr(), _stream(hs))
    check(rc, "kvz_select_heads")
    return valid, thres, kept, counts


def rowmax(score: torch.Tensor) -> torch.Tensor:
    """Maximum over the last dim of 16-bit scores (head-score production, reference test.py:22-25)."""
    lib = _lib.load()
    score = score.contiguous()
    row_len = score.shape[-1]
    rows = score.numel() // row_len
    out = torch.empty(score.shape[:-1], dtype=score.dtype, device=score.device)
    rc = lib.kvz_rowmax16(score.data_ptr(), rows, row_len, _dtype_code(score.dtype), out.data_ptr(), _stream(score))
    check(rc, "kvz_rowmax16")
    return out


def select_topk_rows(score: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-row exact top-k over the last dim (reference attention/score.py:104-120)."""
    lib = _lib.load()
    score = score.contiguous()
    row_len = score.shape[-1]
    rows = score.numel() // row_len
    valid = torch.empty(score.shape, dtype=torch.bool, device=score.device)
    counts = torch.empty(rows, dtype=torch.int32, device=score.device)
    rc = lib.kvz_select_topk_rows(score.data_ptr(), rows, row_len, int(k), _dtype_code(score.dtype),
                                  valid.data_ptr(), counts.data_ptr(), _stream(score))
    check(rc, "kvz_select_topk_rows")
    return valid, counts


# --------------------------------------------------------------------------------------------------
# a8 / a9  compaction
# --------------------------------------------------------------------------------------------------
class CompactPlan:
    """Device-side result of ``kvz_compact_plan`` for all layers (reference attention/kvcache.py:168-185)."""

    def __init__(self, layers: int, Hkv: int, N: int, sink: int, klen: int, slack: int, device):
        self.layers, self.Hkv, self.N, self.sink, self.klen, self.slack = layers, Hkv, N, sink, klen, slack
        self.ntiles = (klen + COMPACT_TILE - 1) // COMPACT_TILE
        i32 = dict(dtype=torch.int32, device=device)
        # one allocation so that a single D2H copy brings all the metadata to the host
        meta = torch.empty(layers * Hkv * 2 + layers * (Hkv + 1) + layers, **i32)
        o = 0
        self.len_k = meta[o:o + layers * Hkv].view(layers, Hkv); o += layers * Hkv
        self.seg_start = meta[o:o + layers * Hkv].view(layers, Hkv); o += layers * Hkv
        self.cu_len_k = meta[o:o + layers * (Hkv + 1)].view(layers, Hkv + 1); o += layers * (Hkv + 1)
        self.max_len_k = meta[o:o + layers]
        self.meta = meta
        self.tile_base = torch.empty(layers * Hkv * self.ntiles, **i32)


def is_head_level(valid: torch.Tensor) -> bool:
    """True for a head-level mask: one value per (layer, head) broadcast over the context (stride 0 on the last dim)."""
    return valid.dim() >= 2 and valid.shape[-1] > 1 and valid.stride(-1) == 0


def compact_plan(valid: torch.Tensor, sink: int, klen: int, slack: int = 0) -> CompactPlan:
    """valid ``[L, 1, Hkv, N]`` (or ``[L, Hkv, N]``) bool -> per-layer varlen metadata on the device.
    A head-level mask (``[L, 1, Hkv, 1]`` expanded over N, stride 0) is planned from its L*Hkv bytes."""
    lib = _lib.load()
    L = valid.shape[0]
    Hkv, N = valid.shape[-2], valid.shape[-1]
    heads = is_head_level(valid)
    valid = valid[..., :1].contiguous() if heads else valid.contiguous()
    plan = CompactPlan(L, Hkv, N, sink, klen, slack, valid.device)
    fn = lib.kvz_compact_plan_heads if heads else lib.kvz_compact_plan
    rc = fn(valid.data_ptr(), L, Hkv, N, sink, klen, slack, plan.len_k.data_ptr(),
            plan.cu_len_k.data_ptr(), plan.seg_start.data_ptr(), plan.max_len_k.data_ptr(),
            plan.tile_base.data_ptr(), _stream(valid))
    check(rc, "kvz_compact_plan_heads" if heads else "kvz_compact_plan")
    plan.valid = valid
    plan.heads = heads
    return plan


def compact_layer(k: torch.Tensor, v: torch.Tensor, plan: CompactPlan, layer: int, total_rows: int
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gather one layer: k, v ``[1, Hkv, klen, D]`` -> ``[total_rows, D]`` each (order preserving, head major)."""
    lib = _lib.load()
    _, Hkv, klen, D = k.shape
    assert klen == plan.klen and Hkv == plan.Hkv
    assert k.stride(-1) == 1 and k.stride(-2) == D and v.stride() == k.stride()
    k_out = torch.empty((total_rows, D), dtype=k.dtype, device=k.device)
    v_out = torch.empty((total_rows, D), dtype=v.dtype, device=v.device)
    if total_rows == 0:
        return k_out, v_out
    assert not getattr(plan, "heads", False), "head-level plans go through compact_layers"
    valid_l = plan.valid.view(plan.layers, Hkv, plan.N)[layer]
    tb = plan.tile_base.view(plan.layers, Hkv * plan.ntiles)[layer]
    rc = lib.kvz_compact_layer(k.data_ptr(), v.data_ptr(), k.stride(1), valid_l.data_ptr(), tb.data_ptr(),
                               plan.seg_start[layer].data_ptr(), Hkv, plan.N, plan.sink, klen, D,
                               k.element_size(), k_out.data_ptr(), v_out.data_ptr(), _stream(k))
    check(rc, "kvz_compact_layer")
    return k_out, v_out


def compact_layers(ks: Sequence[torch.Tensor], vs: Sequence[torch.Tensor], plan: CompactPlan,
                   totals: Sequence[int]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """All layers in ONE launch (pointer tables).  ``totals[l]`` = rows to allocate for layer l."""
    lib = _lib.load()
    L = len(ks)
    assert L == plan.layers
    _, Hkv, klen, D = ks[0].shape
    dev = ks[0].device
    hs = ks[0].stride(1)
    for k, v in zip(ks, vs):
        assert k.shape == ks[0].shape and v.shape == ks[0].shape
        assert k.stride(-1) == 1 and k.stride(-2) == D and k.stride(1) == hs and v.stride() == k.stride()
    # (an all-evicted layer still gets a 1-row buffer so that its table entry is a valid pointer)
    k_outs = [torch.empty((max(int(t), 1), D), dtype=ks[0].dtype, device=dev)[:int(t)] for t in totals]
    v_outs = [torch.empty((max(int(t), 1), D), dtype=ks[0].dtype, device=dev)[:int(t)] for t in totals]
    table = torch.tensor([[t.data_ptr() for t in ks], [t.data_ptr() for t in vs],
                          [t.data_ptr() for t in k_outs], [t.data_ptr() for t in v_outs]],
                         dtype=torch.int64).to(dev, non_blocking=False)
    fn = lib.kvz_compact_layers_heads if getattr(plan, "heads", False) else lib.kvz_compact_layers
    rc = fn(table[0].data_ptr(), table[1].data_ptr(), hs, plan.valid.data_ptr(),
            plan.tile_base.data_ptr(), plan.seg_start.data_ptr(), L, Hkv, plan.N, plan.sink,
            klen, D, ks[0].element_size(), table[2].data_ptr(), table[3].data_ptr(),
            _stream(ks[0]))
    check(rc, "kvz_compact_layers")
    # keep the pointer table alive until the kernel has consumed it
    table.record_stream(torch.cuda.current_stream(dev))
    return k_outs, v_outs


# --------------------------------------------------------------------------------------------------
# a10 / a11  append
# --------------------------------------------------------------------------------------------------
def update_flatten_view(cache: torch.Tensor, state: torch.Tensor, headlens: torch.Tensor,
                        cu_headlens: torch.Tensor) -> torch.Tensor:
    """Drop-in for ``tiny_api_cuda.update_flatten_view`` (reference csrc/csrc/cuda_api.cu:68-111):
    same arguments, same dtype checks (-> RuntimeError), returns a fresh tensor."""
    lib = _lib.load()
    if headlens.dtype != torch.int32:
        raise KvzError("expected headlens to be int32")
    if cu_headlens.dtype != torch.int32:
        raise KvzError("expected cu_headlens to be int32")
    origin_len, dim = cache.shape
    head_num = headlens.shape[0]
    if state.shape[0] % head_num != 0:
        raise KvzError("state rows must be divisible by head count")
    t = state.shape[0] // head_num
    cache = cache.contiguous()
    state = state.contiguous()
    out = torch.empty((origin_len + head_num * t, dim), dtype=cache.dtype, device=cache.device)
    rc = lib.kvz_update_flatten_view(cache.data_ptr(), state.data_ptr(), headlens.data_ptr(),
                                     cu_headlens.data_ptr(), head_num, t, dim, cache.element_size(),
                                     out.data_ptr(), _stream(cache))
    check(rc, "kvz_update_flatten_view")
    return out


def append_inplace(k_cache: torch.Tensor, v_cache: torch.Tensor, k_state: torch.Tensor, v_state: torch.Tensor,
                   seg_start: torch.Tensor, base_len: torch.Tensor, len_offset: int = 0) -> None:
    """O(t) append of ``k_state/v_state [1, Hkv, t, D]`` after each head's current rows (slack layout)."""
    lib = _lib.load()
    _, Hkv, t, D = k_state.shape
    assert k_state.stride(-1) == 1 and v_state.stride(-1) == 1 and v_state.shape == k_state.shape
    rc = lib.kvz_append_inplace(k_cache.data_ptr(), v_cache.data_ptr(), k_state.data_ptr(), v_state.data_ptr(),
                                k_state.stride(1), k_state.stride(2), v_state.stride(1), v_state.stride(2),
                                seg_start.data_ptr(), base_len.data_ptr(), int(len_offset), Hkv, t, D,
                                k_cache.element_size(), _stream(k_cache))
    check(rc, "kvz_append_inplace")


# --------------------------------------------------------------------------------------------------
# a13  variable-length attention
# --------------------------------------------------------------------------------------------------
def attn_workspace(Hkv: int, G: int, q_len: int, D: int, device) -> torch.Tensor:
    """Scratch of the attention kernels: partial results of the key ranges (decode kernel) or of the key splits (multi-row kernel).
    No initialisation is required; it is zero-filled once anyway (a reserved counter region of the layout)."""
    need = _lib.load().kvz_varlen_attn_workspace_bytes(Hkv, G, q_len, D, 0)
    return torch.zeros(max(int(need), 16), dtype=torch.uint8, device=device)


def _meta_host(k_start_host, k_len_host, Hkv):
    """ctypes int32 array [starts ++ lens] for the kernel arguments, or None."""
    if k_start_host is None or k_len_host is None or Hkv > 64:
        return None
    import ctypes as C
    return (C.c_int32 * (2 * Hkv))(*k_start_host, *k_len_host)


def varlen_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, k_start: torch.Tensor, k_len: torch.Tensor,
                q_len: int, max_len_k: int, causal: bool = True, softmax_scale: Optional[float] = None,
                workspace: Optional[torch.Tensor] = None, k_len_offset: int = 0, meta_host=None,
                offset_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q ``[Hkv*q_len, G, D]``; k, v ``[rows, D]`` (or ``[rows, 1, D]``); head h owns rows
    ``k_start[h] : k_start[h]+k_len[h]``.  Returns ``[Hkv*q_len, G, D]``.  ``meta_host``: ctypes array from
    ``_meta_host`` (the segments as the host knows them) or None; ``workspace`` from ``attn_workspace``."""
    lib = _lib.load()
    HQ, G, D = q.shape
    Hkv = HQ // q_len
    assert Hkv * q_len == HQ and k_start.shape[0] == Hkv and k_len.shape[0] == Hkv
    assert k_start.dtype == torch.int32 and k_len.dtype == torch.int32
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    if out is None:
        out = torch.empty_like(q)
    if workspace is None:  # (a caller-provided workspace is validated by the library itself: KVZ_EWORKSPACE)
        workspace = attn_workspace(Hkv, G, q_len, D, q.device)
    rc = lib.kvz_varlen_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), k_start.data_ptr(), k_len.data_ptr(),
                             int(k_len_offset), _ptr(offset_dev), meta_host, Hkv, G,
                             q_len, D, int(max_len_k), float(scale), 1 if causal else 0, _dtype_code(q.dtype),
                             out.data_ptr(), workspace.data_ptr(), workspace.numel(), _stream(q))
    check(rc, "kvz_varlen_attn")
    return out


def varlen_attn_append(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, k_state: torch.Tensor,
                       v_state: torch.Tensor, k_start: torch.Tensor, k_len: torch.Tensor, k_len_offset: int, max_len_k: int,
                       softmax_scale: Optional[float] = None, workspace: Optional[torch.Tensor] = None,
                       meta_host=None, offset_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode step in one launch: append the new token's K, V (``[1, Hkv, 1, D]``, any head stride) to the per-head slack
    of the flat cache and attend ``q`` (``[Hkv, G, D]``) over ``k_len + k_len_offset + 1`` keys.  ``k_len_offset`` counts the
    tokens appended BEFORE this call.  Bit-identical to ``append_inplace`` followed by ``varlen_attn``."""
    lib = _lib.load()
    Hkv, G, D = q.shape
    assert k_state.shape[-3] == Hkv and k_state.shape[-2] == 1 and v_state.shape[-2] == 1
    assert k_state.stride(-1) == 1 and v_state.stride(-1) == 1 and q.is_contiguous()
    assert k_start.dtype == torch.int32 and k_len.dtype == torch.int32
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    if out is None:
        out = torch.empty_like(q)
    if workspace is None:
        workspace = attn_workspace(Hkv, G, 1, D, q.device)
    rc = lib.kvz_varlen_attn_append(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_state.data_ptr(),
                                    v_state.data_ptr(), k_state.stride(-3), v_state.stride(-3), k_start.data_ptr(),
                                    k_len.data_ptr(), int(k_len_offset), _ptr(offset_dev), meta_host, Hkv, G, D, int(max_len_k), float(scale),
                                    _dtype_code(q.dtype), out.data_ptr(), workspace.data_ptr(), workspace.numel(), _stream(q))
    check(rc, "kvz_varlen_attn_append")
    return out


def flash_fwd(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, causal: bool = True,
              softmax_scale: Optional[float] = None, return_lse: bool = False):
    """Dense GQA attention ``[1, H, q, D] x [1, Hkv, k, D] -> [1, q, H, D]`` with the causal mask aligned bottom-right
    (reference attention/attn.py:75-89, flash-attn's dense kernel).  ``key`` / ``value`` may be views of the dense cache
    (any head stride, rows contiguous); nothing is copied.  ``return_lse``: also the fp32 row LSE ``[1, H, q]``."""
    lib = _lib.load()
    b, H, q_len, D = query.shape
    Hkv, klen = key.shape[1], key.shape[2]
    G = H // Hkv
    assert b == 1 and key.shape[0] == 1 and G * Hkv == H and value.shape == key.shape
    if query.stride(-1) != 1 or query.stride(1) % 8 or query.stride(2) % 8:
        query = query.contiguous()

    def rows(t):  # [1, Hkv, k, D] view -> usable as "[rows, D], head h starts at row h * step"
        return t.stride(-1) == 1 and t.stride(2) == D and t.stride(1) % D == 0
    if not (rows(key) and rows(value) and key.stride(1) == value.stride(1)):
        key, value = key.contiguous(), value.contiguous()
    step = key.stride(1) // D if Hkv > 1 else klen
    meta = _meta_host([h * step for h in range(Hkv)], [klen] * Hkv, Hkv)
    assert meta is not None, "flash_fwd: more than 64 KV heads"
    out = torch.empty(1, q_len, H, D, dtype=query.dtype, device=query.device)
    lse = torch.empty(1, Hkv, q_len, G, dtype=torch.float32, device=query.device) if return_lse else None
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    need = int(lib.kvz_flash_workspace_bytes(Hkv, G, q_len, D))  # > 0 only when there are too few query rows to fill the chip
    ws = torch.empty(need, dtype=torch.uint8, device=query.device) if need else None
    rc = lib.kvz_flash_fwd(query.data_ptr(), G * query.stride(1), query.stride(1), query.stride(2), key.data_ptr(),
                           value.data_ptr(), None, None, 0, meta, Hkv, G, q_len, D, float(scale), 1 if causal else 0,
                           _dtype_code(query.dtype), out.data_ptr(), G * D, D, H * D, _ptr(lse), _ptr(ws), need,
                           _stream(query))
    check(rc, "kvz_flash_fwd")
    if return_lse:
        return out, lse.permute(0, 1, 3, 2).reshape(1, H, q_len)
    return out


def flash_fwd_window(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, sink: int, start: int, end: int,
                     stats: torch.Tensor, softmax_scale: Optional[float] = None) -> Optional[torch.Tensor]:
    """The dense causal forward of a scoring pass (``flash_fwd``) that also writes the row statistics of ``KVScore._get_score`` for the
    window ``sink ++ [start, end) ++ last q_len keys`` into ``stats`` (``[Hkv, stride, 2]`` fp32: (m_r, l'_r), row ``g*q_len + i``)
    from its own QK^T tiles.  Returns ``[1, q, H, D]``, or None when the 32-row kernel does not take the shape (the caller then
    scores with the two-pass kernels)."""
    lib = _lib.load()
    b, H, q_len, D = query.shape
    Hkv, klen = key.shape[1], key.shape[2]
    G = H // Hkv
    if (b != 1 or D != 128 or query.stride(-1) != 1 or query.stride(1) % 8 or query.stride(2) % 8 or key.stride(-1) != 1
            or key.stride(2) != D or key.stride(1) % D or value.stride(-1) != 1 or value.stride(2) != D or key.stride(1) != value.stride(1)):
        return None
    step = key.stride(1) // D if Hkv > 1 else klen
    meta = _meta_host([h * step for h in range(Hkv)], [klen] * Hkv, Hkv)
    if meta is None:
        return None
    out = torch.empty(1, q_len, H, D, dtype=query.dtype, device=query.device)
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    rc = lib.kvz_flash_fwd_window(query.data_ptr(), G * query.stride(1), query.stride(1), query.stride(2), key.data_ptr(),
                                  value.data_ptr(), meta, Hkv, G, q_len, D, float(scale), _dtype_code(query.dtype), out.data_ptr(),
                                  G * D, D, H * D, int(sink), int(start), int(end), stats.data_ptr(), stats.stride(0) // 2, _stream(query))
    if rc == -4:  # KVZ_EUNSUPPORTED: not a shape of the 32-row kernel
        check(0, "kvz_flash_fwd_window")
        return None
    check(rc, "kvz_flash_fwd_window")
    return out


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, seqused_k=None):
    """Call-compatible stand-in for the reference's use of ``flash_attn.flash_attn_varlen_func``
    (reference attention/attn.py:61-71): every "sequence" is one KV head with ``nheads_k = 1``."""
    assert dropout_p == 0.0
    Hkv = cu_seqlens_k.shape[0] - 1
    q_len = int(max_seqlen_q)
    k2 = k.view(-1, k.shape[-1])
    v2 = v.view(-1, v.shape[-1])
    k_start = cu_seqlens_k[:-1].contiguous()
    k_len = (cu_seqlens_k[1:] - cu_seqlens_k[:-1]).contiguous() if seqused_k is None else seqused_k
    assert q.shape[0] == Hkv * q_len
    return varlen_attn(q, k2, v2, k_start, k_len, q_len, int(max_seqlen_k), causal=causal,
                       softmax_scale=softmax_scale)
