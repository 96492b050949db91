"""KV importance scoring and selection — host-side mirror of the reference's ``KVScore`` mixin
(reference attention/score.py:12-120): same method names, argument meaning and return values, with the
arithmetic done by the HIP kernels behind ``include/kvzip_hip.h``.

Differences that are deliberate (and invisible through the interface):
  * ``score[layer]`` are views into ONE preallocated ``[L, 1, Hkv, N]`` buffer that ``_get_score`` fills in
    place (the reference grows L tensors with ``torch.cat`` per chunk, score.py:33-34);
  * ``_threshold`` finds the order statistic with a radix histogram instead of a full sort (score.py:93);
  * ``_threshold_uniform`` breaks ties towards the lowest index (``torch.topk`` leaves it unspecified).
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch

from . import ops


class KVScore:
    """Functions to compute and threshold the score of KV pairs (mixed into the cache classes)."""

    def __init__(self):
        self.n_heads_kv = None
        self.n_layers = None
        self.dtype = None
        self.device = None
        self.get_score = True
        self.causal_mask_score = None  # kept for interface compatibility; the mask is applied inside the kernel
        self.score = None
        self.sink = None
        self.start_idx, self.end_idx = None, None
        self.ctx_len = None
        self._score_buf: Optional[torch.Tensor] = None
        self._score_fill: List[int] = []
        self._score_ws: Optional[torch.Tensor] = None

    # reference: attention/score.py:25-31
    def init_score(self):
        self.get_score = True
        self.causal_mask_score = None
        n = int(self.ctx_len) if self.ctx_len is not None else 0
        self._score_buf = torch.empty((self.n_layers, 1, self.n_heads_kv, n), dtype=self.dtype, device=self.device)
        self._score_fill = [0 for _ in range(self.n_layers)]
        self.score = [self._score_buf[l][:, :, :0] for l in range(self.n_layers)]

    # reference: attention/score.py:33-34
    def _update_score(self, layer_idx: int, score: torch.Tensor):
        """Append an externally computed ``[1, Hkv, m]`` block (kept for API compatibility)."""
        m = score.shape[-1]
        f = self._score_fill[layer_idx]
        self._ensure_score_capacity(f + m)
        self._score_buf[layer_idx][:, :, f:f + m].copy_(score)
        self._score_fill[layer_idx] = f + m
        self.score[layer_idx] = self._score_buf[layer_idx][:, :, :f + m]

    def _ensure_score_capacity(self, need: int):
        if self._score_buf.shape[-1] >= need:
            return
        new = torch.empty((self.n_layers, 1, self.n_heads_kv, need), dtype=self.dtype, device=self.device)
        old = self._score_buf.shape[-1]
        if old:
            new[..., :old].copy_(self._score_buf)
        self._score_buf = new
        self.score = [new[l][:, :, :self._score_fill[l]] for l in range(self.n_layers)]

    # reference: attention/score.py:36-65
    def _get_score(self, query_states: torch.Tensor, key_states: torch.Tensor, layer_idx: int):
        """query ``[1, H, q, D]``, key ``[1, Hkv, klen, D]`` (cache ++ repeat chunk).  Writes the chunk's
        ``[1, Hkv, end_idx-start_idx]`` scores straight into the layer's score buffer."""
        m = self.end_idx - self.start_idx
        f = self._score_fill[layer_idx]
        self._ensure_score_capacity(f + m)
        out = self._score_buf[layer_idx][:, :, f:f + m]
        bsz, H, q_len, D = query_states.shape
        need = ops._lib.load().kvz_score_workspace_bytes(self.n_heads_kv, H // self.n_heads_kv, q_len, m, self.sink)
        if self._score_ws is None or self._score_ws.numel() < need:
            self._score_ws = torch.empty(need, dtype=torch.uint8, device=query_states.device)
        ops.score_chunk(query_states, key_states, self.sink, self.start_idx, self.end_idx, out=out,
                        workspace=self._score_ws)
        self._score_fill[layer_idx] = f + m
        self.score[layer_idx] = self._score_buf[layer_idx][:, :, :f + m]

    # ------------------------------------------------------------------------------------------
    def _stacked_score(self, score) -> torch.Tensor:
        """``[L, 1, Hkv, N]`` tensor of the scores without copying when they already live in the buffer."""
        if isinstance(score, list):
            buf = self._score_buf
            if (buf is not None and len(score) == buf.shape[0] and all(
                    s.data_ptr() == buf[l].data_ptr() and s.shape[-1] == buf.shape[-1] for l, s in enumerate(score))):
                return buf
            return torch.stack(score, dim=0)
        return score

    # reference: attention/score.py:88-102
    def _threshold(self, score: Union[torch.Tensor, List[torch.Tensor]], ratio: float):
        """-> (valids bool like score, thres python float).  One global order statistic over all
        layers x heads x tokens; strict ``>`` (ties at the threshold are evicted)."""
        score = self._stacked_score(score)
        valid, thres, kept, rows = self._threshold_device(score, ratio)
        return valid, thres.item() if ratio < 1 else 0.

    def _threshold_device(self, score: torch.Tensor, ratio: float):
        """Device-side part of ``_threshold`` (no host sync): valid, thres[1], kept[1], row_counts[L*Hkv]."""
        if score.stride(-1) == 0 or not score.is_contiguous():
            score = score.contiguous()  # e.g. head-level scores expanded over the context (model/wrapper.py:56)
        return ops.select_threshold(score, ratio, row_len=score.shape[-1])

    # reference: attention/score.py:104-120
    def _threshold_uniform(self, scores: Union[torch.Tensor, List[torch.Tensor]], ratio: float):
        """-> (valids ``[L, 1, Hkv, N]`` bool, 0): exactly ``int(N*ratio)`` kept per (layer, head)."""
        score = self._stacked_score(scores if isinstance(scores, list) else list(scores))
        n_seq = score.shape[-1]
        k = int(n_seq * ratio) if ratio < 1 else n_seq
        valid, _ = ops.select_topk_rows(score, k)
        return valid, 0
