"""KV importance scoring and selection — host-side mirror of the reference's ``KVScore`` mixin
(reference attention/score.py:12-120): same method names, argument meaning and return values, with the
arithmetic done by the HIP kernels behind ``include/kvzip_hip.h``.

Differences that are deliberate (and invisible through the interface):
  * ``score[layer]`` are views into ONE preallocated ``[L, 1, Hkv, N]`` buffer that ``_get_score`` fills in
    place (the reference grows L tensors with ``torch.cat`` per chunk, score.py:33-34);
  * ``_threshold`` finds the order statistic with a radix histogram instead of a full sort (score.py:93);
  * ``_threshold_uniform`` breaks ties towards the lowest index (``torch.topk`` leaves it unspecified);
  * ``_get_score`` is ASYNCHRONOUS with respect to the caller's stream: the scores of a layer are a side product that
    nothing in the forward pass consumes, so the kernels of consecutive layers are issued round-robin on
    ``n_score_streams`` (default 2) side streams.  The tail of one persistent kernel, the launch gaps and the two tiny
    merge / finalize kernels of a call then overlap with the big kernels of the next one (+16 % scoring throughput on
    MI355X).  Ordering is kept with events: the side stream waits for the caller's stream (inputs), ``update`` of a layer
    waits for that layer's previous scoring call (it overwrites the rows that call read), and reading ``.score``,
    thresholding or pruning waits for everything outstanding.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch

from . import ops


# Side streams are shared by all cache objects of a process and are CHOSEN BY MEASUREMENT: HIP maps streams onto a handful of
# hardware queues (4 by default) in a way the API does not expose, and two streams that land on the same queue do not
# overlap at all (observed: every other freshly created pair).  Candidates of both priorities are created once and a pair is
# accepted when two spin kernels launched on it really run concurrently.
_SIDE_STREAMS = {}


def _overlaps(a: "torch.cuda.Stream", b: "torch.cuda.Stream", cycles: int = 400_000) -> bool:
    """True if spin kernels on the two streams run concurrently (wall time well below twice one kernel)."""
    def timed(streams):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1)
    timed([a])  # warm-up (first launch on a stream creates its queue)
    timed([b])
    one = min(timed([a]), timed([b]))
    both = timed([a, b])
    return both < 1.5 * one


def _side_streams(device, n: int):
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _SIDE_STREAMS.get(key)
    if pool is not None and len(pool) >= n:
        return pool[:n]
    with torch.cuda.device(key):
        cands = [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1)]
        cands += [torch.cuda.Stream(device=dev) for _ in range(4)]
        chosen = []
        try:
            for c in cands:
                if all(_overlaps(c, o) for o in chosen):
                    chosen.append(c)
                if len(chosen) >= n:
                    break
        except Exception:  # no spin kernel in this build: take the candidates as they come
            chosen = cands[:n]
        while len(chosen) < n:  # fewer independent queues than requested: the extra streams simply do not overlap
            chosen.append(chosen[-1] if chosen else cands[0])
    _SIDE_STREAMS[key] = chosen
    return chosen[:n]


class KVScore:
    """Functions to compute and threshold the score of KV pairs (mixed into the cache classes)."""

    def __init__(self):
        self.n_heads_kv = None
        self.n_layers = None
        self.dtype = None
        self.device = None
        self.get_score = True
        self.causal_mask_score = None  # kept for interface compatibility; the mask is applied inside the kernel
        self._score = None
        self.sink = None
        self.start_idx, self.end_idx = None, None
        self.ctx_len = None
        self._score_buf: Optional[torch.Tensor] = None
        self._score_fill: List[int] = []
        self._score_ws: List[Optional[torch.Tensor]] = []
        self.n_score_streams = 2       # 1 = score on the caller's stream
        self._score_exclusive = False  # True: the next calls run alone on the caller's stream (clean kernel timings)
        self._score_side: List["torch.cuda.Stream"] = []
        self._score_events = {}        # layer -> event recorded after its latest scoring call
        self._score_ev_pool = {}       # layer -> reusable event object

    # ---- asynchronous scoring: bookkeeping -----------------------------------------------------------------
    @property
    def score(self):
        """Per-layer scores (reference attribute).  Reading it orders the caller's stream behind outstanding scoring."""
        self._wait_score()
        return self._score

    @score.setter
    def score(self, value):
        self._score = value

    def _wait_score(self, layer_idx: Optional[int] = None):
        """Make the current stream wait for the scoring calls still in flight (of one layer, or of all)."""
        if not self._score_events:
            return
        cur = torch.cuda.current_stream()
        if layer_idx is None:
            for ev in self._score_events.values():
                cur.wait_event(ev)
            self._score_events = {}
        else:
            ev = self._score_events.pop(layer_idx, None)
            if ev is not None:
                cur.wait_event(ev)

    # reference: attention/score.py:25-31
    def init_score(self):
        self.get_score = True
        self.causal_mask_score = None
        n = int(self.ctx_len) if self.ctx_len is not None else 0
        self._score_buf = torch.empty((self.n_layers, 1, self.n_heads_kv, n), dtype=self.dtype, device=self.device)
        self._wait_score()
        self._score_fill = [0 for _ in range(self.n_layers)]
        self._score = [self._score_buf[l][:, :, :0] for l in range(self.n_layers)]

    # reference: attention/score.py:33-34
    def _update_score(self, layer_idx: int, score: torch.Tensor):
        """Append an externally computed ``[1, Hkv, m]`` block (kept for API compatibility)."""
        m = score.shape[-1]
        f = self._score_fill[layer_idx]
        self._ensure_score_capacity(f + m)
        self._wait_score(layer_idx)
        self._score_buf[layer_idx][:, :, f:f + m].copy_(score)
        self._score_fill[layer_idx] = f + m
        self._score[layer_idx] = self._score_buf[layer_idx][:, :, :f + m]

    def _ensure_score_capacity(self, need: int):
        if self._score_buf.shape[-1] >= need:
            return
        self._wait_score()
        new = torch.empty((self.n_layers, 1, self.n_heads_kv, need), dtype=self.dtype, device=self.device)
        old = self._score_buf.shape[-1]
        if old:
            new[..., :old].copy_(self._score_buf)
        self._score_buf = new
        self._score = [new[l][:, :, :self._score_fill[l]] for l in range(self.n_layers)]

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
        nstreams = 1 if (self._score_exclusive or not query_states.is_cuda) else max(1, int(self.n_score_streams))
        slot = layer_idx % nstreams if nstreams > 1 else 0
        while len(self._score_ws) <= slot:
            self._score_ws.append(None)
        if self._score_ws[slot] is None or self._score_ws[slot].numel() < need:
            self._wait_score()  # (the old workspace of this slot may still be in use)
            self._score_ws[slot] = torch.empty(need, dtype=torch.uint8, device=query_states.device)
        if nstreams == 1:
            self._wait_score()  # the caller's stream: everything before it is ordered anyway, later calls wait for it
            ops.score_chunk(query_states, key_states, self.sink, self.start_idx, self.end_idx, out=out,
                            workspace=self._score_ws[0])
        else:
            self._score_side = _side_streams(query_states.device, nstreams)
            side, cur = self._score_side[slot], torch.cuda.current_stream(query_states.device)
            side.wait_stream(cur)                # the inputs (and this layer's cache update) were produced there
            query_states.record_stream(side)     # a temporary of the forward pass must outlive the side stream's use
            # (key_states is a view of the cache storage, which is only reallocated after _wait_score)
            ops.score_chunk(query_states, key_states, self.sink, self.start_idx, self.end_idx, out=out,
                            workspace=self._score_ws[slot], stream=side)
            ev = self._score_ev_pool.get(layer_idx)
            if ev is None:
                ev = self._score_ev_pool[layer_idx] = torch.cuda.Event()
            ev.record(side)
            self._score_events[layer_idx] = ev
        self._score_fill[layer_idx] = f + m
        self._score[layer_idx] = self._score_buf[layer_idx][:, :, :f + m]

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
        return valid.view(score.shape), thres.item() if ratio < 1 else 0.

    def _threshold_device(self, score: torch.Tensor, ratio: float):
        """Device-side part of ``_threshold`` (no host sync): valid, thres[1], kept[1], row_counts[L*Hkv]."""
        if score.shape[-1] > 1 and score.stride(-1) == 0:
            # head-level scores expanded over the context (model/wrapper.py:56): select on the [L, Hkv] values themselves;
            # the mask comes back as the same kind of stride-0 view (nothing of size N is read or written)
            N = score.shape[-1]
            valid_h, thres, kept, rows = ops.select_heads(score[..., 0], N, ratio)
            return valid_h.unsqueeze(-1).expand(score.shape), thres, kept, rows
        if not score.is_contiguous():
            score = score.contiguous()
        return ops.select_threshold(score, ratio, row_len=score.shape[-1])

    # reference: attention/score.py:104-120
    def _threshold_uniform(self, scores: Union[torch.Tensor, List[torch.Tensor]], ratio: float):
        """-> (valids ``[L, 1, Hkv, N]`` bool, 0): exactly ``int(N*ratio)`` kept per (layer, head)."""
        score = self._stacked_score(scores if isinstance(scores, list) else list(scores))
        n_seq = score.shape[-1]
        k = int(n_seq * ratio) if ratio < 1 else n_seq
        valid, _ = ops.select_topk_rows(score, k)
        return valid, 0
