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
    ``n_score_streams`` side streams (default: automatic, three with the pruned fp16 call, two otherwise).  The tail of one persistent kernel, the launch gaps and the two tiny
    merge / finalize kernels of a call then overlap with the big kernels of the next ones (+9 % scoring throughput on
    MI355X with two streams; round 5, same boxes: three streams are 3-4.5 % SLOWER than two, four 7 % - rounds 2-3 had measured
    three ahead when a call still had a separate merge launch).  Ordering is kept with events: the side stream waits for the caller's stream (inputs), ``update`` of a layer
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


def _spin_time(streams, cycles: int = 400_000) -> float:
    """Wall time (ms) of one spin kernel per stream, launched together."""
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
    return timed(streams)


def _overlaps(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True if spin kernels on the two streams run concurrently (wall time well below twice one kernel)."""
    _spin_time([a])  # warm-up (first launch on a stream creates its queue)
    _spin_time([b])
    one = min(_spin_time([a]), _spin_time([b]))
    both = _spin_time([a, b])
    return both < 1.5 * one


def _all_overlap(streams) -> bool:
    """True if spin kernels on ALL the streams run at the same time (pairwise overlap does not prove it)."""
    if len(streams) < 2:
        return True
    one = min(_spin_time([st]) for st in streams)
    return min(_spin_time(streams), _spin_time(streams)) < 1.5 * one


def _side_streams(device, n: int):
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _SIDE_STREAMS.get(key)
    if pool is not None and len(pool) >= n:
        return pool[:n]
    with torch.cuda.device(key):
        # candidates of NORMAL priority (measured: two high-priority streams among the three cost 1-4 % in bench.py - kernels on
        # the remaining normal-priority stream fall behind and the caller's stream then waits for that layer's previous call)
        cands = [torch.cuda.Stream(device=dev) for _ in range(6)]
        chosen = []
        try:
            for c in cands:
                if all(_overlaps(c, o) for o in chosen):
                    chosen.append(c)
                if len(chosen) >= n:
                    break
            while len(chosen) > 2 and not _all_overlap(chosen):  # three-way concurrency has to be seen, not inferred
                chosen.pop()
        except Exception:  # no spin kernel in this build: take the candidates as they come
            chosen = cands[:n]
        import os
        if os.environ.get("KVZ_STREAM_DEBUG"):
            import sys
            print("kvzip_amd: side streams = candidates", [cands.index(c) for c in chosen], "of", len(cands),
                  "| one:", [round(_spin_time([c]), 3) for c in chosen], "all together:", round(_spin_time(chosen), 3), file=sys.stderr)
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
        self._score = None             # externally assigned scores (list / tensor); None = views of the score buffer
        self.sink = None
        self.start_idx, self.end_idx = None, None
        self.ctx_len = None
        self._score_buf: Optional[torch.Tensor] = None
        self._score_fill: List[int] = []
        self._score_ws: List[Optional[torch.Tensor]] = []
        self._ws_need = {}             # (q_len, m, H) -> workspace bytes
        self.n_score_streams = 0       # 0 = automatic: three side streams where the pruned scoring call runs (fp16: its small kernels fill the gaps
                                       # of the other streams' big ones, +4 %), two for the two-pass call (profiles/r5_streams_ab.txt); 1 = the caller's stream
        self._score_exclusive = False  # True: the next calls run alone on the caller's stream (clean kernel timings)
        self._score_side: List["torch.cuda.Stream"] = []
        self._async = -1               # handle of the library's asynchronous-scoring context (events per layer)
        self._pending = False          # scoring calls may still be in flight on a side stream
        self.score_deferred = True     # row slices merged by atomics into a log buffer, ONE finalize launch when the scores are read
        self._score_log: Optional[torch.Tensor] = None   # [L, 1, Hkv, N] int32: bit patterns of the fp32 log-scores (-inf = empty)
        self._log_dirty = False
        self._dev_idx: Optional[int] = None   # index of the ONE device this object works on (resolved at the first scoring call)
        self._q_hold = {}              # side-stream slot -> query tensors of the calls whose tail phases are still pending (pipelined tail)
        self._win_stats: List[Optional[torch.Tensor]] = []  # per layer: row statistics written by the scoring forward (f2)
        self.fuse_forward_score = False       # set by the forward pass this package owns (kvzip_amd.attn via ModelKVzip.scoring)

    # ---- asynchronous scoring: bookkeeping -----------------------------------------------------------------
    def _auto_streams(self, lib, dtype) -> int:
        """side streams of the scoring calls: ``n_score_streams`` if set, else three where the pruned call (knob ``score_prune``) runs and
        two for the two-pass call (the knob off)."""
        if self.n_score_streams:
            return max(1, int(self.n_score_streams))
        return 3 if lib.kvz_debug_get_tunable(b"score_prune") >= 3 else 2

    @property
    def score(self):
        """Per-layer scores (reference attribute: list of L ``[1, Hkv, n]`` tensors, or whatever was assigned).  Reading it
        orders the caller's stream behind outstanding scoring."""
        self._wait_score()
        if self._score is not None:
            return self._score
        if self._score_buf is None:
            return None
        return [self._score_buf[l][:, :, :self._score_fill[l]] for l in range(self.n_layers)]

    @score.setter
    def score(self, value):
        self._score = value

    def _wait_score(self, layer_idx: Optional[int] = None, finalize: bool = True):
        """Make the current stream wait for the scoring calls still in flight (of one layer, or of all).  Waiting for all of them
        with ``finalize`` also turns the log buffer of the deferred path into the 16-bit scores (one launch for everything)."""
        if self._pending and self._async >= 0:
            lib = ops._lib.load()
            dev = torch.device(self.device)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if layer_idx is None:
                self._flush_tails(lib)
            ops.check(lib.kvz_async_wait(self._async, -1 if layer_idx is None else layer_idx, ops.raw_stream(idx)), "kvz_async_wait")
            if layer_idx is None:
                self._pending = False
        if layer_idx is None and finalize and self._log_dirty:
            self._finalize_log()

    def _flush_tails(self, lib):
        """Pipelined tail of the scoring calls (knob ``score_prune`` = 6): the library leaves the bounds / candidate-key phases of a call to
        the next two calls on the same workspace and side stream; before anybody reads the scores (or frees what those phases read) the
        pending phases of every workspace are launched and the done-event of one layer of that stream is recorded behind them
        (``kvz_score_tail_flush_async``), so the ``kvz_async_wait`` that follows covers them.  A wait for ONE layer (the append of its next
        chunk) does not need this: it only orders behind the row-statistics kernel that read the repeat rows."""
        for slot, ws in enumerate(self._score_ws):
            if ws is None or slot >= len(self._score_side) or slot >= self.n_layers:
                continue
            rc = lib.kvz_score_tail_flush_async(self._async, slot, ws.data_ptr(), self._score_side[slot].cuda_stream)
            if rc < 0:
                ops.check(rc, "kvz_score_tail_flush_async")
        self._q_hold = {}   # (freed behind the flush launches: record_stream covers them)

    def _finalize_log(self, hist: Optional[torch.Tensor] = None) -> bool:
        """Log buffer -> 16-bit scores (one launch for all layers and chunks).  ``hist``: a selection workspace that receives the
        first histogram of the global-threshold selection in the same launch (the launch streams every score anyway).  Returns
        True when the histogram was produced."""
        log, buf = self._score_log, self._score_buf
        self._log_dirty = False
        if log is None or buf is None or buf.numel() == 0:
            return False
        lib = ops._lib.load()
        if hist is not None:
            rc = lib.kvz_score_finalize_log_hist(log.data_ptr(), log.numel(), buf.data_ptr(), ops._dtype_code(buf.dtype),
                                                 hist.data_ptr(), hist.numel(), ops._stream(buf))
            ops.check(rc, "kvz_score_finalize_log_hist")
            return True
        rc = lib.kvz_score_finalize_log(log.data_ptr(), log.numel(), buf.data_ptr(), ops._dtype_code(buf.dtype), ops._stream(buf))
        ops.check(rc, "kvz_score_finalize_log")
        return False

    def _new_score_log(self, n: int) -> Optional[torch.Tensor]:
        if not self.score_deferred or n == 0 or not torch.device(self.device).type == "cuda":
            return None
        log = torch.empty((self.n_layers, 1, self.n_heads_kv, n), dtype=torch.int32, device=self.device)
        ops.check(ops._lib.load().kvz_score_log_fill(log.data_ptr(), log.numel(), ops._stream(log)), "kvz_score_log_fill")
        return log

    def close(self):
        """Release the library-side objects of this cache (the asynchronous-scoring context: 2 events per layer) now instead of at
        garbage collection.  Waits for outstanding scoring calls first; the object stays usable (a new context is created on demand).
        A context handle is meant to be driven from ONE host thread (the library does not lock around its per-slot state)."""
        try:
            self._wait_score(finalize=False)
        finally:
            self._release_async()

    def _release_async(self):
        try:   # (nothing may stay pending on a workspace that is about to be freed: its address may come back)
            lib = ops._lib.load()
            for ws in self._score_ws:
                if ws is not None:
                    lib.kvz_score_tail_flush(ws.data_ptr())
        except Exception:
            pass
        if self._async >= 0:
            try:
                ops._lib.load().kvz_async_destroy(self._async)
            except Exception:
                pass
            self._async = -1

    # reference: attention/score.py:25-31
    def init_score(self):
        self.get_score = True
        self.causal_mask_score = None
        n = int(self.ctx_len) if self.ctx_len is not None else 0
        self._wait_score(finalize=False)
        self._log_dirty = False
        self._score_buf = torch.empty((self.n_layers, 1, self.n_heads_kv, n), dtype=self.dtype, device=self.device)
        self._score_log = self._new_score_log(n)
        self._score_fill = [0 for _ in range(self.n_layers)]
        self._score = None

    # reference: attention/score.py:33-34
    def _update_score(self, layer_idx: int, score: torch.Tensor):
        """Append an externally computed ``[1, Hkv, m]`` block (kept for API compatibility)."""
        m = score.shape[-1]
        f = self._score_fill[layer_idx]
        self._ensure_score_capacity(f + m)
        self._wait_score(layer_idx)
        self._score_buf[layer_idx][:, :, f:f + m].copy_(score)
        self._score_fill[layer_idx] = f + m

    def _ensure_score_capacity(self, need: int):
        if self._score_buf.shape[-1] >= need:
            return
        self._wait_score()  # (finalizes what the log buffer holds: the old scores move as 16-bit values)
        new = torch.empty((self.n_layers, 1, self.n_heads_kv, need), dtype=self.dtype, device=self.device)
        old = self._score_buf.shape[-1]
        if old:
            new[..., :old].copy_(self._score_buf)
        self._score_buf = new
        self._score_log = self._new_score_log(need)

    def _check_device(self, dev, layer_idx: int):
        """A cache object (side streams, events, workspaces) works on ONE device, and the library launches on the CURRENT HIP
        device: both scoring entries (``_get_score``, ``_score_forward``) refuse anything else instead of launching on the wrong GPU
        (``ModelKVzip`` makes the model's device current around its forward passes)."""
        di = self._dev_idx
        if di is None:  # (resolved once: the cache object lives on ONE device)
            d0 = torch.device(self.device)
            di = self._dev_idx = d0.index if d0.index is not None else dev.index
        if dev.index != di:
            raise ops.KvzError(f"layer {layer_idx} lives on {dev}, the cache on {self.device}: a cache object (side streams, events, "
                               "workspaces) works on ONE device - load the model on one GPU (one context per GPU is the multi-GPU scheme)")
        if di != torch.cuda.current_device():
            raise ops.KvzError(f"the cache lives on {dev} but cuda:{torch.cuda.current_device()} is current: wrap the forward pass in "
                               f"`with torch.cuda.device({dev.index}):` (the library launches on the current HIP device)")

    # reference: attention/score.py:36-65
    def _get_score(self, query_states: torch.Tensor, key_states: torch.Tensor, layer_idx: int):
        """query ``[1, H, q, D]``, key ``[1, Hkv, klen, D]`` (cache ++ repeat chunk).  Writes the chunk's
        ``[1, Hkv, end_idx-start_idx]`` scores straight into the layer's score buffer.  ONE call into the library: the
        side-stream ordering (events) lives behind ``kvz_score_chunk_async``."""
        lib = ops._lib.load()
        m = self.end_idx - self.start_idx
        f = self._score_fill[layer_idx]
        if self._score_buf.shape[-1] < f + m:
            self._ensure_score_capacity(f + m)
        buf = self._score_buf
        bsz, H, q_len, D = query_states.shape
        Hkv = self.n_heads_kv
        klen = key_states.shape[2]
        if not query_states.is_cuda:
            raise ops.KvzError("the HIP path needs device tensors (no CPU fallback)")
        assert bsz == 1 and query_states.stride(3) == 1 and query_states.stride(2) == D
        assert key_states.stride(3) == 1 and key_states.stride(2) == D and key_states.dtype == query_states.dtype
        dev = query_states.device
        self._check_device(dev, layer_idx)
        need = self._ws_need.get((q_len, m, H))
        if need is None:
            need = self._ws_need[(q_len, m, H)] = lib.kvz_score_workspace_bytes(Hkv, H // Hkv, q_len, m, self.sink)
        nstreams = 1 if self._score_exclusive else self._auto_streams(lib, query_states.dtype)
        slot = layer_idx % nstreams if nstreams > 1 else 0
        # pipelined tail (knob score_prune = 6): three workspace sets per side stream, the library rotates through them
        pipelined = lib.kvz_debug_get_tunable(b"score_prune") == 6
        if pipelined:
            need = 3 * ((need + 255) // 256 * 256)
        while len(self._score_ws) <= slot:
            self._score_ws.append(None)
        ws = self._score_ws[slot]
        if ws is None or ws.numel() < need:
            self._wait_score(finalize=False)  # (the old workspace of this slot may still be in use)
            ws = self._score_ws[slot] = torch.empty(need, dtype=torch.uint8, device=dev)
        if self._async < 0:
            self._async = lib.kvz_async_create(self.n_layers)
            if self._async < 0:
                ops.check(self._async, "kvz_async_create")
        cur = ops.raw_stream(dev.index)
        if nstreams == 1:
            self._wait_score(finalize=False)  # the caller's stream: everything before it is ordered anyway, later calls wait for it
            side = cur
        else:
            if len(self._score_side) < nstreams:
                self._wait_score(finalize=False)  # (the pool may be rebuilt: nothing may be pending on the streams it replaces)
                self._score_side = _side_streams(dev, nstreams)
            st = self._score_side[slot]
            side = st.cuda_stream
            query_states.record_stream(st)  # a temporary of the forward pass must outlive the side stream's use
            # (key_states is a view of the cache storage, which is only reallocated after _wait_score)
            self._pending = True
            if pipelined:
                # the candidate-key phase of this call reads the query rows again two calls later on this stream: the tensor stays
                # referenced until then (three per stream), or until the tails are flushed
                hold = self._q_hold.setdefault(slot, [])
                hold.append(query_states)
                if len(hold) > 3:
                    del hold[0]
        n_tot = buf.shape[-1]
        log = self._score_log
        pend = getattr(self, "_pend_app", None)
        if pend is not None and (pend[0] != layer_idx or log is None or log.shape[-1] != n_tot
                                 or key_states.data_ptr() != self._store_k[layer_idx].data_ptr()):
            self._flush_append()  # (not the call update() expected: issue its append on its own)
            pend = None
        if pend is not None:
            # update() of this layer left its append for this call: ONE library call appends the repeat chunk's K,V on the caller's
            # stream and issues the scoring kernels on the side stream (kvz_update_score_async_log)
            self._pend_app = None
            _, ks, vs, fill = pend
            sk, sv = self._store_k[layer_idx], self._store_v[layer_idx]
            out_ptr = log.data_ptr() + (layer_idx * Hkv * n_tot + f) * 4
            rc = lib.kvz_update_score_async_log(self._async, layer_idx, cur, side, sk.data_ptr(), sv.data_ptr(), sk.stride(1), fill,
                                                ks.data_ptr(), vs.data_ptr(), ks.stride(1), ks.stride(2), vs.stride(1), vs.stride(2),
                                                ks.shape[-2], query_states.data_ptr(), query_states.stride(1), self.sink,
                                                self.start_idx, self.end_idx, q_len, Hkv, H // Hkv, D,
                                                ops._dtype_code(query_states.dtype), out_ptr, n_tot, ws.data_ptr(), ws.numel())
            self._log_dirty = True
        elif log is not None and log.shape[-1] == n_tot:
            # deferred path: pass B merges its row slices by atomics into the log buffer, the finalize launch happens once, when
            # the scores are read (2 launches per call)
            out_ptr = log.data_ptr() + (layer_idx * Hkv * n_tot + f) * 4
            rc = lib.kvz_score_chunk_async_log(self._async, layer_idx, cur, side, query_states.data_ptr(), query_states.stride(1),
                                               key_states.data_ptr(), key_states.stride(1), klen, self.sink, self.start_idx,
                                               self.end_idx, q_len, Hkv, H // Hkv, D, ops._dtype_code(query_states.dtype), out_ptr,
                                               n_tot, ws.data_ptr(), ws.numel())
            self._log_dirty = True
        else:
            out_ptr = buf.data_ptr() + (layer_idx * Hkv * n_tot + f) * buf.element_size()
            rc = lib.kvz_score_chunk_async(self._async, layer_idx, cur, side, query_states.data_ptr(), query_states.stride(1),
                                           key_states.data_ptr(), key_states.stride(1), klen, self.sink, self.start_idx,
                                           self.end_idx, q_len, Hkv, H // Hkv, D, ops._dtype_code(query_states.dtype), out_ptr, n_tot,
                                           ws.data_ptr(), ws.numel())
        ops.check(rc, "kvz_score_chunk_async")
        self._score_fill[layer_idx] = f + m

    # f2: the scoring forward's attention kernel emits the row statistics itself
    def _score_forward(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int,
                       softmax_scale: Optional[float] = None) -> Optional[torch.Tensor]:
        """Dense causal attention of a scoring forward pass AND the scores of the current chunk from ONE QK^T over the window
        keys: ``kvz_flash_fwd_window`` (caller's stream) applies the scoring rounding chain to the accumulators of the key tiles of
        ``sink ++ [start_idx, end_idx) ++ repeat chunk`` and leaves the per-row softmax statistics; the column-maximum pass then
        runs on a side stream (``kvz_score_from_stats_async_log``).  The reference computes that QK^T twice (attention/attn.py:53-54
        and :75-89); the two-pass scoring call of ``_get_score`` computes the window's part a second time as well.
        Returns the attention output ``[1, q, H, D]``, or None when the shape is not one the 32-row kernel takes (head_dim 128,
        enough rows) or the deferred log buffer is not in use - the caller then goes through ``_get_score`` + the plain forward."""
        lib = ops._lib.load()
        m = self.end_idx - self.start_idx
        f = self._score_fill[layer_idx]
        if self._score_buf.shape[-1] < f + m:
            self._ensure_score_capacity(f + m)
        buf, log = self._score_buf, self._score_log
        n_tot = buf.shape[-1]
        bsz, H, q_len, D = query_states.shape
        Hkv = self.n_heads_kv
        if (log is None or log.shape[-1] != n_tot or not query_states.is_cuda or D != 128 or bsz != 1
                or query_states.stride(3) != 1 or query_states.stride(2) != D or key_states.stride(3) != 1 or key_states.stride(2) != D):
            return None
        if getattr(self, "_pend_app", None) is not None:
            self._flush_append()  # the forward reads the rows of the repeat chunk
        dev = query_states.device
        self._check_device(dev, layer_idx)
        R = (H // Hkv) * q_len
        stride = (R + 127) // 128 * 128
        while len(self._win_stats) <= layer_idx:
            self._win_stats.append(None)
        st = self._win_stats[layer_idx]
        if st is None or st.shape[1] < stride:
            # (one buffer per layer: its previous reader is this layer's previous scoring call, which update() has waited for)
            st = self._win_stats[layer_idx] = torch.empty((Hkv, stride, 2), dtype=torch.float32, device=dev)
        out = ops.flash_fwd_window(query_states, key_states, value_states, self.sink, self.start_idx, self.end_idx, st,
                                   softmax_scale=softmax_scale)
        if out is None:
            return None
        if self._async < 0:
            self._async = lib.kvz_async_create(self.n_layers)
            if self._async < 0:
                ops.check(self._async, "kvz_async_create")
        nstreams = 1 if self._score_exclusive else (max(1, int(self.n_score_streams)) if self.n_score_streams else 2)
        slot = layer_idx % nstreams if nstreams > 1 else 0
        cur = ops.raw_stream(dev.index)
        if nstreams == 1:
            self._wait_score(finalize=False)
            side = cur
        else:
            if len(self._score_side) < nstreams:
                self._wait_score(finalize=False)  # (see _get_score)
                self._score_side = _side_streams(dev, nstreams)
            sst = self._score_side[slot]
            side = sst.cuda_stream
            query_states.record_stream(sst)
            self._pending = True
        out_ptr = log.data_ptr() + (layer_idx * Hkv * n_tot + f) * 4
        rc = lib.kvz_score_from_stats_async_log(self._async, layer_idx, cur, side, query_states.data_ptr(), query_states.stride(1),
                                                key_states.data_ptr(), key_states.stride(1), key_states.shape[2], self.sink,
                                                self.start_idx, self.end_idx, q_len, Hkv, H // Hkv, D,
                                                ops._dtype_code(query_states.dtype), st.data_ptr(), st.stride(0) // 2, out_ptr, n_tot)
        ops.check(rc, "kvz_score_from_stats_async_log")
        self._log_dirty = True
        self._score_fill[layer_idx] = f + m
        return out

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

    def _threshold_device(self, score: torch.Tensor, ratio: float, prehist: Optional[torch.Tensor] = None):
        """Device-side part of ``_threshold`` (no host sync): valid, thres[1], kept[1], row_counts[L*Hkv]."""
        if score.shape[-1] > 1 and score.stride(-1) == 0:
            # head-level scores expanded over the context (model/wrapper.py:56): select on the [L, Hkv] values themselves;
            # the mask comes back as the same kind of stride-0 view (nothing of size N is read or written)
            N = score.shape[-1]
            valid_h, thres, kept, rows = ops.select_heads(score[..., 0], N, ratio)
            return valid_h.unsqueeze(-1).expand(score.shape), thres, kept, rows
        if not score.is_contiguous():
            score = score.contiguous()
        return ops.select_threshold(score, ratio, row_len=score.shape[-1], prehist=prehist if score.is_contiguous() else None)

    # reference: attention/score.py:104-120
    def _threshold_uniform(self, scores: Union[torch.Tensor, List[torch.Tensor]], ratio: float):
        """-> (valids ``[L, 1, Hkv, N]`` bool, 0): exactly ``int(N*ratio)`` kept per (layer, head)."""
        score = self._stacked_score(scores if isinstance(scores, list) else list(scores))
        n_seq = score.shape[-1]
        k = int(n_seq * ratio) if ratio < 1 else n_seq
        valid, _ = ops.select_topk_rows(score, k)
        return valid, 0
