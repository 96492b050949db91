"""KV caches with eviction — host-side mirror of the reference's ``EvictCache`` / ``RetainCache``
(reference attention/kvcache.py:14-347): same constructor, hooks (``update / _get_score / prepare``), user
calls (``prune / slice / init_score / get_seq_length / _mem``) and attributes; the data path runs in the
HIP kernels of ``include/kvzip_hip.h``.

MI355X-first storage (sized for 288 GB of HBM3E):
  * before pruning every layer owns ONE ``[1, Hkv, capacity, D]`` buffer; ``update`` writes the new rows in
    place (the reference re-allocates with ``torch.cat`` on every call, kvcache.py:75-78) and
    ``key_cache[l]`` is a view of the filled part;
  * ``prune`` compacts all layers in one launch into head-major ragged buffers; with ``layout="slack"``
    (default) every head segment is followed by ``slack`` free rows so that appending the query / generated
    tokens is O(t) and ``slice`` is O(1) (the reference rebuilds the whole flattened cache per layer and
    token through ``update_flatten_view``, kvcache.py:57-73); ``layout="packed"`` reproduces the reference's
    packed ``cu_len_k`` layout and its out-of-place append bit for bit.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .score import KVScore


def _model_props(model, device, dtype):
    cfg = model.config if hasattr(model, "config") else model
    if device is None or dtype is None:
        p = next(model.parameters())
        device = p.device if device is None else device
        dtype = p.dtype if dtype is None else dtype
    return cfg, torch.device(device), dtype


class _CacheBase(KVScore):
    """State shared by EvictCache and RetainCache (reference kvcache.py:18-39 / :222-242)."""

    def __init__(self, model, evict_range: Tuple[int, int], device=None, dtype=None, reserve: int = 4096):
        KVScore.__init__(self)
        cfg, self.device, self.dtype = _model_props(model, device, dtype)
        self.n_layers = cfg.num_hidden_layers
        self.n_heads = cfg.num_attention_heads
        self.n_heads_kv = cfg.num_key_value_heads
        self.n_group_kv = self.n_heads // self.n_heads_kv

        self.start_idx, self.end_idx = evict_range
        self.ctx_len = self.end_idx - self.start_idx
        self.sink = self.start_idx  # retain initial KV pairs for system prompts
        self.prefill_ids = None
        self.ctx_ids = None

        self.get_score = False  # indicator for KV scoring
        self.pruned = False     # whether KV cache is pruned or not
        self.valid = None
        self.valid_pad = torch.ones((1, self.n_heads_kv, self.start_idx), dtype=torch.bool, device=self.device)

        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
        self._seen_tokens = 0
        self._store_k: List[torch.Tensor] = []   # [1, Hkv, capacity, D] per layer (pre-prune storage)
        self._store_v: List[torch.Tensor] = []
        self._fill: List[int] = []               # rows in use per layer
        self._reserve = int(reserve)
        self._views = {}                         # (layer, rows) -> (key view, value view, storage) of the dense cache
        self._pend_app = None                    # append left by update() for the fused update + score call of _get_score
        # scoring pass: update() leaves its append to the _get_score() call that follows it (attention/attn.py:44-54) and the two
        # become ONE library call.  Between the two calls the returned K,V views do not hold the new rows yet, so this is opt-in:
        # kvzip_amd.attn (the forward pass this package owns) and ModelKVzip.scoring switch it on
        self.fuse_update_score = False

    # -- dense (pre-prune) storage --------------------------------------------------------------
    def _dense_append(self, layer_idx: int, key_states: torch.Tensor, value_states: torch.Tensor):
        if key_states.is_cuda and key_states.device.index != self.device.index and self.device.index is not None:
            raise ops.KvzError(f"layer {layer_idx} lives on {key_states.device}, the cache on {self.device}: a cache object (its streams, "
                               "events and workspaces) works on ONE device - load the model on one GPU (one context per GPU is the "
                               "multi-GPU scheme, kvzip_amd/dist.py)")
        self._flush_append()
        t = key_states.shape[-2]
        if len(self._store_k) <= layer_idx:
            _, Hkv, _, D = key_states.shape
            cap = t + self._reserve
            self._store_k.append(torch.empty((1, Hkv, cap, D), dtype=key_states.dtype, device=key_states.device))
            self._store_v.append(torch.empty((1, Hkv, cap, D), dtype=value_states.dtype, device=value_states.device))
            self._fill.append(0)
            self.key_cache.append(None)
            self.value_cache.append(None)
        f = self._fill[layer_idx]
        sk, sv = self._store_k[layer_idx], self._store_v[layer_idx]
        cap = sk.shape[2]
        if f + t > cap:  # amortised growth
            self._wait_score(finalize=False)  # (scoring calls in flight read the storage that is being replaced)
            new_cap = max(2 * cap, f + t + self._reserve)
            for store in (self._store_k, self._store_v):
                old = store[layer_idx]
                new = torch.empty((1, old.shape[1], new_cap, old.shape[3]), dtype=old.dtype, device=old.device)
                new[:, :, :f].copy_(old[:, :, :f])
                store[layer_idx] = new
            sk, sv = self._store_k[layer_idx], self._store_v[layer_idx]
            self._views.clear()
        fast = key_states.is_cuda and key_states.stride(-1) == 1 and value_states.stride(-1) == 1
        if fast and self.get_score and not self.pruned and self.fuse_update_score and self.score_deferred:
            # scoring pass: _get_score of this layer follows (attention/attn.py:44-54) and issues the append together with the
            # scoring kernels in ONE library call (kvz_update_score_async_log); anything else that touches the cache flushes first
            self._pend_app = (layer_idx, key_states, value_states, f)
        elif fast:
            # the previous (asynchronous) scoring call of this layer read the rows that are about to be overwritten
            if self._pending:
                self._wait_score(layer_idx)
            # one launch for K and V (strided sources accepted), scalars only: rows f .. f+t of every head
            lib = ops._lib.load()
            rc = lib.kvz_dense_append(sk.data_ptr(), sv.data_ptr(), sk.stride(1), f, key_states.data_ptr(), value_states.data_ptr(),
                                      key_states.stride(1), key_states.stride(2), value_states.stride(1), value_states.stride(2),
                                      sk.shape[1], t, sk.shape[3], sk.element_size(), ops._stream(sk))
            ops.check(rc, "kvz_dense_append")
        else:
            if self._pending:
                self._wait_score(layer_idx)
            sk[:, :, f:f + t].copy_(key_states)
            sv[:, :, f:f + t].copy_(value_states)
        self._fill[layer_idx] = f + t
        # views of the filled part: the same lengths come back chunk after chunk (update ... slice), so they are cached
        vw = self._views.get((layer_idx, f + t))
        if vw is None or vw[2] is not sk:
            vw = self._views[(layer_idx, f + t)] = (sk[:, :, :f + t], sv[:, :, :f + t], sk)
            if len(self._views) > 16 * max(1, self.n_layers):
                self._views.clear()
        self.key_cache[layer_idx], self.value_cache[layer_idx] = vw[0], vw[1]

    def _flush_append(self):
        """Issue an append that update() left for _get_score (nothing else may see the cache without its rows)."""
        p = self._pend_app
        if p is None:
            return
        self._pend_app = None
        layer_idx, key_states, value_states, f = p
        if self._pending:
            self._wait_score(layer_idx)
        sk, sv = self._store_k[layer_idx], self._store_v[layer_idx]
        lib = ops._lib.load()
        rc = lib.kvz_dense_append(sk.data_ptr(), sv.data_ptr(), sk.stride(1), f, key_states.data_ptr(), value_states.data_ptr(),
                                  key_states.stride(1), key_states.stride(2), value_states.stride(1), value_states.stride(2),
                                  sk.shape[1], key_states.shape[-2], sk.shape[3], sk.element_size(), ops._stream(sk))
        ops.check(rc, "kvz_dense_append")

    def _dense_meta(self, cap: int, device):
        """(segment starts h*cap, zeros) int32 [Hkv] for the dense append launch, cached per capacity."""
        m = getattr(self, "_dense_meta_cache", None)
        if m is None or m[0] != cap:
            seg = torch.arange(self.n_heads_kv, dtype=torch.int32, device=device) * cap
            m = (cap, seg, torch.zeros(self.n_heads_kv, dtype=torch.int32, device=device))
            self._dense_meta_cache = m
        return m[1], m[2]

    def adopt_dense(self, store_k: List[torch.Tensor], store_v: List[torch.Tensor], filled: int):
        """Wrap already prefilled per-layer ``[1, Hkv, capacity, D]`` buffers without copying (e.g. the KV a
        serving engine prefilled elsewhere); ``filled`` rows are in use."""
        self._flush_append()
        self._wait_score(finalize=False)  # scoring calls still in flight read the storage that is being replaced
        self._views.clear()
        self._store_k, self._store_v = list(store_k), list(store_v)
        self._fill = [filled for _ in store_k]
        self.key_cache = [k[:, :, :filled] for k in self._store_k]
        self.value_cache = [v[:, :, :filled] for v in self._store_v]
        self._seen_tokens = filled

    def _dense_slice(self, seen_token_prev: int):
        self._flush_append()
        for l in range(len(self._store_k)):
            self._fill[l] = seen_token_prev
            self.key_cache[l] = self._store_k[l][:, :, :seen_token_prev]
            self.value_cache[l] = self._store_v[l][:, :, :seen_token_prev]

    def __del__(self):
        # asynchronous scoring reads the cache storage, the score buffer and the workspaces from side streams: order the
        # current stream behind it before the caching allocator may hand that memory to somebody else
        try:
            self._pend_app = None
            self._wait_score(finalize=False)
        except Exception:
            pass
        self._release_async()

    # reference: kvcache.py:108-112
    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        if len(self.key_cache) <= layer_idx:
            return 0
        return self._seen_tokens

    # reference: kvcache.py:140-150
    def _get_valid(self, layer_idx: int, n_seq: int) -> torch.Tensor:
        """full mask ``[1, Hkv, n_seq]`` = ones(sink) ++ valid[layer] ++ ones(rest)  (debug / interop helper; the
        kernels derive it on the fly)."""
        valid = torch.cat([self.valid_pad, self.valid[layer_idx]], dim=-1)
        size = list(valid.shape)
        size[-1] = n_seq - valid.shape[-1]
        ones = torch.ones(size, device=valid.device, dtype=torch.bool)
        return torch.cat([valid, ones], dim=-1)

    def _select(self, ratio: float, level: str):
        """-> (valid [L,1,Hkv,N] bool, thres float, r_real float); one host synchronisation."""
        self._flush_append()
        if "uniform" in level:
            self.valid, thres = self._threshold_uniform(self.score, ratio)
            n = self.valid.numel()
            k = int(self.valid.shape[-1] * ratio) if ratio < 1 else self.valid.shape[-1]
            kept = k * (n // self.valid.shape[-1])
        else:
            # scores still in the log buffer of the deferred scoring path: the launch that turns them into 16-bit values also
            # produces the first histogram of the selection (it streams every score anyway)
            hist = None
            buf = self._score_buf
            if (self._score is None and self._log_dirty and self._score_log is not None and buf is not None and buf.numel() > 0
                    and ratio < 1 and all(f == buf.shape[-1] for f in self._score_fill)):
                self._wait_score(finalize=False)
                hist = ops.select_workspace(buf.device)
                if not self._finalize_log(hist):
                    hist = None
            score = self._stacked_score(self.score)
            valid, thres_dev, kept_dev, _ = self._threshold_device(score, ratio, prehist=hist if score is buf else None)
            self.valid = valid.view(score.shape)
            host = torch.stack([thres_dev.double().squeeze(0), kept_dev.double().squeeze(0)]).cpu()  # one D2H sync
            if hist is not None and ratio < 1 and host[0] != host[0]:
                # a NaN threshold out of the pre-built histogram: either the scores hold NaNs (legitimate: the plain path says the
                # same) or the histogram was not of these scores (the kernels then flag the lost rank with NaN, kvz_select.hip)
                valid, thres_dev, kept_dev, _ = self._threshold_device(score, ratio)
                plain = torch.stack([thres_dev.double().squeeze(0), kept_dev.double().squeeze(0)]).cpu()
                if plain[0] == plain[0]:
                    raise ops.KvzError("selection: the histogram built by the finalize launch does not belong to these scores")
                self.valid, host = valid.view(score.shape), plain
            thres = float(host[0]) if ratio < 1 else 0.
            kept = int(host[1])
            n = self.valid.numel()
        assert self.valid.size(-1) == self.ctx_len
        # reference kvcache.py:133-134: rmv.float().mean() (fp32) -> 1 - python float
        r_ = 1 - float(np.float32(n - kept) / np.float32(n))
        self._evicted = n - kept
        return thres, r_


class EvictCache(_CacheBase):
    """KV cache that evicts KV from the cache before decoding (reference attention/kvcache.py:14-213)."""

    def __init__(self, model, evict_range: Tuple[int, int], device=None, dtype=None, layout: str = "slack",
                 slack: int = 1024, reserve: int = 4096, verbose: bool = True):
        super().__init__(model, evict_range, device=device, dtype=dtype, reserve=reserve)
        assert layout in ("slack", "packed")
        self.layout = layout
        self.slack = int(slack) if layout == "slack" else 0
        self.verbose = verbose
        self.info: Dict[str, Any] = {"flatten": False, "offset": None}
        self._attn_ws: Optional[torch.Tensor] = None
        self._dyn_off: Optional[torch.Tensor] = None   # device part of the appended-token count (graph-captured generation steps)
        self._dyn_val = 0                              # ... and what it holds, as the host knows it
        self._layout_epoch = 0                         # bumped whenever the flat cache moves (captured graphs hold its addresses)

    # reference: kvcache.py:41-80
    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int, cache_kwargs=dict()):
        """Update KV cache and return (K, V) of the layer."""
        if layer_idx == 0:
            seen_token = (cache_kwargs or {}).get("seen_token", key_states.size(-2))
            self._seen_tokens += seen_token

        if not self.info["flatten"]:
            self._dense_append(layer_idx, key_states, value_states)
        elif self.layout == "slack":
            t = key_states.size(-2)
            off = self.info["offset"][layer_idx]
            if off + t > self.slack:
                self._grow_slack(off + t)
            ops.append_inplace(self.key_cache[layer_idx], self.value_cache[layer_idx], key_states, value_states,
                               self.info["seg_start"][layer_idx], self.info["len_k"][layer_idx], off)
        else:  # packed: the reference's out-of-place rebuild (csrc/csrc/cuda_api.cu)
            cu_klen = self.info["cu_len_k"][layer_idx]
            head_lens = self.info["len_k"][layer_idx] + self.info["offset"][layer_idx]
            dim = key_states.size(-1)
            self.key_cache[layer_idx] = ops.update_flatten_view(
                self.key_cache[layer_idx], key_states.contiguous().view(-1, dim), head_lens, cu_klen)
            self.value_cache[layer_idx] = ops.update_flatten_view(
                self.value_cache[layer_idx], value_states.contiguous().view(-1, dim), head_lens, cu_klen)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    # reference: kvcache.py:82-106
    def slice(self, seen_token_prev: int):
        """Evict KV of queries and generated tokens from the cache (for the reuse of the context cache)."""
        if not self.info["flatten"]:
            self._dense_slice(seen_token_prev)
        elif self.layout == "slack":
            pass  # appended rows live in the slack: resetting the offsets below is all it takes
        else:
            for l in range(self.n_layers):
                cu_klen = self.info["cu_len_k"][l].tolist()
                head_lens = self.info["len_k_host"][l]
                self.key_cache[l] = torch.cat(
                    [self.key_cache[l][cu_klen[h]:cu_klen[h] + head_lens[h]] for h in range(self.n_heads_kv)])
                self.value_cache[l] = torch.cat(
                    [self.value_cache[l][cu_klen[h]:cu_klen[h] + head_lens[h]] for h in range(self.n_heads_kv)])
                self.info["cu_len_k"][l] -= self.info["offset"][l] * self.info["cu_head"]
        self.info["offset"] = [0 for _ in range(self.n_layers)]
        self._seen_tokens = seen_token_prev

    # reference: kvcache.py:114-121
    def _mem(self) -> float:
        """Memory usage of the cache in GB (rows in use, K + V)."""
        mem = 0
        for l in range(len(self.key_cache)):
            if self.info["flatten"]:
                rows = self.info["rows_used"][l] + self.n_heads_kv * self.info["offset"][l]
                mem += rows * self.key_cache[l].shape[-1] * self.key_cache[l].element_size()
            else:
                mem += self.key_cache[l].numel() * self.key_cache[l].element_size()
        return round(2 * mem / 10**9, 1)

    # reference: kvcache.py:123-138
    def prune(self, ratio: float, level: str = "pair"):
        """Prune the KV cache.  -> (thres, real kept ratio)."""
        thres, r_ = self._select(ratio, level)
        self.prepare_init()
        self.pruned = True
        if self.verbose:
            print(f"ratio {r_:.2f} ({level}), {self._mem()} GB (evict {self._evicted:.0f} pairs)")
        return thres, r_

    # reference: kvcache.py:152-185
    def prepare_init(self):
        """Evict KV and prepare the varlen metadata: one plan + one gather launch for all layers."""
        self._flush_append()
        L, Hkv = self.n_layers, self.n_heads_kv
        klen = self.key_cache[0].shape[2]
        plan = ops.compact_plan(self.valid, self.sink, klen, slack=self.slack)
        meta = plan.meta.cpu()  # the single host sync of prepare_init
        o = 0
        len_h = meta[o:o + L * Hkv].view(L, Hkv); o += L * Hkv
        o += L * Hkv
        o += L * (Hkv + 1)
        max_h = meta[o:o + L]
        rows_used = len_h.sum(-1).tolist()
        totals = [r + Hkv * self.slack for r in rows_used]
        ks, vs = ops.compact_layers(self.key_cache, self.value_cache, plan, totals)
        self.key_cache, self.value_cache = list(ks), list(vs)
        self._store_k, self._store_v, self._fill = [], [], []  # release the dense storage
        self._views.clear()
        self._layout_epoch = getattr(self, "_layout_epoch", 0) + 1
        self._plan = plan
        cu_head = torch.arange(Hkv + 1, dtype=torch.int32, device=self.device)
        self.info = {
            "flatten": True,
            "layout": self.layout,
            "cu_head": cu_head,
            "len_k": [plan.len_k[l] for l in range(L)],          # kv lengths of heads in a layer (device int32)
            "len_k_host": len_h.tolist(),
            "max_len_k": [int(x) for x in max_h.tolist()],       # python ints: no per-step device sync
            "cu_len_k": [plan.cu_len_k[l].clone() for l in range(L)],  # packed-layout cumulative lengths
            "seg_start": [plan.seg_start[l] for l in range(L)],  # first row of each head segment
            "rows_used": rows_used,
            "offset": [0 for _ in range(L)],                     # rows appended after pruning, per layer
        }
        self._refresh_meta_host()

    def _refresh_meta_host(self):
        """ctypes copies [seg_start ++ len_k] per layer: passed by value to the attention kernel (slack layout only: in the packed
        layout the segment starts move with every appended token and live on the device)."""
        if self.layout != "slack":
            self.info["meta_host"] = None
            return
        seg = torch.stack(self.info["seg_start"]).cpu().tolist() if self.info["seg_start"] else []
        self.info["meta_host"] = [ops._meta_host(seg[l], self.info["len_k_host"][l], self.n_heads_kv) for l in range(self.n_layers)]

    def _grow_slack(self, need: int):
        """Re-lay out every layer with a larger slack (rare: only when more than ``slack`` tokens are appended)."""
        new_slack = max(2 * self.slack, need + 256)
        Hkv = self.n_heads_kv
        for l in range(self.n_layers):
            lens = self.info["len_k_host"][l]
            off = self.info["offset"][l]
            D = self.key_cache[l].shape[-1]
            total = sum(lens) + Hkv * new_slack
            nk = torch.empty((total, D), dtype=self.key_cache[l].dtype, device=self.device)
            nv = torch.empty((total, D), dtype=self.value_cache[l].dtype, device=self.device)
            old_start = self.info["seg_start"][l].tolist()
            starts, acc = [], 0
            for h in range(Hkv):
                starts.append(acc)
                n = lens[h] + off
                nk[acc:acc + n].copy_(self.key_cache[l][old_start[h]:old_start[h] + n])
                nv[acc:acc + n].copy_(self.value_cache[l][old_start[h]:old_start[h] + n])
                acc += lens[h] + new_slack
            self.key_cache[l], self.value_cache[l] = nk, nv
            self.info["seg_start"][l] = torch.tensor(starts, dtype=torch.int32, device=self.device)
        self.slack = new_slack
        self._layout_epoch += 1
        self._refresh_meta_host()

    # reference: kvcache.py:187-213
    def prepare(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                layer_idx: int):
        """Flatten the queries for variable-length attention and bump the per-layer offsets."""
        bsz, n_heads_q, q_len, dim = query_states.shape
        cu_len_q = q_len * self.info["cu_head"]
        if q_len == 1:
            query_states = query_states.reshape(-1, self.n_group_kv, dim)  # pure view: [Hkv*1, G, D]
        else:
            query_states = query_states.view(bsz, self.n_heads_kv, self.n_group_kv, q_len, dim)
            query_states = query_states.transpose(2, 3).contiguous().view(-1, self.n_group_kv, dim)

        self.info["offset"][layer_idx] += q_len
        info = {
            "cu_len_q": cu_len_q,
            "max_len_q": q_len,
            "max_len_k": self.info["max_len_k"][layer_idx] + self.info["offset"][layer_idx],
            # MI355X layout: segment starts + base lengths + host-side offset
            "k_start": self.info["seg_start"][layer_idx],
            "k_len": self.info["len_k"][layer_idx],
            "k_len_offset": self.info["offset"][layer_idx],
            "meta_host": self._meta_host(layer_idx),
        }
        if self.layout == "packed":
            self.info["cu_len_k"][layer_idx] += cu_len_q
            info["cu_len_k"] = self.info["cu_len_k"][layer_idx]
            info["k_start"] = info["cu_len_k"][:-1]
        else:
            info["cu_len_k"] = None
        return query_states, key_states.view(-1, 1, dim), value_states.view(-1, 1, dim), info

    def attend(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor, info: dict,
               causal: bool = True, softmax_scale: Optional[float] = None) -> torch.Tensor:
        """Variable-length attention over the pruned cache: what the reference gets from
        ``flash_attn_varlen_func`` (attention/attn.py:61-71).  Returns ``[Hkv*q_len, G, D]``."""
        dim = query_states.shape[-1]
        ws = self._attn_workspace(info["max_len_q"], dim, query_states.device)
        return ops.varlen_attn(query_states, key_states.view(-1, dim), value_states.view(-1, dim), info["k_start"],
                               info["k_len"], info["max_len_q"], info["max_len_k"], causal=causal,
                               softmax_scale=softmax_scale, workspace=ws,
                               k_len_offset=info["k_len_offset"], meta_host=info.get("meta_host"))

    def _attn_workspace(self, q_len: int, dim: int, device) -> torch.Tensor:
        """Zero-initialised scratch of the attention kernel, one per query length (its size does not depend on the key lengths)."""
        cache = getattr(self, "_attn_ws_cache", None)
        if cache is None:
            cache = self._attn_ws_cache = {}
        ws = cache.get((q_len, dim))
        if ws is None:
            if len(cache) > 4:
                cache.clear()
            ws = cache[(q_len, dim)] = ops.attn_workspace(self.n_heads_kv, self.n_group_kv, q_len, dim, device)
        return ws

    def _meta_host(self, layer_idx: int):
        """Head segments of a layer as kernel arguments (host copy made at prune / slack growth), or None."""
        mh = self.info.get("meta_host")
        return mh[layer_idx] if mh is not None else None

    def update_attend(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int,
                      softmax_scale: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Decode step (one new token) on the pruned cache, slack layout: ``update`` + ``prepare`` + ``attend`` in one
        launch of the attention kernel, which also writes the token's K, V row into the slack of every head.
        query ``[1, H, 1, D]``, key / value ``[1, Hkv, 1, D]``; returns ``[Hkv, G, D]`` like ``attend``.  The cache
        bookkeeping (``_seen_tokens``, ``info["offset"]``) ends up exactly as after the three separate calls."""
        assert self.info["flatten"] and self.layout == "slack" and query_states.shape[-2] == 1
        if layer_idx == 0:
            self._seen_tokens += 1
        off = self.info["offset"][layer_idx]
        if off + 1 > self.slack:
            self._grow_slack(off + 1)
        dim = query_states.shape[-1]
        max_len_k = self.info["max_len_k"][layer_idx] + off + 1
        ws = self._attn_workspace(1, dim, query_states.device)
        # (the appended-token count = host part + device part: see decode_graph; without a graph there is no device part)
        if self._dyn_off is not None and off < self._dyn_val:  # (after slice(): the device part restarts with the host's count)
            self._sync_dyn(off)
        out = ops.varlen_attn_append(query_states.reshape(-1, self.n_group_kv, dim), self.key_cache[layer_idx],
                                     self.value_cache[layer_idx], key_states, value_states,
                                     self.info["seg_start"][layer_idx], self.info["len_k"][layer_idx], off - self._dyn_val, max_len_k,
                                     softmax_scale=softmax_scale, workspace=ws, meta_host=self._meta_host(layer_idx),
                                     offset_dev=self._dyn_off, out=out)
        self.info["offset"][layer_idx] = off + 1
        return out

    # ---- a whole generation step as ONE HIP graph ---------------------------------------------------------------------------
    def decode_graph(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, softmax_scale: Optional[float] = None):
        """Capture the attention part of ONE generation step over all layers - L x (append the token's K,V + variable-length
        attention) - into a HIP graph and return a ``DecodeGraph``; ``replay()`` runs the step for the tensors that are in the
        static buffers at that moment (``query [L, 1, H, 1, D]``, ``key`` / ``value [L, 1, Hkv, 1, D]``; outputs in ``.out``
        ``[L, Hkv, G, D]``).  Replays need unchanged kernel arguments, so the part of the appended-token count that changes from
        token to token lives in a device counter that the last node of the graph advances (``k_len_offset_dev``); the host
        bookkeeping (``info["offset"]``, ``_seen_tokens``) advances in ``replay()``.  What a serving engine does with its
        static input buffers; the per-layer hooks of a Python forward pass (``update_attend``) stay available beside it."""
        assert self.info["flatten"] and self.layout == "slack"
        L = self.n_layers
        assert query.shape[0] == L and key.shape[0] == L and value.shape[0] == L and query.shape[-2] == 1
        lib = ops._lib.load()
        offsets0, seen0 = list(self.info["offset"]), self._seen_tokens
        assert min(offsets0) == max(offsets0), "a generation step starts with the same number of appended tokens in every layer"
        if offsets0[0] + 2 > self.slack:
            self._grow_slack(offsets0[0] + 2)
        if self._dyn_off is None:
            self._dyn_off = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._sync_dyn(offsets0[0])   # the WHOLE count lives on the device from here on: the captured host part is 0
        out = torch.empty((L, self.n_heads_kv, self.n_group_kv, query.shape[-1]), dtype=query.dtype, device=query.device)
        ws = self._attn_workspace(1, query.shape[-1], query.device)
        dyn0 = self._dyn_val
        torch.cuda.synchronize(query.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for l in range(L):
                self.update_attend(query[l], key[l], value[l], l, softmax_scale=softmax_scale, out=out[l])
            ops.check(lib.kvz_add_i32(self._dyn_off.data_ptr(), 1, ops._stream(self._dyn_off)), "kvz_add_i32")
        # (capturing executes nothing: the host bookkeeping goes back to where it was)
        self.info["offset"], self._seen_tokens, self._dyn_val = offsets0, seen0, dyn0
        # the captured kernels hold RAW ADDRESSES: the graph object keeps every tensor behind them alive (the shared attention
        # workspace may be dropped from the cache's own table by later calls with other query lengths, the static input buffers
        # belong to the caller)
        pinned = (ws, self._dyn_off, query, key, value, list(self.key_cache), list(self.value_cache),
                  list(self.info["seg_start"]), list(self.info["len_k"]))
        return DecodeGraph(self, graph, out, pinned)

    def _sync_dyn(self, value: int):
        """device part of the appended-token count := value (one fill on the current stream)"""
        if self._dyn_val != value or value == 0:
            self._dyn_off.fill_(value)
            self._dyn_val = value


class DecodeGraph:
    """One captured generation step of an ``EvictCache`` (see ``EvictCache.decode_graph``)."""

    def __init__(self, kv: "EvictCache", graph, out: torch.Tensor, pinned=()):
        self.kv, self.graph, self.out = kv, graph, out
        self.epoch = kv._layout_epoch
        self._pinned = pinned   # every tensor whose address a captured kernel holds (see decode_graph)

    def replay(self) -> torch.Tensor:
        kv = self.kv
        off = kv.info["offset"]
        if self.epoch != kv._layout_epoch:
            raise ops.KvzError("decode graph: the flat cache was re-laid out since the capture (prune / slack growth): capture again")
        if max(off) + 1 > kv.slack:
            raise ops.KvzError("decode graph: the slack of the head segments is used up (the graph holds the addresses of the flat "
                               "cache: grow the slack and capture again)")
        assert min(off) == max(off)
        if off[0] != kv._dyn_val:  # (tokens appended by the per-layer hooks, or slice(), since the last replay)
            kv._sync_dyn(off[0])
        self.graph.replay()
        kv.info["offset"] = [o + 1 for o in off]
        kv._seen_tokens += 1
        kv._dyn_val += 1
        return self.out


class RetainCache(_CacheBase):
    """KV cache that keeps the full KV in memory and subsamples it at every attention call, so that several
    compression ratios can be evaluated from a single prefill (reference attention/kvcache.py:216-347).
    Produces the same flattened tensors as :class:`EvictCache` (the reference's internal cross-check)."""

    def __init__(self, model, evict_range: Tuple[int, int], device=None, dtype=None, reserve: int = 4096,
                 verbose: bool = True):
        super().__init__(model, evict_range, device=device, dtype=dtype, reserve=reserve)
        self.verbose = verbose
        self._cu_head = torch.arange(self.n_heads_kv + 1, dtype=torch.int32, device=self.device)

    # reference: kvcache.py:244-267
    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int, cache_kwargs=dict()):
        if layer_idx == 0:
            seen_token = (cache_kwargs or {}).get("seen_token", key_states.shape[-2])
            self._seen_tokens += seen_token
        self._dense_append(layer_idx, key_states, value_states)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    # reference: kvcache.py:269-276
    def slice(self, seen_token_prev: int):
        assert len(self.key_cache[0].shape) == 4, "Cache at each layer should be 4D tensor"
        self._dense_slice(seen_token_prev)
        self._seen_tokens = seen_token_prev

    # reference: kvcache.py:278-282
    def _mem(self) -> float:
        mem = self.n_layers * self.key_cache[0].numel() * self.key_cache[0].element_size()
        return round(2 * mem / 10**9, 1)

    # reference: kvcache.py:284-298
    def prune(self, ratio: float, level: str = "pair"):
        """Prune the KV cache (fake): only the mask is computed; it is applied before every attention."""
        thres, r_ = self._select(ratio, level)
        self.pruned = True
        if self.verbose:
            print(f"ratio {r_:.2f} ({level}), threshold {thres:.4f} (evict {self._evicted:.0f} pairs)")
        return thres, r_

    # reference: kvcache.py:312-347
    def prepare(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                layer_idx: int):
        bsz, n_heads_q, q_len, dim = query_states.shape
        klen = key_states.size(2)
        query_states = query_states.view(bsz, self.n_heads_kv, self.n_group_kv, q_len, dim)
        query_states = query_states.transpose(2, 3).contiguous().view(-1, self.n_group_kv, dim)
        cu_seqlens_q = q_len * self._cu_head

        plan = ops.compact_plan(self.valid[layer_idx:layer_idx + 1], self.sink, klen, slack=0)
        total = int(plan.cu_len_k[0, -1])  # host sync, as the reference's boolean indexing
        if plan.heads:  # head-level mask (one byte per head): the batched entry point takes it
            ks, vs = ops.compact_layers([key_states], [value_states], plan, [total])
            k_flat, v_flat = ks[0], vs[0]
        else:
            k_flat, v_flat = ops.compact_layer(key_states, value_states, plan, 0, total)
        info = {
            "cu_len_q": cu_seqlens_q,
            "cu_len_k": plan.cu_len_k[0],
            "max_len_q": q_len,
            "max_len_k": int(plan.max_len_k[0]),
            "k_start": plan.seg_start[0],
            "k_len": plan.len_k[0],
            "k_len_offset": 0,
        }
        return query_states, k_flat.view(-1, 1, dim), v_flat.view(-1, 1, dim), info

    attend = EvictCache.attend
    _attn_workspace = EvictCache._attn_workspace
    _attn_ws = None
