"""CPU ORACLE — test infrastructure, NOT part of the product path.

A plain restatement, on CPU tensors, of the reference algorithm for the KV-eviction hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
``kvzip_amd`` never does (the product path raises when the HIP library is missing).

Parity status: PINNED for a1-a12/a14 — every function below is checked bit-for-bit against golden vectors
produced by importing the reference's own Python (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
``varlen_attn`` (a13) restates flash-attn 2.7.4.post1's published semantics (third party, absent from
/root/reference): PARITY UNPINNED at that boundary; it is anchored on the reference's call site, on the identity
"compacted varlen attention == dense attention with evicted keys masked" and on an independent implementation of the
published semantics (torch's math-backend SDPA with a bottom-right aligned causal mask, tests/test_oracle_golden.py).

Each function cites the reference lines it follows (paths relative to snu-mllab/KVzip).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple, Union

import torch


# ------------------------------------------------------------------------------------------------
# a1/a2  KVScore._get_score + _make_mask/_mask_causal      reference: attention/score.py:36-85
# ------------------------------------------------------------------------------------------------
def causal_mask(window: int, dtype: torch.dtype) -> torch.Tensor:
    """fp32 tensor: 0 where key j <= query i, finfo(dtype).min elsewhere (score.py:67-74: torch.full without
    dtype gives float32)."""
    i = torch.arange(window).view(window, 1)
    j = torch.arange(window).view(1, window)
    return torch.where(j <= i, torch.zeros((), dtype=torch.float32),
                       torch.full((), torch.finfo(dtype).min, dtype=torch.float32))


def get_score(query_states: torch.Tensor, key_states: torch.Tensor, sink: int, start_idx: int,
              end_idx: int) -> torch.Tensor:
    """query ``[1,H,q,D]``, key ``[1,Hkv,klen,D]`` -> ``[1,Hkv,end-start]`` in the input dtype.

    Same arithmetic as score.py:44-63, one KV head at a time: half matmul (fp32 accumulate, one rounding),
    division by the python float sqrt(D) in fp32 opmath rounded to half, fp32 mask add on the last q x q
    block rounded to half, half softmax (fp32 internals, one rounding), max over (group, query).
    """
    bsz, H, q_len, D = query_states.shape
    Hkv = key_states.shape[1]
    assert bsz == 1
    G = H // Hkv
    m = end_idx - start_idx
    mask = causal_mask(q_len, query_states.dtype)
    out = torch.empty((1, Hkv, m), dtype=query_states.dtype)
    for h in range(Hkv):
        kh = key_states[0, h]
        keys = torch.cat([kh[:sink], kh[start_idx:end_idx], kh[kh.shape[0] - q_len:]], dim=0)  # [k, D]
        qh = query_states[0, h * G:(h + 1) * G]                                             # [G, q, D]
        a = torch.matmul(qh, keys.t().contiguous()) / math.sqrt(D)                          # [G, q, k]
        a[..., -q_len:] += mask                                                              # in-place, as :85
        p = torch.softmax(a, dim=-1)
        out[0, h] = p[..., sink:sink + m].amax(dim=(0, 1))
    return out


def get_score_chain_fp32(query_states: torch.Tensor, key_states: torch.Tensor, sink: int, start_idx: int,
                         end_idx: int) -> torch.Tensor:
    """The same function written as an explicit fp32 rounding chain (what the HIP kernel implements):
    x = half(half(q.k [fp32]) / float32(sqrt(D))); masked keys dropped; p = exp(x - max) / sum; max over rows."""
    dt = query_states.dtype
    bsz, H, q_len, D = query_states.shape
    Hkv = key_states.shape[1]
    G = H // Hkv
    m = end_idx - start_idx
    c = torch.tensor(math.sqrt(D), dtype=torch.float32)
    out = torch.empty((1, Hkv, m), dtype=dt)
    vis = torch.arange(q_len).view(1, q_len) <= torch.arange(q_len).view(q_len, 1)  # [i, j]
    for h in range(Hkv):
        kh = key_states[0, h].float()
        keys = torch.cat([kh[:sink], kh[start_idx:end_idx], kh[kh.shape[0] - q_len:]], dim=0)
        qh = query_states[0, h * G:(h + 1) * G].float()
        x = (qh @ keys.t()).to(dt).float()
        x = (x / c).to(dt).float()
        x[..., -q_len:] = torch.where(vis, x[..., -q_len:], torch.full((), -float("inf")))
        mx = x.amax(dim=-1, keepdim=True)
        e = torch.exp(x - mx)
        p = (e / e.sum(dim=-1, keepdim=True)).to(dt)
        out[0, h] = p[..., sink:sink + m].amax(dim=(0, 1))
    return out


# ------------------------------------------------------------------------------------------------
# a4  KVScore._threshold                                   reference: attention/score.py:88-102
# ------------------------------------------------------------------------------------------------
def threshold(score: Union[torch.Tensor, List[torch.Tensor]], ratio: float) -> Tuple[torch.Tensor, float]:
    if isinstance(score, list):
        score = torch.stack(score, dim=0)
    if ratio < 1:
        flat = score.reshape(-1).float()
        n = max(int(flat.numel() * ratio) - 1, 0)
        # n-th largest (0-based) == (numel-1-n)-th smallest; kthvalue is 1-based
        thres = torch.kthvalue(flat, flat.numel() - n).values.item()
        valids = score.float() > thres
    else:
        valids = torch.ones_like(score, dtype=torch.bool)
        thres = 0.
    return valids, thres


# ------------------------------------------------------------------------------------------------
# a5  KVScore._threshold_uniform                           reference: attention/score.py:104-120
# ------------------------------------------------------------------------------------------------
def threshold_heads(head_scores: torch.Tensor, n_ctx: int, ratio: float) -> Tuple[torch.Tensor, float]:
    """Head-level selection WITHOUT the expansion: the reference feeds ``head_scores[L,Hkv]`` expanded to ``[L,1,Hkv,n_ctx]``
    to ``threshold`` (model/wrapper.py:54-57 -> attention/score.py:88-102).  In the sorted expanded tensor every head value
    occupies ``n_ctx`` consecutive slots, hence ``sorted[idx] == sorted_heads[idx // n_ctx]``.
    Returns ``(kept heads bool [L,Hkv], thres)``; pinned against tests/golden/g4_head_score.npz."""
    if ratio < 1:
        flat = head_scores.reshape(-1)
        sorted_heads = torch.sort(flat, descending=True).values
        n = max(int(flat.numel() * n_ctx * ratio) - 1, 0)
        thres = sorted_heads[n // n_ctx].item()
        return head_scores > thres, thres
    return torch.ones_like(head_scores, dtype=torch.bool), 0.


def threshold_uniform(scores: Union[torch.Tensor, Sequence[torch.Tensor]], ratio: float
                      ) -> Tuple[torch.Tensor, int]:
    """Per (layer, head) row keep exactly k = int(N*ratio).  torch.topk's choice among tied values is
    unspecified; this oracle (and the HIP kernel) keep the LOWEST-index ties, which coincides with the
    reference on tie-free rows."""
    valids = []
    for score in scores:
        if ratio < 1:
            n_seq = score.shape[-1]
            k = int(n_seq * ratio)
            s = score.float()
            valid = torch.zeros_like(score, dtype=torch.bool)
            if k > 0:
                kth = torch.kthvalue(s, n_seq - k + 1, dim=-1, keepdim=True).values  # k-th largest
                greater = s > kth
                ties = s == kth
                need = k - greater.sum(-1, keepdim=True)
                tie_rank = ties.long().cumsum(-1) - 1
                valid = greater | (ties & (tie_rank < need))
        else:
            valid = torch.ones_like(score, dtype=torch.bool)
        valids.append(valid)
    return torch.stack(valids), 0


# ------------------------------------------------------------------------------------------------
# a8/a9  EvictCache._get_valid + prepare_init              reference: attention/kvcache.py:140-185
# ------------------------------------------------------------------------------------------------
def get_valid(valid_layer: torch.Tensor, sink: int, n_seq: int) -> torch.Tensor:
    """valid_layer ``[1,Hkv,N]`` -> ``[1,Hkv,n_seq]`` = ones(sink) ++ valid ++ ones(rest)."""
    Hkv = valid_layer.shape[1]
    pad = torch.ones((1, Hkv, sink), dtype=torch.bool)
    tail = torch.ones((1, Hkv, n_seq - sink - valid_layer.shape[-1]), dtype=torch.bool)
    return torch.cat([pad, valid_layer, tail], dim=-1)


def prepare_init(key_cache: Sequence[torch.Tensor], value_cache: Sequence[torch.Tensor], valid: torch.Tensor,
                 sink: int):
    """-> (flat_k list, flat_v list, len_k list[int32 Hkv], cu_len_k list[int32 Hkv+1], max_len_k list)"""
    ks, vs, lens, cus, mxs = [], [], [], [], []
    for layer, (k, v) in enumerate(zip(key_cache, value_cache)):
        _, Hkv, klen, D = k.shape
        full = get_valid(valid[layer], sink, klen)[0]  # [Hkv, klen]
        rows_k, rows_v, ln = [], [], []
        for h in range(Hkv):
            idx = torch.nonzero(full[h]).squeeze(-1)
            rows_k.append(k[0, h].index_select(0, idx))
            rows_v.append(v[0, h].index_select(0, idx))
            ln.append(idx.numel())
        ks.append(torch.cat(rows_k, 0))
        vs.append(torch.cat(rows_v, 0))
        ln = torch.tensor(ln, dtype=torch.int32)
        lens.append(ln)
        cus.append(torch.cat([torch.zeros(1, dtype=torch.int32), ln.cumsum(0).int()]))
        mxs.append(ln.max())
    return ks, vs, lens, cus, mxs


# ------------------------------------------------------------------------------------------------
# a11  update_flatten_view                                 reference: csrc/csrc/cuda_api.cu:15-111
# ------------------------------------------------------------------------------------------------
def update_flatten_view(cache: torch.Tensor, state: torch.Tensor, headlens: torch.Tensor,
                        cu_headlens: torch.Tensor) -> torch.Tensor:
    """out = cat_h( cache[cu[h] : cu[h]+headlens[h]], state[h*t:(h+1)*t] )   (cuda_api.cu:29-39 offsets)"""
    H = headlens.shape[0]
    t = state.shape[0] // H
    out = torch.empty((cache.shape[0] + H * t, cache.shape[1]), dtype=cache.dtype)
    for h in range(H):
        hl, src = int(headlens[h]), int(cu_headlens[h])
        dst = src + h * t
        ins = int(cu_headlens[h + 1]) + h * t
        out[dst:dst + hl] = cache[src:src + hl]
        out[ins:ins + t] = state[h * t:(h + 1) * t]
    return out


# ------------------------------------------------------------------------------------------------
# a12  EvictCache.prepare (query re-layout)                reference: attention/kvcache.py:187-213
# ------------------------------------------------------------------------------------------------
def prepare_query(query_states: torch.Tensor, Hkv: int) -> torch.Tensor:
    bsz, H, q_len, D = query_states.shape
    G = H // Hkv
    return query_states.view(bsz, Hkv, G, q_len, D).transpose(2, 3).contiguous().view(-1, G, D)


# ------------------------------------------------------------------------------------------------
# a14  EvictCache.slice (flatten branch)                   reference: attention/kvcache.py:82-106
# ------------------------------------------------------------------------------------------------
def slice_flat(cache: torch.Tensor, cu_len_k: torch.Tensor, len_k: torch.Tensor) -> torch.Tensor:
    return torch.cat([cache[int(cu_len_k[h]):int(cu_len_k[h]) + int(len_k[h])] for h in range(len_k.shape[0])])


# ------------------------------------------------------------------------------------------------
# a13  flash_attn_varlen_func(causal=True) as used at attention/attn.py:61-71  (PARITY UNPINNED)
# ------------------------------------------------------------------------------------------------
def varlen_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, k_start: Sequence[int], k_len: Sequence[int],
                q_len: int, causal: bool = True, scale: float = None) -> torch.Tensor:
    """q ``[Hkv*q_len, G, D]``, k/v ``[rows, D]``; fp32 softmax(q.k^T*scale), bottom-right aligned causal mask,
    output rounded once to the input dtype."""
    HQ, G, D = q.shape
    Hkv = HQ // q_len
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    out = torch.zeros_like(q)
    for h in range(Hkv):
        ln, st = int(k_len[h]), int(k_start[h])
        if ln == 0:
            continue
        kh = k.view(-1, D)[st:st + ln].float()
        vh = v.view(-1, D)[st:st + ln].float()
        qh = q[h * q_len:(h + 1) * q_len].float()            # [q_len, G, D]
        s = torch.einsum("igd,jd->igj", qh, kh) * scale       # [q_len, G, ln]
        if causal:
            i = torch.arange(q_len).view(q_len, 1, 1)
            j = torch.arange(ln).view(1, 1, ln)
            s = torch.where(j <= i + (ln - q_len), s, torch.full((), -float("inf")))
        mx = s.amax(-1, keepdim=True)
        mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
        e = torch.exp(s - mx)
        den = e.sum(-1, keepdim=True)
        p = torch.where(den > 0, e / den, torch.zeros_like(e))
        out[h * q_len:(h + 1) * q_len] = torch.einsum("igj,jd->igd", p, vh).to(q.dtype)
    return out


def dense_masked_attn(q: torch.Tensor, k_full: torch.Tensor, v_full: torch.Tensor, full_valid: torch.Tensor,
                      q_len: int) -> torch.Tensor:
    """Cross-oracle for the compaction identity: dense attention over the FULL K/V ``[Hkv, klen, D]`` with evicted
    keys masked to -inf; causal alignment counts only kept keys (bottom-right over the compacted sequence)."""
    HQ, G, D = q.shape
    Hkv = HQ // q_len
    out = torch.zeros_like(q)
    for h in range(Hkv):
        keep = full_valid[h]
        rank = keep.long().cumsum(0) - 1                       # position in the compacted sequence
        ln = int(keep.sum())
        s = torch.einsum("igd,jd->igj", q[h * q_len:(h + 1) * q_len].float(), k_full[h].float()) / math.sqrt(D)
        i = torch.arange(q_len).view(q_len, 1, 1)
        vis = keep.view(1, 1, -1) & (rank.view(1, 1, -1) <= i + (ln - q_len))
        s = torch.where(vis, s, torch.full((), -float("inf")))
        p = torch.softmax(s, dim=-1)
        out[h * q_len:(h + 1) * q_len] = torch.einsum("igj,jd->igd", p, v_full[h].float()).to(q.dtype)
    return out
