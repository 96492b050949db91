"""The pipelined tail of the pruned scoring call (round 6, knob ``score_prune`` = 6): behind the row-statistics kernel of call i of a side
stream ONE launch runs the merge of call i, the bounds of call i-1 and the candidate-key pass of call i-2 of that stream, each call on its
own third of the workspace (``kvz_score.hip``: ``score_tail_kernel``, ``TailState``).  The same kernels' bodies on the same data in the same
order per call as the chained call (knob 3), so everything the cache object hands out must be THE SAME BITS: scores, threshold, mask -
whatever the number of layers (fewer layers than pipeline stages included), chunk shapes that change inside the pass, a flush in the
middle, NaN inputs."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _set_knob(v):
    from kvzip_amd import _lib
    return _lib.load().kvz_debug_set_tunable(b"score_prune", v)


def _scoring_pass(L, H, Hkv, D, sink, N, chunk, dtype, knob, nstreams=3, fused=True, poke=None, seed=5, read_mid=False):
    """A whole scoring pass through the cache object (update + _get_score per layer and chunk, slice, select) with the knob set."""
    from kvzip_amd.kvcache import EvictCache
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(seed)
    chunks = []
    for c, st in enumerate(range(0, N, chunk)):
        m = min(chunk, N - st)
        chunks.append((sink + st, sink + st + m, m + (13 if c == 0 else 26)))
    q_max = max(c[2] for c in chunks)
    cap = sink + N + q_max + 8
    store = []
    for l in range(L):
        t = torch.empty((1, Hkv, cap, D), dtype=dtype, device=DEV)
        t[:, :, :sink + N] = torch.randn(1, Hkv, sink + N, D, generator=g, device=DEV, dtype=torch.float32).to(dtype)
        store.append(t)
    Q = [torch.randn(1, H, q_max, D, generator=g, device=DEV, dtype=torch.float32).to(dtype) for _ in range(L)]
    Kr = [torch.randn(1, Hkv, q_max, D, generator=g, device=DEV, dtype=torch.float32).to(dtype) for _ in range(L)]
    if poke is not None:
        poke(Q, store)
    _set_knob(knob)
    try:
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dtype, verbose=False)
        kv.n_score_streams = nstreams
        kv.fuse_update_score = fused
        kv.adopt_dense(store, store, sink + N)
        kv.init_score()
        mid = None
        for c, (st, en, q_len) in enumerate(chunks):
            kv.start_idx, kv.end_idx = st, en
            seen = kv._seen_tokens
            for l in range(L):
                # (fresh query tensors per call, as a forward pass makes them: the pipeline has to keep them alive)
                q = Q[l][:, :, :q_len].clone()
                kr = Kr[l][:, :, :q_len]
                k_all, _ = kv.update(kr, kr, l)
                kv._get_score(q, k_all, l)
                del q
            kv.slice(seen)
            if read_mid and c == 0:
                mid = torch.stack([s.clone() for s in kv.score])   # (reading the scores flushes the pipeline in the middle of the pass)
        kv.start_idx, kv.get_score = sink, False
        kv.valid = None
        thres, r_real = kv._select(0.3, "pair")
        score = torch.stack([s for s in kv.score]).clone()
        valid = kv.valid.clone()
        torch.cuda.synchronize()
        kv.close()
    finally:
        _set_knob(-1)
    return score, thres, valid, mid


def _bits(t):
    return t.view(torch.int16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("L,nstreams", [(7, 3), (2, 3), (1, 3), (5, 2), (9, 1)])
def test_pipelined_tail_returns_the_bits_of_the_chained_call(L, nstreams, dtype):
    """9 / 7 / 5 layers: every side stream sees several calls per chunk; 2 / 1 layers: fewer calls than pipeline stages, so consecutive
    chunks of ONE layer follow each other on a stream; the last chunk is shorter (another call shape in the same pipeline)."""
    H, Hkv, D, sink, N, chunk = 8, 2, 128, 16, 2500, 1000
    s3, t3, v3, _ = _scoring_pass(L, H, Hkv, D, sink, N, chunk, dtype, 3, nstreams)
    s6, t6, v6, _ = _scoring_pass(L, H, Hkv, D, sink, N, chunk, dtype, 6, nstreams)
    assert s3.numel() == L * Hkv * N
    assert torch.equal(_bits(s3), _bits(s6)) and t3 == t6 and torch.equal(v3, v6)
    # and the unfused form (update and _get_score as two library calls)
    s6u, t6u, v6u, _ = _scoring_pass(L, H, Hkv, D, sink, N, chunk, dtype, 6, nstreams, fused=False)
    assert torch.equal(_bits(s3), _bits(s6u)) and t3 == t6u and torch.equal(v3, v6u)


def test_pipelined_tail_flushed_in_the_middle_of_a_pass():
    """reading ``kv.score`` after the first chunk launches the pending phases; the scores read there and at the end are those of the chained call"""
    L, H, Hkv, D, sink, N, chunk = 6, 8, 2, 128, 16, 2100, 700
    s3, t3, v3, m3 = _scoring_pass(L, H, Hkv, D, sink, N, chunk, torch.float16, 3, read_mid=True)
    s6, t6, v6, m6 = _scoring_pass(L, H, Hkv, D, sink, N, chunk, torch.float16, 6, read_mid=True)
    assert torch.equal(_bits(m3), _bits(m6))
    assert torch.equal(_bits(s3), _bits(s6)) and t3 == t6 and torch.equal(v3, v6)


def test_pipelined_tail_head_dim_64_and_nan():
    """D = 64 (another instance of the fused launch) and a NaN query row (its KV head is poisoned by the bounds phase, one call later)"""
    L, H, Hkv, sink, N, chunk = 4, 4, 2, 8, 1500, 600

    def poke(Q, store):
        Q[1][0, 3, 40, 5] = float("nan")   # layer 1, query head 3 -> KV head 1

    s3, t3, v3, _ = _scoring_pass(L, H, Hkv, 64, sink, N, chunk, torch.float16, 3, poke=poke)
    s6, t6, v6, _ = _scoring_pass(L, H, Hkv, 64, sink, N, chunk, torch.float16, 6, poke=poke)
    assert torch.isnan(s3[1, 0, 1].float()).all() and not torch.isnan(s3[1, 0, 0].float()).any()
    nn = lambda t: torch.nan_to_num(t.float(), nan=7.0)
    assert torch.equal(nn(s3), nn(s6)) and torch.equal(v3, v6)
    assert (t3 == t6) or (t3 != t3 and t6 != t6)


def test_tail_flush_entry_points():
    """kvz_score_tail_flush on a workspace nobody used: 0; the flush of a cache object leaves nothing pending (second flush: 0)"""
    from kvzip_amd import _lib
    lib = _lib.load()
    ws = torch.empty(1024, dtype=torch.uint8, device=DEV)
    assert lib.kvz_score_tail_flush(ws.data_ptr()) == 0
    assert lib.kvz_score_tail_flush_async(12345, 0, ws.data_ptr(), None) < 0   # (bad handle)
