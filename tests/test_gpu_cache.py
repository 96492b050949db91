"""GPU: the drop-in cache objects (kvzip_amd.EvictCache / RetainCache) replay the reference's EvictCache life
cycle recorded in tests/golden/g6_cache_*.npz: scoring -> prune -> query prefill (t=7) -> 2 decode steps -> slice.
Integer / byte results are bit-exact; scores and attention outputs carry their stated tolerance."""
import types

import numpy as np
import pytest
import torch

import kvzip_oracle as orc
from conftest import from_bits, load_golden, to_bits, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(L, H, Hkv):
    return types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)


def _run_scoring(kv, g, L, sink, bf):
    chunks = [(sink, sink + 50), (sink + 50, sink + 80)]
    kv.init_score()
    for ci, (st, en) in enumerate(chunks):
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            q = from_bits(g[f"sc/{ci}/{l}/q"], bf).to(DEV)
            kr = from_bits(g[f"sc/{ci}/{l}/k"], bf).to(DEV)
            vr = from_bits(g[f"sc/{ci}/{l}/v"], bf).to(DEV)
            kfull, _ = kv.update(kr, vr, l)
            assert kfull.shape[2] == seen + kr.shape[2]
            kv._get_score(q, kfull, l)
        kv.slice(seen)
        assert kv._seen_tokens == seen and kv.key_cache[0].shape[2] == seen
    kv.start_idx, kv.get_score = sink, False


@pytest.mark.parametrize("tag", ["f16_pair", "bf16_pair", "f16_uniform"])
@pytest.mark.parametrize("layout", ["packed", "slack"])
def test_evict_cache_life_cycle(tag, layout):
    from kvzip_amd.kvcache import EvictCache
    g = load_golden(f"g6_cache_{tag}.npz")
    L, H, Hkv, D, sink, N, bf = g["meta"].tolist()
    G = H // Hkv
    dt = torch.bfloat16 if bf else torch.float16
    level = "pair-uniform" if "uniform" in tag else "pair"
    kv = EvictCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, layout=layout, slack=16, reserve=8,
                    verbose=False)
    assert kv.get_seq_length() == 0
    for l in range(L):
        K, V = kv.update(from_bits(g[f"K0/{l}"], bf).to(DEV), from_bits(g[f"V0/{l}"], bf).to(DEV), l)
        assert np.array_equal(to_bits(K), g[f"K0/{l}"])
    assert kv.get_seq_length() == sink + N and kv._seen_tokens == sink + N
    _run_scoring(kv, g, L, sink, bf)
    # scores: rounding-chain tolerance (see tests/test_gpu_kernels.py::test_score_chunk_golden)
    for l in range(L):
        assert kv.score[l].shape == (1, Hkv, N)
        d = ulp_diff(kv.score[l], from_bits(g[f"score/{l}"], bf))
        assert (d <= 1).float().mean() >= 0.99 and d.max() <= 8
    # integer-exact part: selection + compaction from the REFERENCE's scores
    kv.score = [from_bits(g[f"score/{l}"], bf).to(DEV) for l in range(L)]
    ratio = float(g["ratio"][0])
    thres, r_real = kv.prune(ratio, level)
    assert thres == g["thres"][0] and r_real == g["r_real"][0]
    assert np.array_equal(kv.valid.cpu().numpy(), g["valid"])
    assert kv.pruned and kv.info["flatten"]
    assert kv._mem() == g["mem_gb"][0]
    for l in range(L):
        assert np.array_equal(kv.info["len_k"][l].cpu().numpy(), g[f"len_k/{l}"])
        assert kv.info["max_len_k"][l] == int(g[f"max_len_k/{l}"][0])
        cu = g[f"cu_len_k/{l}"]
        seg = kv.info["seg_start"][l].cpu().numpy()
        for h in range(Hkv):
            n = int(cu[h + 1] - cu[h])
            assert np.array_equal(to_bits(kv.key_cache[l][seg[h]:seg[h] + n]), g[f"flatK/{l}"][cu[h]:cu[h + 1]])
            assert np.array_equal(to_bits(kv.value_cache[l][seg[h]:seg[h] + n]), g[f"flatV/{l}"][cu[h]:cu[h + 1]])
        if layout == "packed":
            assert np.array_equal(kv.info["cu_len_k"][l].cpu().numpy(), cu)
            assert np.array_equal(to_bits(kv.key_cache[l]), g[f"flatK/{l}"])
    # generation: query prefill of 7 tokens, then two decode steps (reference model/wrapper.py:251-284)
    seen = kv._seen_tokens
    for si, t in enumerate((7, 1, 1)):
        for l in range(L):
            q = from_bits(g[f"gen/{si}/{l}/q"], bf).to(DEV)
            kn = from_bits(g[f"gen/{si}/{l}/k"], bf).to(DEV)
            vn = from_bits(g[f"gen/{si}/{l}/v"], bf).to(DEV)
            ke, ve = kv.update(kn, vn, l)
            qe, ke2, ve2, info = kv.prepare(q, ke, ve, l)
            assert np.array_equal(to_bits(qe), g[f"gen/{si}/{l}/q_out"])
            assert np.array_equal(info["cu_len_q"].cpu().numpy(), g[f"gen/{si}/{l}/cu_len_q"])
            assert [info["max_len_q"], info["max_len_k"]] == g[f"gen/{si}/{l}/max_len"].tolist()
            cu = g[f"gen/{si}/{l}/cu_len_k"]
            if layout == "packed":
                assert np.array_equal(info["cu_len_k"].cpu().numpy(), cu)
                assert np.array_equal(to_bits(ke2.view(-1, D)), g[f"gen/{si}/{l}/k_out"])
                assert np.array_equal(to_bits(ve2.view(-1, D)), g[f"gen/{si}/{l}/v_out"])
            else:
                seg = info["k_start"].cpu().numpy()
                ln = info["k_len"].cpu().numpy() + info["k_len_offset"]
                for h in range(Hkv):
                    assert ln[h] == cu[h + 1] - cu[h]
                    assert np.array_equal(to_bits(ke2.view(-1, D)[seg[h]:seg[h] + ln[h]]),
                                          g[f"gen/{si}/{l}/k_out"][cu[h]:cu[h + 1]])
            att = kv.attend(qe, ke2, ve2, info)
            want = from_bits(g[f"gen/{si}/{l}/attn"], bf).float()
            tol = 8e-3 if bf else 1e-3
            assert (att.cpu().float() - want).abs().max() <= tol
    assert [kv._seen_tokens, kv.get_seq_length()] == g["seen_after_gen"].tolist()
    kv.slice(seen)
    assert kv._seen_tokens == g["seen_after_slice"][0]
    assert kv.info["offset"] == [0] * L
    for l in range(L):
        if layout == "packed":
            assert np.array_equal(to_bits(kv.key_cache[l]), g[f"sliced/K/{l}"])
            assert np.array_equal(kv.info["cu_len_k"][l].cpu().numpy(), g[f"sliced/cu_len_k/{l}"])


@pytest.mark.parametrize("tag", ["f16_pair", "f16_uniform"])
def test_retain_cache_equals_evict_cache(tag):
    """The reference's internal cross-check: RetainCache.prepare == EvictCache compacted path."""
    from kvzip_amd.kvcache import EvictCache, RetainCache
    g = load_golden(f"g6_cache_{tag}.npz")
    L, H, Hkv, D, sink, N, bf = g["meta"].tolist()
    dt = torch.bfloat16 if bf else torch.float16
    level = "pair-uniform" if "uniform" in tag else "pair"
    caches = [EvictCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, layout="packed", verbose=False),
              RetainCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)]
    for kv in caches:
        for l in range(L):
            kv.update(from_bits(g[f"K0/{l}"], bf).to(DEV), from_bits(g[f"V0/{l}"], bf).to(DEV), l)
        kv.score = [from_bits(g[f"score/{l}"], bf).to(DEV) for l in range(L)]
    r0 = caches[0].prune(0.4, level)
    r1 = caches[1].prune(0.4, level)
    assert r0 == r1 and torch.equal(caches[0].valid, caches[1].valid)
    for si, t in enumerate((7, 1)):
        for l in range(L):
            q = from_bits(g[f"gen/{si}/{l}/q"], bf).to(DEV)
            kn = from_bits(g[f"gen/{si}/{l}/k"], bf).to(DEV)
            vn = from_bits(g[f"gen/{si}/{l}/v"], bf).to(DEV)
            outs = []
            for kv in caches:
                ke, ve = kv.update(kn, vn, l)
                outs.append(kv.prepare(q, ke, ve, l))
            (qe, ke, ve, ie), (qr, kr, vr, ir) = outs
            assert torch.equal(qe, qr) and torch.equal(ke, kr) and torch.equal(ve, vr)
            assert torch.equal(ie["cu_len_k"], ir["cu_len_k"]) and ie["max_len_k"] == ir["max_len_k"]
            assert np.array_equal(to_bits(ke.view(-1, D)), g[f"gen/{si}/{l}/k_out"])
            a0 = caches[0].attend(qe, ke, ve, ie)
            a1 = caches[1].attend(qr, kr, vr, ir)
            assert torch.equal(a0, a1)


def test_head_level_prune_config5_shape():
    """--level head path (model/wrapper.py:40-58): scores given as an expanded [L,1,Hkv,N] tensor."""
    from kvzip_amd.kvcache import EvictCache
    g = load_golden("g4_head_score.npz")
    hs = from_bits(g["qwen2.5-14b/head_score"], True)           # [48, 8] bf16
    L, Hkv = hs.shape
    N, sink, D = 1000, 4, 64
    kv = EvictCache(_cfg(L, Hkv * 5, Hkv), (sink, sink + N), device=DEV, dtype=torch.bfloat16, verbose=False)
    for l in range(L):
        kv.update(torch.randn(1, Hkv, sink + N, D, dtype=torch.bfloat16, device=DEV),
                  torch.randn(1, Hkv, sink + N, D, dtype=torch.bfloat16, device=DEV), l)
    kv.score = hs.to(DEV).unsqueeze(-1).expand(-1, -1, N).unsqueeze(1)  # as load_head_score returns it
    thres, r_real = kv.prune(0.6, "head")
    assert thres == 0.53515625
    kept = g["qwen2.5-14b/kept/1000/0.6"]
    assert kept.sum() == 229
    for l in range(L):
        assert np.array_equal(kv.info["len_k"][l].cpu().numpy(), sink + kept[l].astype(np.int32) * N)


def test_slack_growth_and_update_after_many_tokens():
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N = 1, 4, 2, 64, 2, 64
    kv = EvictCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=torch.float16, slack=4, verbose=False)
    K = torch.randn(1, Hkv, sink + N, D, device=DEV).half()
    kv.update(K, K.clone(), 0)
    kv.score = [torch.rand(1, Hkv, N, device=DEV).half()]
    kv.prune(0.5)
    lens = kv.info["len_k_host"][0]
    news = []
    for step in range(9):  # exceeds the slack of 4 -> re-layout
        kn = torch.randn(1, Hkv, 1, D, device=DEV).half()
        news.append(kn)
        ke, ve = kv.update(kn, kn, 0)
        kv.prepare(torch.randn(1, H, 1, D, device=DEV).half(), ke, ve, 0)
    seg = kv.info["seg_start"][0].tolist()
    for h in range(Hkv):
        tail = kv.key_cache[0][seg[h] + lens[h]: seg[h] + lens[h] + 9]
        assert torch.equal(tail, torch.cat([n[0, h] for n in news]))


def test_async_scoring_equals_single_stream():
    """Scoring calls of consecutive layers overlap on side streams (kvzip_amd/score.py).  The result must be bit-identical to
    the single-stream run through the whole update -> _get_score -> slice cycle: the next chunk's ``update`` overwrites the rows
    the previous chunk's scoring of that layer read, so a missing event shows up as corrupted scores."""
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N, chunk = 6, 8, 2, 128, 16, 1536, 256
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(77)
    K = [torch.randn(1, Hkv, sink + N, D, generator=g, device=DEV).half() for _ in range(L)]
    V = [torch.randn(1, Hkv, sink + N, D, generator=g, device=DEV).half() for _ in range(L)]
    chunks = [(sink + c, min(sink + c + chunk, sink + N)) for c in range(0, N, chunk)]
    q_in = [[torch.randn(1, H, (en - st) + 9, D, generator=g, device=DEV).half() for _ in range(L)] for st, en in chunks]
    k_in = [[torch.randn(1, Hkv, (en - st) + 9, D, generator=g, device=DEV).half() for _ in range(L)] for st, en in chunks]

    def run(nstreams, deferred=True):
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=torch.float16, verbose=False)
        kv.n_score_streams = nstreams
        kv.score_deferred = deferred
        for l in range(L):
            kv.update(K[l], V[l], l)
        kv.init_score()
        for c, (st, en) in enumerate(chunks):
            kv.start_idx, kv.end_idx = st, en
            seen = kv._seen_tokens
            for l in range(L):
                k_all, _ = kv.update(k_in[c][l], k_in[c][l], l)
                kv._get_score(q_in[c][l], k_all, l)
            kv.slice(seen)
        kv.start_idx, kv.get_score = sink, False
        score = torch.stack([s.clone() for s in kv.score])   # reading .score waits for the side streams
        thres, r_real = kv.prune(0.4)
        return score, thres, kv.valid.clone()

    s1, t1, v1 = run(1)
    for n in (2, 3, 3):
        s2, t2, v2 = run(n)
        assert torch.equal(s1.view(torch.int16), s2.view(torch.int16))
        assert t1 == t2 and torch.equal(v1, v2)
    # deferred finalize (row slices of pass B merged by atomics into a log buffer, ONE finalize launch when .score is read)
    # against the direct path (finalize launch per call): the same bits
    for n in (1, 3):
        s3, t3, v3 = run(n, deferred=False)
        assert torch.equal(s1.view(torch.int16), s3.view(torch.int16))
        assert t1 == t3 and torch.equal(v1, v3)


def test_deferred_scores_and_external_blocks():
    """The deferred path next to the reference's ``_update_score`` (externally computed blocks written straight into the score
    buffer), a buffer that has to grow, and ``.score`` read between chunks: every block must come out as the direct kernel call
    (``ops.score_chunk``) produces it."""
    from kvzip_amd import ops
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N, chunk = 2, 8, 2, 128, 8, 1024, 256
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(5)
    K = [torch.randn(1, Hkv, sink + N, D, generator=g, device=DEV).half() for _ in range(L)]
    kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=torch.float16, verbose=False)
    for l in range(L):
        kv.update(K[l], K[l], l)
    kv.ctx_len = N // 2           # too small on purpose: the buffers have to grow once
    kv.init_score()
    want = [[] for _ in range(L)]
    for c, st in enumerate(range(sink, sink + N, chunk)):
        en = st + chunk
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            q = torch.randn(1, H, chunk + 7, D, generator=g, device=DEV).half()
            kr = torch.randn(1, Hkv, chunk + 7, D, generator=g, device=DEV).half()
            k_all, _ = kv.update(kr, kr, l)
            direct = ops.score_chunk(q, k_all, sink, st, en)
            want[l].append(direct)
            if c == 1 and l == 0:
                kv._update_score(l, direct)      # an externally computed block
            else:
                kv._get_score(q, k_all, l)
        kv.slice(seen)
        if c == 2:
            _ = kv.score                          # a read in the middle finalizes what is there; scoring goes on afterwards
    for l in range(L):
        assert torch.equal(kv.score[l].view(torch.int16), torch.cat(want[l], dim=-1).view(torch.int16)), l


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_update_attend_equals_update_prepare_attend(dtype):
    """The fused generation step (append + attention in one launch) against the three separate calls: attention outputs,
    cache rows and bookkeeping must be bit-identical, over enough tokens to cross split boundaries."""
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N = 3, 14, 2, 128, 8, 1500
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(31)

    def make():
        gg = torch.Generator(device=DEV).manual_seed(5)
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dtype, verbose=False, slack=64)
        for l in range(L):
            kv.update(torch.randn(1, Hkv, sink + N, D, generator=gg, device=DEV).to(dtype),
                      torch.randn(1, Hkv, sink + N, D, generator=gg, device=DEV).to(dtype), l)
        kv.score = [torch.rand(1, Hkv, N, generator=gg, device=DEV).to(dtype) for _ in range(L)]
        kv.prune(0.35)
        return kv

    a, b = make(), make()
    for step in range(70):  # more than the initial slack: also exercises _grow_slack on both paths
        for l in range(L):
            q = torch.randn(1, H, 1, D, generator=g, device=DEV).to(dtype)
            # K after RoPE / V out of the projection are strided views in the real forward pass
            kk = torch.randn(1, 1, Hkv, D, generator=g, device=DEV).to(dtype).transpose(1, 2)
            vv = torch.randn(1, 1, Hkv, D, generator=g, device=DEV).to(dtype).transpose(1, 2)
            kf, vf = a.update(kk, vv, l)
            qf, kf, vf, info = a.prepare(q, kf, vf, l)
            want = a.attend(qf, kf, vf, info)
            got = b.update_attend(q, kk, vv, l)
            assert torch.equal(want.view(torch.int16), got.view(torch.int16)), (step, l)
    assert a._seen_tokens == b._seen_tokens and a.info["offset"] == b.info["offset"]
    for l in range(L):
        seg = a.info["seg_start"][l].tolist()
        lens = (a.info["len_k"][l] + a.info["offset"][l]).tolist()
        assert a.info["seg_start"][l].tolist() == b.info["seg_start"][l].tolist()
        for h in range(Hkv):
            assert torch.equal(a.key_cache[l][seg[h]:seg[h] + lens[h]], b.key_cache[l][seg[h]:seg[h] + lens[h]])
            assert torch.equal(a.value_cache[l][seg[h]:seg[h] + lens[h]], b.value_cache[l][seg[h]:seg[h] + lens[h]])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_graph_equals_per_layer_hooks(dtype):
    """A generation step captured in ONE HIP graph (EvictCache.decode_graph: L x (append + attention) + the device-side token
    counter) against the per-layer hook of the Python forward pass (update_attend): outputs, cache rows and bookkeeping bit-identical
    over several tokens, across a slice() (multi-query reuse), with hook steps and graph replays interleaved, and the graph must
    refuse to run once the flat cache has been re-laid out."""
    from kvzip_amd import ops
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N = 3, 14, 2, 128, 8, 1500
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(77)

    def make():
        gg = torch.Generator(device=DEV).manual_seed(5)
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dtype, verbose=False, slack=64)
        for l in range(L):
            kv.update(torch.randn(1, Hkv, sink + N, D, generator=gg, device=DEV).to(dtype),
                      torch.randn(1, Hkv, sink + N, D, generator=gg, device=DEV).to(dtype), l)
        kv.score = [torch.rand(1, Hkv, N, generator=gg, device=DEV).to(dtype) for _ in range(L)]
        kv.prune(0.35)
        return kv

    a, b = make(), make()
    qb = torch.empty(L, 1, H, 1, D, dtype=dtype, device=DEV)
    kb = torch.empty(L, 1, Hkv, 1, D, dtype=dtype, device=DEV)
    vb = torch.empty(L, 1, Hkv, 1, D, dtype=dtype, device=DEV)
    seen0 = a._seen_tokens
    step_graph = None

    def one_token(use_graph):
        qb.copy_(torch.randn(qb.shape, generator=g, device=DEV).to(dtype))
        kb.copy_(torch.randn(kb.shape, generator=g, device=DEV).to(dtype))
        vb.copy_(torch.randn(vb.shape, generator=g, device=DEV).to(dtype))
        want = torch.stack([a.update_attend(qb[l], kb[l], vb[l], l) for l in range(L)])
        if use_graph:
            got = step_graph.replay().clone()
        else:
            got = torch.stack([b.update_attend(qb[l], kb[l], vb[l], l) for l in range(L)])
        assert torch.equal(want.view(torch.int16), got.view(torch.int16))
        assert a._seen_tokens == b._seen_tokens and a.info["offset"] == b.info["offset"]

    one_token(False)                      # a token through the hooks first: the graph is captured mid-generation
    step_graph = b.decode_graph(qb, kb, vb)
    assert b.info["offset"] == a.info["offset"] and b._seen_tokens == a._seen_tokens  # capturing executes nothing
    for _ in range(5):
        one_token(True)
    one_token(False)                      # hooks and replays interleave (the device counter is re-synchronised)
    one_token(True)
    a.slice(seen0); b.slice(seen0)        # next query on the same compressed context
    for _ in range(3):
        one_token(True)
    for l in range(L):
        seg = a.info["seg_start"][l].tolist()
        lens = (a.info["len_k"][l] + a.info["offset"][l]).tolist()
        for h in range(Hkv):
            assert torch.equal(a.key_cache[l][seg[h]:seg[h] + lens[h]], b.key_cache[l][seg[h]:seg[h] + lens[h]])
            assert torch.equal(a.value_cache[l][seg[h]:seg[h] + lens[h]], b.value_cache[l][seg[h]:seg[h] + lens[h]])
    b._grow_slack(200)
    with pytest.raises(ops.KvzError):
        step_graph.replay()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_score_forward_fused_statistics(dtype):
    """f2: the scoring forward's attention kernel emits the row statistics of the scores from its own QK^T tiles
    (KVScore._score_forward -> kvz_flash_fwd_window + kvz_score_from_stats_async_log).  The attention output must be bit-identical
    to the plain forward, the scores must agree with the CPU oracle like the two-pass kernels do, and with the two-pass kernels
    themselves up to the summation order of the row sums (different tile order: a score moves by one step of its grid at most)."""
    from kvzip_amd import ops
    from kvzip_amd.kvcache import EvictCache
    from conftest import check_score_parity
    lib = ops._lib.load()
    H, Hkv, D, sink, N = 28, 4, 128, 32, 1500
    cfg = types.SimpleNamespace(num_hidden_layers=1, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator().manual_seed(123)
    K0 = torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)
    V0 = torch.randn(1, Hkv, sink + N, D, generator=g).to(dtype)
    chunks = [(sink, sink + 700, 713), (sink + 700, sink + N, 826)]
    ins = [(torch.randn(1, H, q_len, D, generator=g).to(dtype), torch.randn(1, Hkv, q_len, D, generator=g).to(dtype),
            torch.randn(1, Hkv, q_len, D, generator=g).to(dtype)) for _, _, q_len in chunks]
    want = torch.cat([orc.get_score(q, torch.cat([K0, kr], dim=2), sink, st, en) for (st, en, _), (q, kr, _) in zip(chunks, ins)], dim=-1)

    def run(fused):
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dtype, verbose=False)
        kv.update(K0.to(DEV), V0.to(DEV), 0)
        kv.init_score()
        kv.fuse_update_score = True
        outs = []
        for (st, en, q_len), (q, kr, vr) in zip(chunks, ins):
            kv.start_idx, kv.end_idx = st, en
            seen = kv._seen_tokens
            k_all, v_all = kv.update(kr.to(DEV), vr.to(DEV), 0)
            qd = q.to(DEV)
            if fused:
                o = kv._score_forward(qd, k_all, v_all, 0)
                assert o is not None
            else:
                kv._get_score(qd, k_all, 0)
                o = ops.flash_fwd(qd, k_all, v_all)
            outs.append(o.clone())
            kv.slice(seen)
        return kv.score[0].cpu(), outs

    prev = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
    prev_s = lib.kvz_debug_set_tunable(b"flash2_split", 0)   # (the window kernel never splits a unit's keys: compare like with like)
    try:
        s_fused, o_fused = run(True)
        s_two, o_two = run(False)
    finally:
        lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev)
        lib.kvz_debug_set_tunable(b"flash2_split", prev_s)
    for a, b in zip(o_fused, o_two):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "the statistics must not change the attention output"
    check_score_parity(f"fused_forward/{dtype}", s_fused, want)
    d = ulp_diff(s_fused, s_two)
    print(f"fused vs two-pass {dtype}: {float((d == 0).float().mean()):.5f} identical, worst {int(d.max())}")
    assert (d == 0).float().mean() >= 0.995 and d.max() <= 2
