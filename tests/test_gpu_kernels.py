"""GPU parity tests proper: every HIP kernel, called through the C ABI, against the golden vectors and the
CPU oracle on seeded inputs.  Integer / byte work (selection masks, compaction, append, metadata) must be
BIT-EXACT; floating point work carries the tolerance stated in the test."""
import math
import types

import numpy as np
import pytest
import torch

import kvzip_oracle as orc
from conftest import ROOT, check_attn, check_mask_flips, check_score_parity, from_bits, load_golden, to_bits, ulp_diff

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def ops():
    from kvzip_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------
# a4 selection
# ------------------------------------------------------------------------------------------------
def _select(score_cpu, ratio, row_len=None):
    valid, thres, kept, rows = ops().select_threshold(score_cpu.to(DEV), ratio, row_len=row_len)
    torch.cuda.synchronize()
    return valid.cpu(), float(thres.item()), int(kept.item()), (rows.cpu() if rows is not None else None)


def test_select_threshold_golden():
    g = load_golden("g2_threshold.npz")
    for tag, bf in (("bf16", True), ("f16", False), ("odd", False)):
        score = from_bits(g[f"{tag}/score"], bf)
        for key in [k for k in g.files if k.startswith(f"{tag}/thres/")]:
            r = key.split("/")[-1]
            row_len = score.shape[-1]
            valid, thres, kept, rows = _select(score, float(r), row_len=row_len)
            want = np.unpackbits(g[f"{tag}/valid/{r}"])[:score.numel()].astype(bool)
            assert np.array_equal(valid.numpy().reshape(-1), want), (tag, r)
            assert thres == g[f"{tag}/thres/{r}"][0], (tag, r)
            assert kept == int(want.sum())
            assert np.array_equal(rows.numpy(), want.reshape(-1, row_len).sum(-1).astype(np.int32))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(28, 1, 4, 4096), (3, 1, 2, 1000), (1, 1, 1, 13)])
def test_select_threshold_vs_oracle(dtype, shape):
    g = torch.Generator().manual_seed(sum(shape) + (1 if dtype == torch.bfloat16 else 0))
    score = (torch.rand(shape, generator=g) ** 6).to(dtype)
    for ratio in (0.0, 0.05, 0.3, 0.5, 0.97, 1.0):
        want_v, want_t = orc.threshold(score, ratio)
        valid, thres, kept, rows = _select(score, ratio, row_len=shape[-1])
        assert torch.equal(valid, want_v), (dtype, shape, ratio)
        assert thres == want_t
        assert kept == int(want_v.sum())
        assert torch.equal(rows, want_v.view(-1, shape[-1]).sum(-1).int())


def test_select_prehist_with_a_foreign_histogram_is_flagged():
    """``kvz_select_threshold_prehist`` trusts that the workspace holds the histogram of exactly these scores.  When
    it does not (here: an all-zero histogram, so the wanted rank lies beyond its total) the kernels must not turn uninitialised LDS
    into a threshold: the threshold comes back NaN and nothing is kept (``EvictCache._select`` then re-runs the plain path and raises)."""
    o = ops()
    g = torch.Generator().manual_seed(3)
    score = (torch.rand(4, 1, 2, 1024, generator=g) ** 3).to(torch.float16).to(DEV)
    ws = o.select_workspace(DEV)
    ws.zero_()
    valid, thres, kept, rows = o.select_threshold(score, 0.3, row_len=1024, prehist=ws)
    assert torch.isnan(thres).all() and int(kept.item()) == 0 and not bool(valid.any())
    valid, thres, kept, rows = o.select_threshold(score, 0.3, row_len=1024)          # the plain path on the same scores
    want, wt = orc.threshold(score.cpu(), 0.3)
    assert float(thres.item()) == wt and torch.equal(valid.cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n_rows,row_len,holes", [(8, 4096, False), (6, 1000, True), (1, 13, True), (3, 2001, False)])
def test_finalize_log_with_histogram_then_select(dtype, n_rows, row_len, holes):
    """Round 4: ``kvz_score_finalize_log_hist`` (log buffer -> 16-bit scores AND the first histogram of the selection in one launch)
    followed by ``kvz_select_threshold_prehist`` (two launches) against the plain finalize + three-launch selection and the oracle:
    scores, masks, thresholds and counts bit-identical - also with entries that were never scored (they keep the value the score
    buffer holds) and sizes that are not multiples of 8."""
    o = ops()
    lib = o._lib.load()
    n = n_rows * row_len
    g = torch.Generator().manual_seed(n + (7 if dtype == torch.bfloat16 else 0))
    logs = -(torch.rand(n, generator=g) * 9.0)                       # log-scores (non-positive fp32)
    logs[::17] = 0.0                                                  # score 1.0
    log_bits = logs.view(torch.int32).clone()
    log_bits[logs == 0.0] = 1                                         # (pass B's encoding of a value >= 0)
    stale = (torch.rand(n, generator=g) ** 4).to(dtype)               # what the score buffer holds where nothing was scored
    if holes:
        empty = torch.rand(n, generator=g) < 0.2
        log_bits[empty] = torch.tensor(0xFF800000 - (1 << 32), dtype=torch.int32)   # the fill pattern (-inf)
    log_d = log_bits.to(DEV)
    a, b = stale.clone().to(DEV), stale.clone().to(DEV)
    st = o._stream(a)
    o.check(lib.kvz_score_finalize_log(log_d.data_ptr(), n, a.data_ptr(), o._dtype_code(dtype), st), "finalize")
    ws = o.select_workspace(DEV)
    o.check(lib.kvz_score_finalize_log_hist(log_d.data_ptr(), n, b.data_ptr(), o._dtype_code(dtype), ws.data_ptr(), ws.numel(), st), "finalize_hist")
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    if holes:
        assert torch.equal(a.cpu()[empty].view(torch.int16), stale[empty].view(torch.int16))
    score = a.view(n_rows, 1, 1, row_len)
    for ratio in (0.0, 0.05, 0.3, 0.6, 0.97):
        o.check(lib.kvz_score_finalize_log_hist(log_d.data_ptr(), n, b.data_ptr(), o._dtype_code(dtype), ws.data_ptr(), ws.numel(), st), "finalize_hist")
        v1, t1, k1, r1 = o.select_threshold(score, ratio, row_len=row_len)
        v2, t2, k2, r2 = o.select_threshold(b.view(score.shape), ratio, row_len=row_len, prehist=ws)
        want_v, want_t = orc.threshold(score.cpu(), ratio)
        assert torch.equal(v1, v2) and torch.equal(v1.cpu(), want_v), (ratio,)
        assert float(t1.item()) == float(t2.item()) == want_t
        assert int(k1.item()) == int(k2.item()) == int(want_v.sum()) and torch.equal(r1, r2)


def test_select_threshold_signs_and_zeros():
    score = torch.tensor([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.0, 65504.0, -65504.0, 6e-8, -6e-8, 0.0],
                         dtype=torch.float16).view(1, 1, 1, -1)
    for ratio in (0.0, 0.2, 0.34, 0.5, 0.75, 0.99):
        want_v, want_t = orc.threshold(score, ratio)
        valid, thres, kept, _ = _select(score, ratio)
        assert torch.equal(valid, want_v), ratio
        assert thres == want_t, ratio  # note 0.0 == -0.0


def test_select_head_level_known_answers():
    """config 5: head-level scores of utils/head_score (qwen2.5-14b at ratio 0.6 keeps 229/384 heads)."""
    g = load_golden("g4_head_score.npz")
    for name in ("qwen2.5-14b", "qwen2.5-7b", "llama3.1-8b"):
        hs = from_bits(g[f"{name}/head_score"], bool(g[f"{name}/is_bf16"][0]))
        for ctx_len in (64, 1000):
            score = hs.unsqueeze(-1).expand(-1, -1, ctx_len).unsqueeze(1).contiguous()
            for r in (0.3, 0.6, 0.9):
                valid, thres, kept, rows = _select(score, r, row_len=ctx_len)
                want = g[f"{name}/kept/{ctx_len}/{r!r}"]
                assert np.array_equal(valid[:, 0, :, 0].numpy(), want)
                assert thres == g[f"{name}/thres/{ctx_len}/{r!r}"][0]
                assert np.array_equal(rows.view(want.shape).numpy(), want.astype(np.int32) * ctx_len)


def test_select_heads_short_circuit():
    """a17: selection on the [L,Hkv] head scores themselves (kvz_select_heads) == the reference's result on the N-fold
    expanded tensor (golden g4 at N = 64 / 1000) == the oracle at the BASELINE context length N = 131072; and == the
    token-level kernel on a materialised expansion (bit-identical mask, threshold, counts)."""
    from kvzip_amd import ops
    g = load_golden("g4_head_score.npz")
    for name in ("qwen2.5-14b", "qwen2.5-7b", "llama3.1-8b"):
        hs = from_bits(g[f"{name}/head_score"], bool(g[f"{name}/is_bf16"][0]))
        hd = hs.to(DEV)
        for ctx_len in (64, 1000):
            for r in (0.3, 0.6, 0.9):
                valid, thres, kept, rows = ops.select_heads(hd, ctx_len, r)
                want = g[f"{name}/kept/{ctx_len}/{r!r}"]
                assert np.array_equal(valid.cpu().numpy(), want)
                assert float(thres) == g[f"{name}/thres/{ctx_len}/{r!r}"][0]
                assert int(kept) == int(want.sum()) * ctx_len
                assert np.array_equal(rows.cpu().view(want.shape).numpy(), want.astype(np.int32) * ctx_len)
        for N in (131072, 32768, 1, 7):
            for r in (0.0, 1e-9, 0.3, 0.6, 0.95, 1.0, 1.5):
                valid, thres, kept, rows = ops.select_heads(hd, N, r)
                wv, wt = orc.threshold_heads(hs, N, r)
                assert torch.equal(valid.cpu(), wv) and float(thres) == wt, (name, N, r)
                assert int(kept) == int(wv.sum()) * N
        # the same selection by the token-level kernel on the materialised expansion
        N = 1000
        score = hd.unsqueeze(-1).expand(-1, -1, N).unsqueeze(1).contiguous()
        for r in (0.3, 0.6):
            v_tok, t_tok, k_tok, rows_tok = ops.select_threshold(score, r, row_len=N)
            v_h, t_h, k_h, rows_h = ops.select_heads(hd, N, r)
            assert torch.equal(v_tok[:, 0, :, 0], v_h) and float(t_tok) == float(t_h) and int(k_tok) == int(k_h)
            assert torch.equal(rows_tok, rows_h)


def test_rowmax_head_score_production():
    """f4: kvz_rowmax16 == torch.stack(score).squeeze().amax(-1) (reference test.py:22-25), fp16 and bf16, odd lengths."""
    from kvzip_amd import ops
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float16, torch.bfloat16):
        for shape in ((3, 2, 1000), (28, 4, 4099), (1, 1, 1), (5, 8, 63)):
            s = (torch.rand(shape, generator=g) ** 4).to(dt)
            s[0, 0, 0] = -0.0
            got = ops.rowmax(s.to(DEV)).cpu()
            assert torch.equal(got, s.amax(-1)), (dt, shape)


def test_compact_heads_equals_token_mask_compaction():
    """Head-level plan + gather (one mask byte per head) == token-level plan + gather on the expanded mask: metadata and
    the flattened K, V are bit-identical, for the packed and the slack layout, with a tail beyond the context."""
    from kvzip_amd import ops
    L, Hkv, D, sink, N, tail = 3, 4, 128, 5, 2300, 9
    klen = sink + N + tail
    g = torch.Generator(device=DEV).manual_seed(2)
    ks = [torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half() for _ in range(L)]
    vs = [torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half() for _ in range(L)]
    heads = torch.rand(L, 1, Hkv, 1, generator=g, device=DEV) < 0.5
    heads[1] = False  # a layer whose context is dropped entirely: only sink and tail survive
    for slack in (0, 16):
        p_h = ops.compact_plan(heads.expand(L, 1, Hkv, N), sink, klen, slack=slack)
        p_t = ops.compact_plan(heads.expand(L, 1, Hkv, N).contiguous(), sink, klen, slack=slack)
        assert p_h.heads and not p_t.heads
        assert torch.equal(p_h.meta, p_t.meta) and torch.equal(p_h.tile_base, p_t.tile_base)
        totals = (p_t.len_k.sum(-1) + Hkv * slack).tolist()
        kh, vh = ops.compact_layers(ks, vs, p_h, totals)
        kt, vt = ops.compact_layers(ks, vs, p_t, totals)
        seg, ln = p_t.seg_start.cpu(), p_t.len_k.cpu()
        for l in range(L):
            for h in range(Hkv):
                a, n = int(seg[l, h]), int(ln[l, h])
                assert n == sink + tail + (N if bool(heads[l, 0, h, 0]) else 0)
                assert torch.equal(kh[l][a:a + n], kt[l][a:a + n]) and torch.equal(vh[l][a:a + n], vt[l][a:a + n])


def test_select_full_size_properties():
    """BASELINE size (Qwen2.5-7B @128k: 28*4*131072 scores): size-independent properties."""
    L, Hkv, N = 28, 4, 131072
    g = torch.Generator(device=DEV).manual_seed(0)
    score = (torch.rand((L, 1, Hkv, N), generator=g, device=DEV) ** 8).to(torch.float16)
    valid, thres, kept, rows = ops().select_threshold(score, 0.3, row_len=N)
    t = thres.item()
    n = score.numel()
    idx = max(int(n * 0.3) - 1, 0)
    # (1) mask == (score > thres); (2) thres is an order statistic: #(> thres) <= idx < #(>= thres)
    assert torch.equal(valid, score.float() > t)
    gt = int((score.float() > t).sum())
    ge = int((score.float() >= t).sum())
    assert gt <= idx < ge
    assert int(kept.item()) == gt and int(rows.sum()) == gt
    assert torch.equal(rows.view(L, Hkv).long(), valid.view(L, Hkv, N).sum(-1))
    # idempotence: selecting again on the same scores gives the same mask
    valid2, thres2, _, _ = ops().select_threshold(score, 0.3, row_len=N)
    assert torch.equal(valid, valid2) and thres2.item() == t


# ------------------------------------------------------------------------------------------------
# a5 uniform top-k
# ------------------------------------------------------------------------------------------------
def test_select_topk_rows_golden():
    g = load_golden("g3_threshold_uniform.npz")
    score = from_bits(g["score"], False)
    N = score.shape[-1]
    for key in [k for k in g.files if k.startswith("valid/")]:
        r = float(key.split("/")[1])
        k = int(N * r) if r < 1 else N
        valid, counts = ops().select_topk_rows(score.to(DEV), k)
        want = np.unpackbits(g[key])[:score.numel()].astype(bool)
        assert np.array_equal(valid.cpu().numpy().reshape(-1), want), r
        assert (counts.cpu() == k).all()


def test_select_topk_rows_contract_on_reference_rows_with_ties():
    """G13 (round 5, reference-generated): pair-uniform selection on rows WITH ties through kvz_select_topk_rows against the masks of
    the reference's _threshold_uniform (attention/score.py:104-120): exactly k per (layer, head), the same kept multiset, the
    identical mask away from each row's boundary value (conftest.check_uniform_contract) - the documented contract on exactly the
    inputs where torch.topk's order is implementation-defined."""
    from conftest import check_uniform_contract
    g = load_golden("g13_uniform_ties.npz")
    for tag, bf in (("bf16", True), ("f16q", False)):
        score = from_bits(g[f"{tag}/score"], bf)
        N = score.shape[-1]
        for r in g["ratios"].tolist():
            k = int(N * r)
            ref = torch.from_numpy(np.unpackbits(g[f"{tag}/valid/{r!r}"])[:score.numel()]).bool().view(score.shape)
            valid, counts = ops().select_topk_rows(score.to(DEV), k)
            assert (counts.cpu() == k).all()
            check_uniform_contract(f"hip/{tag}/{r}", score, valid.cpu(), ref, k)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_select_topk_rows_ties_vs_oracle(dtype):
    g = torch.Generator().manual_seed(11)
    score = (torch.rand((4, 1, 3, 3001), generator=g) ** 3).to(dtype)  # bf16: heavy ties
    for ratio in (0.01, 0.3, 0.77):
        want, _ = orc.threshold_uniform([score[i] for i in range(4)], ratio)
        k = int(3001 * ratio)
        valid, counts = ops().select_topk_rows(score.to(DEV), k)
        assert torch.equal(valid.cpu(), want)
        assert (valid.cpu().view(-1, 3001).sum(-1) == k).all()


# ------------------------------------------------------------------------------------------------
# a8/a9 compaction
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f16_pair", "bf16_pair", "f16_uniform"])
@pytest.mark.parametrize("batched", [False, True])
def test_compact_golden(tag, batched):
    g = load_golden(f"g6_cache_{tag}.npz")
    L, H, Hkv, D, sink, N, bf = g["meta"].tolist()
    K = [from_bits(g[f"K0/{l}"], bf).to(DEV) for l in range(L)]
    V = [from_bits(g[f"V0/{l}"], bf).to(DEV) for l in range(L)]
    valid = torch.from_numpy(g["valid"]).to(DEV)
    plan = ops().compact_plan(valid, sink, sink + N, slack=0)
    meta = plan.meta.cpu()
    torch.cuda.synchronize()
    len_k = plan.len_k.cpu()
    for l in range(L):
        assert np.array_equal(len_k[l].numpy(), g[f"len_k/{l}"])
        assert np.array_equal(plan.cu_len_k[l].cpu().numpy(), g[f"cu_len_k/{l}"])
        assert int(plan.max_len_k[l]) == int(g[f"max_len_k/{l}"][0])
        assert np.array_equal(plan.seg_start[l].cpu().numpy(), g[f"cu_len_k/{l}"][:-1])
    totals = len_k.sum(-1).tolist()
    if batched:
        ko, vo = ops().compact_layers(K, V, plan, totals)
    else:
        ko, vo = zip(*[ops().compact_layer(K[l], V[l], plan, l, totals[l]) for l in range(L)])
    for l in range(L):
        assert np.array_equal(to_bits(ko[l]), g[f"flatK/{l}"])
        assert np.array_equal(to_bits(vo[l]), g[f"flatV/{l}"])


@pytest.mark.parametrize("dtype,D", [(torch.float16, 128), (torch.bfloat16, 128), (torch.float16, 64)])
@pytest.mark.parametrize("slack", [0, 37])
def test_compact_vs_oracle_ragged(dtype, D, slack):
    """ragged masks incl. an empty head, a full head, tile-straddling runs, sink and trailing rows, head stride
    larger than klen*D (views into a cache with spare capacity)."""
    L, Hkv, sink, N, extra = 2, 3, 5, 2500, 9
    klen = sink + N + extra
    cap = klen + 100
    g = torch.Generator().manual_seed(D + slack)
    store_k = [torch.randn(1, Hkv, cap, D, generator=g).to(dtype) for _ in range(L)]
    store_v = [torch.randn(1, Hkv, cap, D, generator=g).to(dtype) for _ in range(L)]
    valid = torch.rand(L, 1, Hkv, N, generator=g) < 0.3
    valid[0, 0, 0] = False
    valid[1, 0, 2] = True
    valid[1, 0, 1, 1000:1100] = True
    K = [s[:, :, :klen] for s in store_k]
    V = [s[:, :, :klen] for s in store_v]
    fk, fv, lens, cus, mxs = orc.prepare_init([k.contiguous() for k in K], [v.contiguous() for v in V], valid, sink)
    Kd = [s.to(DEV)[:, :, :klen] for s in store_k]
    Vd = [s.to(DEV)[:, :, :klen] for s in store_v]
    plan = ops().compact_plan(valid.to(DEV), sink, klen, slack=slack)
    len_k = plan.len_k.cpu()
    for l in range(L):
        assert torch.equal(len_k[l], lens[l])
        assert torch.equal(plan.cu_len_k[l].cpu(), cus[l])
        assert int(plan.max_len_k[l]) == int(mxs[l])
    totals = (len_k.sum(-1) + slack * Hkv).tolist()
    ko, vo = ops().compact_layers(Kd, Vd, plan, totals)
    seg = plan.seg_start.cpu()
    for l in range(L):
        for h in range(Hkv):
            a, n = int(seg[l, h]), int(len_k[l, h])
            c = int(cus[l][h])
            assert torch.equal(ko[l][a:a + n].cpu(), fk[l][c:c + n])
            assert torch.equal(vo[l][a:a + n].cpu(), fv[l][c:c + n])


def test_compact_all_kept_and_none_kept():
    L, Hkv, sink, N, D = 1, 2, 0, 1024, 128
    k = torch.randn(1, Hkv, N, D).half().to(DEV)
    v = torch.randn(1, Hkv, N, D).half().to(DEV)
    for fill in (True, False):
        valid = torch.full((L, 1, Hkv, N), fill, dtype=torch.bool, device=DEV)
        plan = ops().compact_plan(valid, sink, N)
        tot = int(plan.len_k.sum())
        assert tot == (Hkv * N if fill else 0)
        ko, vo = ops().compact_layer(k, v, plan, 0, tot)
        if fill:
            assert torch.equal(ko, k.view(-1, D)) and torch.equal(vo, v.view(-1, D))


# ------------------------------------------------------------------------------------------------
# a10/a11 append
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("t", [1, 7])
def test_update_flatten_view_vs_oracle(dtype, t):
    Hkv, D = 4, 128
    g = torch.Generator().manual_seed(t)
    lens = torch.tensor([5, 0, 133, 40], dtype=torch.int32)
    offset = 3
    head_lens = lens + offset
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), head_lens.cumsum(0).int()])
    cache = torch.randn(int(cu[-1]), D, generator=g).to(dtype)
    state = torch.randn(Hkv * t, D, generator=g).to(dtype)
    want = orc.update_flatten_view(cache, state, head_lens, cu)
    got = ops().update_flatten_view(cache.to(DEV), state.to(DEV), head_lens.to(DEV), cu.to(DEV))
    assert torch.equal(got.cpu(), want)


def test_update_flatten_view_dtype_checks():
    o = ops()
    cache = torch.zeros(4, 128, dtype=torch.float16, device=DEV)
    state = torch.zeros(2, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="headlens to be int32"):
        o.update_flatten_view(cache, state, torch.tensor([2, 2], device=DEV), torch.tensor([0, 2, 4], dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match="divisible"):
        o.update_flatten_view(cache, torch.zeros(3, 128, dtype=torch.float16, device=DEV),
                              torch.tensor([2, 2], dtype=torch.int32, device=DEV),
                              torch.tensor([0, 2, 4], dtype=torch.int32, device=DEV))


def test_append_inplace_matches_rebuild():
    """O(t) slack append == the reference's out-of-place rebuild, head by head."""
    Hkv, D, slack = 3, 128, 16
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([7, 0, 50], dtype=torch.int32)
    seg = torch.tensor([0, 7 + slack, 7 + slack + 0 + slack], dtype=torch.int32)
    total = int(lens.sum()) + Hkv * slack
    kc = torch.randn(total, D, generator=g).half().to(DEV)
    vc = torch.randn(total, D, generator=g).half().to(DEV)
    kc0, vc0 = kc.clone(), vc.clone()
    cur = lens.clone().to(DEV)
    for t in (5, 1, 1):
        ks = torch.randn(1, Hkv, t, D, generator=g).half().to(DEV)
        vs = torch.randn(1, Hkv, t, D, generator=g).half().to(DEV)
        ops().append_inplace(kc, vc, ks, vs, seg.to(DEV), cur)
        for h in range(Hkv):
            a = int(seg[h]) + int(cur[h])
            assert torch.equal(kc[a:a + t], ks[0, h]) and torch.equal(vc[a:a + t], vs[0, h])
        cur += t
    for h in range(Hkv):  # untouched prefix
        a, n = int(seg[h]), int(lens[h])
        assert torch.equal(kc[a:a + n], kc0[a:a + n]) and torch.equal(vc[a:a + n], vc0[a:a + n])


# ------------------------------------------------------------------------------------------------
# a1 scoring
# ------------------------------------------------------------------------------------------------
G1 = load_golden("g1_score.npz")


def _score_stats(got, want):
    d = ulp_diff(got, want)
    return float((d == 0).float().mean()), float((d <= 1).float().mean()), int(d.max())


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in G1.files}))
def test_score_chunk_golden(name):
    """Tolerance (north_star: scores feed an integer-exact selection; masks are bit-exact GIVEN the scores):
    the kernel follows the reference's rounding chain; differences come only from fp32 accumulation order
    inside the MFMA and the exp/log implementation: >= 97% of scores bit-identical, >= 99.5% within 1 half-ulp,
    none beyond 8 half-ulps (a 1-ulp flip of a LOGIT moves the probability by several ulps)."""
    Hkv, G, D, sink, start, end, q_len, klen, bf = G1[name + "/meta"].tolist()
    q = from_bits(G1[name + "/q"], bf).to(DEV)
    k = from_bits(G1[name + "/k"], bf).to(DEV)
    want = from_bits(G1[name + "/score"], bf)
    got = ops().score_chunk(q, k, sink, start, end).cpu()
    assert got.shape == want.shape
    check_score_parity(f"golden/{name}", got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_score_chunk_vs_oracle_multi_tile(dtype):
    """shapes that exercise several 128-row tiles, ragged tails, a head stride with slack and G=7."""
    Hkv, G, D, sink, N, q_len = 2, 7, 128, 32, 700, 413
    start, end = sink + 150, sink + 550
    klen, cap = sink + N + q_len, sink + N + q_len + 64
    g = torch.Generator().manual_seed(21)
    q = torch.randn(1, Hkv * G, q_len, D, generator=g).to(dtype)
    kstore = torch.randn(1, Hkv, cap, D, generator=g).to(dtype)
    want = orc.get_score(q, kstore[:, :, :klen].contiguous(), sink, start, end)
    got = ops().score_chunk(q.to(DEV), kstore.to(DEV)[:, :, :klen], sink, start, end).cpu()
    check_score_parity(f"multi_tile/{dtype}", got, want)


@pytest.mark.parametrize("H,Hkv,D,sink,N,start,m,q_len", [
    (4, 2, 128, 16, 400, 16 + 37, 304, 200),     # 520 keys = 5 key tiles: every row tile ends with a key slice of ONE tile
    (4, 2, 128, 0, 300, 10, 60, 60),             # 120 keys: a single key tile, items of one tile only (starved stream)
    (2, 1, 128, 8, 900, 8 + 100, 520, 129),      # Hkv = 1, 9 tiles; 258 rows -> a second row tile with 2 valid rows
    (6, 6, 64, 4, 700, 4 + 3, 5, 300),           # G = 1, five ctx keys only, D = 64
    (8, 2, 128, 32, 640, 32 + 128, 3, 7),        # three ctx keys, seven query rows
    (16, 2, 64, 30, 1200, 30 + 400, 513, 77),    # G = 8, 620 keys (5 tiles), rows spanning several query heads per tile
    (28, 4, 128, 32, 2100, 32, 2000, 2013),      # first chunk of the headline geometry (q = m + 13)
])
def test_score_chunk_edge_shapes(H, Hkv, D, sink, N, start, m, q_len):
    """Shapes that exercise the corners of the persistent pipeline: key slices of a single tile (late staging path), work lists
    shorter than the grid, partial row tiles, G = 1 / Hkv = 1, a handful of ctx keys or query rows."""
    g = torch.Generator().manual_seed(H * 1000 + m)
    q = torch.randn(1, H, q_len, D, generator=g).half()
    k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g).half()
    want = orc.get_score(q, k, sink, start, start + m)
    got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, start, start + m).cpu()
    check_score_parity(f"edge/{H}-{Hkv}-{D}-{sink}-{N}-{start}-{m}-{q_len}", got, want)
    again = ops().score_chunk(q.to(DEV), k.to(DEV), sink, start, start + m).cpu()
    assert torch.equal(got.view(torch.int16), again.view(torch.int16))


def test_score_chunk_fuzz_small_shapes_vs_oracle():
    """60 random small shapes (both dtypes, both head dims, G and Hkv from 1 to 8, windows with gaps, 1..400 ctx keys and query
    rows) against the CPU oracle.  Complements tools/fuzz_score.py, which sweeps large shapes against a second build."""
    import random
    rng = random.Random(2024)
    g = torch.Generator().manual_seed(99)
    all_got, all_want, all_bf = [], [], []
    for n in range(60):
        Hkv, G, D = rng.choice([1, 2, 3, 8]), rng.choice([1, 2, 4, 7]), rng.choice([64, 128])
        dtype = torch.bfloat16 if rng.random() < 0.3 else torch.float16
        sink, m, q_len = rng.choice([0, 1, 16, 30]), rng.randint(1, 400), rng.randint(1, 400)
        start = sink + rng.randint(0, 150)
        klen = start + m + rng.randint(0, 150) + q_len
        q = torch.randn(1, Hkv * G, q_len, D, generator=g).to(dtype)
        k = torch.randn(1, Hkv, klen, D, generator=g).to(dtype)
        want = orc.get_score(q, k, sink, start, start + m)
        got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, start, start + m).cpu()
        all_got.append(got.reshape(-1).view(torch.int16))
        all_want.append(want.reshape(-1).view(torch.int16))
        all_bf.append(torch.full((got.numel(),), dtype == torch.bfloat16))
    # one distribution over the 60 shapes (per dtype: the 16-bit grids differ)
    bf = torch.cat(all_bf)
    for tag, sel, dt in (("f16", ~bf, torch.float16), ("bf16", bf, torch.bfloat16)):
        check_score_parity(f"fuzz60/{tag}", torch.cat(all_got)[sel].view(dt), torch.cat(all_want)[sel].view(dt))


@pytest.mark.parametrize("shape", [(14, 2, 64, 30, 2048, 30, 2030, 2013), (28, 4, 128, 32, 8192, 4032, 6032, 2026),
                                   (32, 8, 128, 32, 3000, 732, 2732, 2026), (8, 2, 128, 16, 900, 16, 916, 37)])
def test_score_chunk_is_deterministic(shape):
    """Race detector for the hand-pipelined kernels (LDS-DMA, bare barriers, register prefetch): the same inputs must give
    bit-identical scores run after run, with an unrelated launch of another shape in between.  (The accuracy tests
    above tolerate a few differing scores, so a rare data race - one was found this way during development - would
    pass them.)"""
    H, Hkv, D, sink, N, start, end, q_len = shape
    g = torch.Generator(device=DEV).manual_seed(5)
    other_q = torch.randn(1, 8, 300, 128, generator=g, device=DEV).half()
    other_k = torch.randn(1, 2, 16 + 700 + 300, 128, generator=g, device=DEV).half()
    for it in range(4):
        q = torch.randn(1, H, q_len, D, generator=g, device=DEV).half()
        k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g, device=DEV).half()
        first = ops().score_chunk(q, k, sink, start, end).view(torch.int16).clone()
        for _ in range(3):
            ops().score_chunk(other_q, other_k, 16, 100, 600)
            again = ops().score_chunk(q, k, sink, start, end).view(torch.int16)
            assert torch.equal(first, again), f"iteration {it}: {int((first != again).sum())} scores changed between runs"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["spiky", "negative", "huge_first"])
def test_score_chunk_reference_moves(dtype, kind):
    """The row-statistics kernel sums exponentials against a REFERENCE that only moves when a block's sum leaves [2^-20, 2^16]
    (cold path).  N(0,1) data never takes that path, so: logits with std ~6 and keys every query loves (+35) at scattered
    positions (reference moves up, several times per row), every logit near -40 (reference moves down at the first block),
    and a first block that holds the global maximum followed by tiny logits.  Same tolerances as everywhere."""
    H, Hkv, D, sink, m, q_len, off = 8, 2, 128, 16, 700, 713, 128
    g = torch.Generator().manual_seed(7)
    q = torch.randn(1, H, q_len, D, generator=g)
    k = torch.randn(1, Hkv, sink + off + m + q_len + 64, D, generator=g)
    if kind == "spiky":
        q, k = q * 2.5, k * 2.5
        for pos in (3, sink + off + 17, sink + off + 400, k.shape[2] - 70):
            k[:, :, pos] = q[:, ::H // Hkv, min(pos, q_len - 1)] * 0.35
    elif kind == "negative":
        u = torch.ones(D) / D ** 0.5
        q, k = q * 0.3 + 22.0 * u, k * 0.3 - 22.0 * u
    else:
        k[:, :, 0] = q[:, ::H // Hkv, :].mean(dim=2) * 6.0
        k[:, :, 1:] *= 0.05
    q, k = q.to(dtype), k.to(dtype)
    want = orc.get_score(q, k, sink, sink + off, sink + off + m)
    got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, sink + off, sink + off + m).cpu()
    assert not torch.isnan(got.float()).any()
    check_score_parity(f"reference_moves/{kind}/{dtype}", got, want)


def test_score_then_select_end_to_end_hamming():
    """End to end: masks from HIP scores vs masks from oracle scores — Hamming distance reported and bounded."""
    Hkv, G, D, sink, N, q_len = 2, 4, 128, 16, 512, 270
    klen = sink + N + q_len
    g = torch.Generator().manual_seed(4)
    q = torch.randn(1, Hkv * G, q_len, D, generator=g).half()
    k = torch.randn(1, Hkv, klen, D, generator=g).half()
    want = orc.get_score(q, k, sink, sink, sink + N)
    got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, sink, sink + N)
    v_ref, t_ref = orc.threshold(want, 0.3)
    v_hip, _, _, _ = ops().select_threshold(got, 0.3)
    check_mask_flips("score_then_select", got.cpu(), want, v_hip, v_ref, t_ref, allowed=0)   # measured 0 of 1024 in every round
    # and the contract that matters: on the ORACLE's scores the HIP mask is bit-exact
    v_hip2, _, _, _ = ops().select_threshold(want.to(DEV), 0.3)
    assert torch.equal(v_hip2.cpu(), v_ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_score_chunk_headline_shape_parity_distribution(dtype):
    """The shape that is 100 % of the bench (Qwen2.5-7B: H28 Hkv4 D128, m = 2000, q = 2026, later chunk): distribution of the
    score differences against the CPU oracle and the mask Hamming distance at ratio 0.3 on the same tensors, printed and
    bounded (bench.py emits the same numbers as `parity_sample`)."""
    H, Hkv, D, sink, m = 28, 4, 128, 32, 2000
    q_len = m + 26
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, H, q_len, D, generator=g).to(dtype)
    k = torch.randn(1, Hkv, sink + m + q_len, D, generator=g).to(dtype)
    want = orc.get_score(q, k, sink, sink, sink + m)
    got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, sink, sink + m).cpu()
    d = ulp_diff(got, want)
    exact, within1, worst = float((d == 0).float().mean()), float((d <= 1).float().mean()), int(d.max())
    v_ref, t_ref = orc.threshold(want.unsqueeze(0), 0.3)
    v_hip, t_hip = orc.threshold(got.unsqueeze(0), 0.3)
    print(f"headline shape {dtype}: {exact:.5f} bit-identical, {within1:.5f} within 1 half-ulp, worst {worst}")
    check_score_parity(f"headline/{dtype}", got, want)
    assert exact >= HEADLINE_EXACT[dtype] and within1 >= 0.9997 and worst <= 4
    assert t_hip == t_ref
    check_mask_flips(f"headline/{dtype}", got, want, v_hip, v_ref, t_ref, allowed=0)   # measured 0 of 8000 in rounds 2-4, both dtypes


# measured on MI355X with the round-2 kernels (profiles/r2_parity_headline.txt): fp16 99.937 % bit-identical / 99.988 % within one
# half-ulp / worst 2, bf16 99.988 % / 100 % / worst 1, mask Hamming distance 0 of 8000 for both.  The bounds are the measured
# non-identical fractions with a margin of 2x; the mask: threshold equal, no flipped entry (round 4; conftest.check_mask_flips).
HEADLINE_EXACT = {torch.float16: 0.9987, torch.bfloat16: 0.99975}


# ------------------------------------------------------------------------------------------------
# a13 variable-length attention   (tolerance from north_star: 1e-3 absolute in fp16)
# ------------------------------------------------------------------------------------------------
G7 = load_golden("g7_attn.npz")


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in G7.files}))
def test_varlen_attn_golden(name):
    Hkv, G, D, q_len, bf = G7[name + "/meta"].tolist()
    q = from_bits(G7[name + "/q"], bf).to(DEV)
    k = from_bits(G7[name + "/k"], bf).to(DEV)
    v = from_bits(G7[name + "/v"], bf).to(DEV)
    ks = torch.from_numpy(G7[name + "/k_start"]).to(DEV)
    kl = torch.from_numpy(G7[name + "/k_len"]).to(DEV)
    want = from_bits(G7[name + "/out"], bf).float()
    got = ops().varlen_attn(q, k, v, ks, kl, q_len, int(kl.max()), causal=True).cpu().float()
    tol = 1e-3 if not bf else 8e-3  # bf16 output spacing is 2^-8 relative
    check_attn(f"varlen_attn_golden/{name}", got, want, tol)


@pytest.mark.parametrize("q_len", [1, 5])
def test_varlen_attn_long_ragged_vs_oracle(q_len):
    Hkv, G, D = 4, 7, 128
    lens = [4097, 33, 1500, 0 + q_len]
    g = torch.Generator().manual_seed(q_len)
    starts, tot = [], 0
    for ln in lens:
        starts.append(tot)
        tot += ln + 3
    q = torch.randn(Hkv * q_len, G, D, generator=g).half()
    k = torch.randn(tot, D, generator=g).half()
    v = torch.randn(tot, D, generator=g).half()
    want = orc.varlen_attn(q, k, v, starts, lens, q_len).float()
    got = ops().varlen_attn(q.to(DEV), k.to(DEV), v.to(DEV), torch.tensor(starts, dtype=torch.int32, device=DEV),
                            torch.tensor(lens, dtype=torch.int32, device=DEV), q_len, max(lens)).cpu().float()
    assert (got - want).abs().max() <= 1e-3


def test_decode_attn_margin_over_seeds():
    """Round 4: the achieved error of the DECODE call (q_len = 1) against the oracle over 24 seeds and
    ragged head lengths, next to north_star's 1e-3 (fp16) - printed and recorded as a distribution, so that the margin of the
    bound is a number and not one lucky seed.  Short heads (a few keys) are where single outputs are largest."""
    Hkv, G, D = 4, 7, 128
    worst = []
    for seed in range(24):
        g = torch.Generator().manual_seed(1000 + seed)
        lens = [int(x) for x in torch.randint(1, 6000, (Hkv,), generator=g)]
        if seed % 4 == 0:
            lens[seed % Hkv] = 1 + seed % 7          # a head with a handful of keys
        starts, tot = [], 0
        for ln in lens:
            starts.append(tot)
            tot += ln + 5
        q = torch.randn(Hkv, G, D, generator=g).half()
        k = torch.randn(tot, D, generator=g).half()
        v = torch.randn(tot, D, generator=g).half()
        want = orc.varlen_attn(q, k, v, starts, lens, 1).float()
        got = ops().varlen_attn(q.to(DEV), k.to(DEV), v.to(DEV), torch.tensor(starts, dtype=torch.int32, device=DEV),
                                torch.tensor(lens, dtype=torch.int32, device=DEV), 1, max(lens)).cpu().float()
        worst.append(float((got - want).abs().max()))
        check_attn(f"decode_margin/seed{seed}", got, want, 1e-3, ulp_of=torch.float16)
    w = sorted(worst)
    print(f"\nDECODE MARGIN over {len(w)} seeds: max |err| min {w[0]:.2e} median {w[len(w) // 2]:.2e} max {w[-1]:.2e} (bound 1e-3)")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_varlen_attn_ragged_items_and_workspace_reuse(dtype):
    """The one-launch attention: work items cut from the SUM of very ragged head lengths (AdaKV-style budgets, a dropped head,
    an empty head), head segments given as kernel arguments or read from the device arrays, the SAME workspace reused call
    after call (the arrival counters must come back to zero), fused append variant included.  Oracle tolerance 1e-3 (fp16)."""
    from kvzip_amd import ops
    Hkv, G, D = 6, 7, 128
    lens = [39000, 17, 120000, 500, 0, 4100]
    slack = 8
    starts, acc = [], 0
    for n in lens:
        starts.append(acc)
        acc += n + slack
    g = torch.Generator(device=DEV).manual_seed(21)
    k = torch.randn(acc, D, generator=g, device=DEV).to(dtype)
    v = torch.randn(acc, D, generator=g, device=DEV).to(dtype)
    ks = torch.tensor(starts, dtype=torch.int32, device=DEV)
    kl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    tol = 1e-3 if dtype == torch.float16 else 8e-3
    ws = ops.attn_workspace(Hkv, G, 1, D, DEV)
    meta = ops._meta_host(starts, lens, Hkv)
    for it in range(4):
        q = torch.randn(Hkv, G, D, generator=g, device=DEV).to(dtype)
        want = orc.varlen_attn(q.cpu(), k.cpu(), v.cpu(), starts, lens, 1).float()
        a = ops.varlen_attn(q, k, v, ks, kl, 1, max(lens), workspace=ws, meta_host=meta if it % 2 == 0 else None)
        check_attn(f"varlen_attn_ragged_items_and_workspace_reuse/{dtype}/item{it}", a, want, tol)
        assert torch.equal(a[4].cpu().float(), torch.zeros(G, D))  # the empty head: zeros, like flash-attn
    # fused append on the same workspace: one new token per head, then the same answer as append + attention
    kn = torch.randn(1, Hkv, 1, D, generator=g, device=DEV).to(dtype)
    vn = torch.randn(1, Hkv, 1, D, generator=g, device=DEV).to(dtype)
    q = torch.randn(Hkv, G, D, generator=g, device=DEV).to(dtype)
    k2, v2 = k.clone(), v.clone()
    got = ops.varlen_attn_append(q, k, v, kn, vn, ks, kl, 0, max(lens) + 1, workspace=ws, meta_host=meta)
    ops.append_inplace(k2, v2, kn, vn, ks, kl, 0)
    ref = ops.varlen_attn(q, k2, v2, ks, kl, 1, max(lens) + 1, k_len_offset=1)
    assert torch.equal(got, ref) and torch.equal(k, k2) and torch.equal(v, v2)
    lens1 = [n + 1 for n in lens]
    want = orc.varlen_attn(q.cpu(), k2.cpu(), v2.cpu(), starts, lens1, 1).float()
    check_attn(f"varlen_attn_ragged_items_and_workspace_reuse/{dtype}", got, want, tol)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("q_len", [64, 257])
def test_varlen_attn_many_query_rows_ragged_on_the_32_row_kernel(q_len, dtype):
    """The same ragged first-generation-step call forced onto the 32-row kernel (kvz_flash2.hip; it takes such calls by itself from
    16 (head, 256-row) units on, i.e. for queries of > 100 tokens) - with the workspace, so the keys of every unit are cut into parts
    (heads of 4 k / 33 / 1.5 k / 0 context keys + the query's own: parts of very different lengths, empty ones included) and merged
    by the second launch; by-value and device-array metadata."""
    from kvzip_amd import ops
    lib = ops._lib.load()
    Hkv, G, D = 4, 7, 128
    lens = [4097 + q_len, 33 + q_len, 1500 + q_len, q_len]
    g = torch.Generator().manual_seed(q_len + 5)
    starts, tot = [], 0
    for ln in lens:
        starts.append(tot)
        tot += ln + 5
    q = torch.randn(Hkv * q_len, G, D, generator=g).to(dtype)
    k = torch.randn(tot, D, generator=g).to(dtype)
    v = torch.randn(tot, D, generator=g).to(dtype)
    want = orc.varlen_attn(q, k, v, starts, lens, q_len).float()
    ks = torch.tensor(starts, dtype=torch.int32, device=DEV)
    kl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    tol = 1e-3 if dtype == torch.float16 else 8e-3   # (one output step where the grid is coarser than that: conftest.check_attn)
    prev = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
    try:
        for meta in (None, ops._meta_host(starts, lens, Hkv)):
            got = ops.varlen_attn(q.to(DEV), k.to(DEV), v.to(DEV), ks, kl, q_len, max(lens), meta_host=meta).cpu().float()
            check_attn(f"varlen_attn_ragged_flash2/{q_len}/{dtype}/meta{meta is not None}", got, want, tol, ulp_of=dtype)
    finally:
        lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("q_len", [64, 257])
def test_varlen_attn_many_query_rows_vs_oracle(q_len, dtype):
    """First generation step on a pruned cache (q_len = len(query) > 1, reference model/wrapper.py:271-274): kvz_varlen_attn
    hands q_len*G > 64 rows to the multi-row kernel (one pass over the keys per 128-row block).  Ragged heads incl. one that
    holds only the new tokens, device arrays and by-value segments.  Oracle tolerance 1e-3 (fp16) / 8e-3 (bf16) absolute plus
    one output ulp (2^-10 / 2^-7 relative): rows that see a handful of keys return values of magnitude 2-4, where the fp16
    grid itself is 2e-3 wide."""
    from kvzip_amd import ops
    Hkv, G, D = 4, 7, 128
    lens = [4097 + q_len, 33 + q_len, 1500 + q_len, q_len]
    g = torch.Generator().manual_seed(q_len)
    starts, tot = [], 0
    for ln in lens:
        starts.append(tot)
        tot += ln + 5
    q = torch.randn(Hkv * q_len, G, D, generator=g).to(dtype)
    k = torch.randn(tot, D, generator=g).to(dtype)
    v = torch.randn(tot, D, generator=g).to(dtype)
    want = orc.varlen_attn(q, k, v, starts, lens, q_len).float()
    ks = torch.tensor(starts, dtype=torch.int32, device=DEV)
    kl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    tol = 1e-3 if dtype == torch.float16 else 8e-3   # (one output step where the grid is coarser than that: conftest.check_attn)
    for meta in (None, ops._meta_host(starts, lens, Hkv)):
        got = ops.varlen_attn(q.to(DEV), k.to(DEV), v.to(DEV), ks, kl, q_len, max(lens), meta_host=meta).cpu().float()
        check_attn(f"varlen_attn_many_query_rows_vs_oracle/{q_len}/{dtype}/meta{meta is not None}", got, want, tol, ulp_of=dtype)
    # the decode kernel (16-row tiles, split keys) must agree with the multi-row kernel on the same call
    if q_len == 64:
        lib = ops._lib.load()
        torch.manual_seed(0)
        q2 = torch.randn(4 * 64, 7, 128, device=DEV).to(dtype); k2 = torch.randn(9000, 128, device=DEV).to(dtype)
        v2 = torch.randn(9000, 128, device=DEV).to(dtype)
        ks2 = torch.tensor([0, 3000, 5000, 8000], dtype=torch.int32, device=DEV)
        kl2 = torch.tensor([2900, 1900, 2500, 64], dtype=torch.int32, device=DEV)
        outs = []
        for rows in (64, 100000):
            prev = lib.kvz_debug_set_tunable(b"flash_min_rows", rows)
            try:
                outs.append(ops.varlen_attn(q2, k2, v2, ks2, kl2, 64, 2900).float())
            finally:
                lib.kvz_debug_set_tunable(b"flash_min_rows", prev)
        # (two kernels, each within one output step of the exact value)
        from conftest import grid_step
        assert ((outs[0] - outs[1]).abs().cpu() <= 2 * torch.maximum(torch.full_like(outs[1], tol), grid_step(outs[1], dtype)).cpu()).all(), \
            float((outs[0] - outs[1]).abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(28, 4, 300, 300, 128), (28, 4, 130, 1500, 128), (8, 8, 65, 200, 64), (6, 2, 1, 77, 128)])
def test_flash_fwd_dense_vs_fp32_reference(shape, dtype):
    """f2, the dense pre-prune forward (reference attention/attn.py:75-89): [1,H,q,D] x [1,Hkv,k,D] views of a LARGER dense
    cache (head stride = capacity), bottom-right causal mask, output [1,q,H,D]; plus the row LSE.  fp32 reference with the
    same mask; tolerance 1e-3 (fp16) / 8e-3 (bf16) + one output ulp on the output, 1e-3 on the LSE."""
    from kvzip_amd import ops
    H, Hkv, q_len, klen, D = shape
    G = H // Hkv
    g = torch.Generator(device=DEV).manual_seed(q_len + klen)
    cap = klen + 37
    kc = torch.randn(1, Hkv, cap, D, generator=g, device=DEV).to(dtype)
    vc = torch.randn(1, Hkv, cap, D, generator=g, device=DEV).to(dtype)
    qq = torch.randn(1, q_len, H, D, generator=g, device=DEV).to(dtype).transpose(1, 2)  # non-contiguous, like q_proj().view()
    key, val = kc[:, :, :klen], vc[:, :, :klen]
    out, lse = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)
    assert out.shape == (1, q_len, H, D) and lse.shape == (1, H, q_len)
    s = torch.einsum("hid,hjd->hij", qq[0].float(), key[0].float().repeat_interleave(G, 0)) / math.sqrt(D)
    i = torch.arange(q_len, device=DEV).view(1, q_len, 1)
    j = torch.arange(klen, device=DEV).view(1, 1, klen)
    s = s.masked_fill(j > i + (klen - q_len), float("-inf"))
    want = torch.einsum("hij,hjd->ihd", torch.softmax(s, -1), val[0].float().repeat_interleave(G, 0))
    tol = 1e-3 if dtype == torch.float16 else 8e-3   # (one output step where the grid is coarser than that: conftest.check_attn)
    check_attn(f"flash_fwd_dense_vs_fp32_reference/{shape}/{dtype}", out[0], want.to(dtype), tol, ulp_of=dtype)   # (the oracle's contract: fp32 result rounded once)
    assert (lse[0] - torch.logsumexp(s, -1)).abs().max() <= 1e-3
    # the attention hook of the model forward goes through the same kernel
    from kvzip_amd.attn import dense_causal_attention
    out2, _ = dense_causal_attention(None, qq, key, val, scaling=1.0 / math.sqrt(D))
    assert torch.equal(out2, out)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(28, 4, 300, 300), (28, 4, 130, 1500), (8, 2, 257, 1000), (6, 2, 70, 77), (16, 16, 129, 700), (7, 1, 1100, 2300),
                                   (24, 8, 300, 900), (6, 3, 260, 500), (4, 4, 1300, 1300)])
def test_flash2_dense_vs_fp32_reference(shape, dtype):
    """The 32-row forward (kvz_flash2.hip: 256-row blocks, 32x32x16 MFMA, LDS-DMA ring) forced onto shapes of every kind - several
    row tiles, a partial last row tile, fewer keys than a tile, a ragged last key tile, G = 1 and Hkv = 1, rows that see a single key -
    against an fp32 reference with the same bottom-right mask, and against the 16-row kernel on the same call."""
    from kvzip_amd import ops
    lib = ops._lib.load()
    H, Hkv, q_len, klen = shape
    D, G = 128, shape[0] // shape[1]
    g = torch.Generator(device=DEV).manual_seed(q_len * 7 + klen)
    cap = klen + 19
    kc = torch.randn(1, Hkv, cap, D, generator=g, device=DEV).to(dtype)
    vc = torch.randn(1, Hkv, cap, D, generator=g, device=DEV).to(dtype)
    qq = torch.randn(1, q_len, H, D, generator=g, device=DEV).to(dtype).transpose(1, 2)
    key, val = kc[:, :, :klen], vc[:, :, :klen]
    prev = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
    try:
        out, lse = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)
        again, lse_again = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)
    finally:
        lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1 << 30)
    try:
        old, lse_old = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)   # the 16-row kernel
    finally:
        lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev)
    assert torch.equal(out, again) and torch.equal(lse, lse_again)
    # round 4: the XCD-aware block order (a head's row tiles on 8 / Hkv XCDs; Hkv = 1, 2, 4, 8: XCDs per head, 16: heads per XCD,
    # 3: head-major fallback; partial groups when 8 / Hkv does not divide the row tiles) only permutes blocks: with one block per
    # (head, row tile) - the balanced partition switched off - the bits are those of the head-major grid
    variants = {}
    # (flash2_split: 0 = one block per unit, 1 = split the last round of blocks along the keys when it saves a tenth of the rounds,
    # 2 = whenever the units do not fill the last round - what these small shapes need to get there at all)
    for name, xcd, split in (("unit_xcd", 1, 0), ("unit_plain", 0, 0), ("split_xcd", 1, 2), ("split_plain", 0, 2)):
        prev_x = lib.kvz_debug_set_tunable(b"flash2_xcd", xcd)
        prev_s = lib.kvz_debug_set_tunable(b"flash2_split", split)
        prev_b = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
        try:
            variants[name] = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)
        finally:
            lib.kvz_debug_set_tunable(b"flash2_xcd", prev_x)
            lib.kvz_debug_set_tunable(b"flash2_split", prev_s)
            lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev_b)
        assert prev_x == 1 and prev_s == 1
    assert torch.equal(variants["unit_xcd"][0], variants["unit_plain"][0]) and torch.equal(variants["unit_xcd"][1], variants["unit_plain"][1])
    s = torch.einsum("hid,hjd->hij", qq[0].float(), key[0].float().repeat_interleave(G, 0)) / math.sqrt(D)
    i = torch.arange(q_len, device=DEV).view(1, q_len, 1)
    j = torch.arange(klen, device=DEV).view(1, 1, klen)
    s = s.masked_fill(j > i + (klen - q_len), float("-inf"))
    want = torch.einsum("hij,hjd->ihd", torch.softmax(s, -1), val[0].float().repeat_interleave(G, 0))
    tol = 1e-3 if dtype == torch.float16 else 8e-3   # (one output step where the grid is coarser than that: conftest.check_attn)
    err = (out[0].float() - want).abs()
    print(f"\nflash2 {shape} {dtype}: max |err| {float(err.max()):.2e} (16-row kernel: {float((old[0].float() - want).abs().max()):.2e})")
    check_attn(f"flash2_dense_vs_fp32_reference/{shape}/{dtype}", out[0], want.to(dtype), tol, ulp_of=dtype)   # (the oracle's contract: fp32 result rounded once)
    assert (lse[0] - torch.logsumexp(s, -1)).abs().max() <= 1e-3
    # the split cuts a unit's keys at other places with another block order, and not at all without it: same contract for each
    for name, (o_v, lse_v) in variants.items():
        check_attn(f"flash2_dense_vs_fp32_reference/{name}/{shape}/{dtype}", o_v[0], want.to(dtype), tol, ulp_of=dtype)
        assert (lse_v[0] - torch.logsumexp(s, -1)).abs().max() <= 1e-3
    from conftest import grid_step
    assert ((out.float() - old.float()).abs().cpu() <= 2 * torch.maximum(torch.full_like(want, tol), grid_step(want, dtype)).unsqueeze(0).cpu()).all()


@pytest.mark.parametrize("split", [2, 0])
def test_flash2_rows_without_keys_and_uneven_units(split):
    """The split last round of the 32-row forward on a call whose units differ as much as they can: fewer keys than query positions,
    so the first row tiles see NO key (they weigh one virtual key tile and must still write zeros and lse = -inf), the last ones a
    growing causal prefix.  Same answer with and without the split."""
    from kvzip_amd import ops
    lib = ops._lib.load()
    H, Hkv, q_len, klen, D = 8, 2, 900, 333, 128
    G = H // Hkv
    g = torch.Generator(device=DEV).manual_seed(99)
    key = torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half()
    val = torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half()
    qq = torch.randn(1, q_len, H, D, generator=g, device=DEV).half().transpose(1, 2)
    prev_s = lib.kvz_debug_set_tunable(b"flash2_split", split)
    prev_b = lib.kvz_debug_set_tunable(b"flash2_min_blocks", 1)
    try:
        out, lse = ops.flash_fwd(qq, key, val, causal=True, return_lse=True)
    finally:
        lib.kvz_debug_set_tunable(b"flash2_split", prev_s)
        lib.kvz_debug_set_tunable(b"flash2_min_blocks", prev_b)
    s_ = torch.einsum("hid,hjd->hij", qq[0].float(), key[0].float().repeat_interleave(G, 0)) / math.sqrt(D)
    i = torch.arange(q_len, device=DEV).view(1, q_len, 1)
    j = torch.arange(klen, device=DEV).view(1, 1, klen)
    s_ = s_.masked_fill(j > i + (klen - q_len), float("-inf"))
    blind = q_len - klen   # positions 0 .. blind-1 see nothing
    want = torch.einsum("hij,hjd->ihd", torch.nan_to_num(torch.softmax(s_, -1), nan=0.0), val[0].float().repeat_interleave(G, 0))
    assert (out[0, :blind] == 0).all() and torch.isinf(lse[0, :, :blind]).all() and (lse[0, :, :blind] < 0).all()
    check_attn(f"flash2_rows_without_keys/split{split}", out[0], want.half(), 1e-3, ulp_of=torch.float16)
    assert (lse[0, :, blind:] - torch.logsumexp(s_, -1)[:, blind:]).abs().max() <= 1e-3


def test_flash_attn_varlen_func_call_compatibility():
    """ops.flash_attn_varlen_func takes the reference's call (attention/attn.py:61-71: q [Hkv*q_len, G, D], k/v [rows, 1, D],
    cu_seqlens_q/k, max lengths, causal=True) and returns what the oracle's restatement of flash-attn's semantics returns."""
    from kvzip_amd import ops
    g = torch.Generator().manual_seed(11)
    Hkv, G, D = 4, 7, 128
    for q_len, lens in ((1, [300, 17, 1024, 5]), (9, [64, 9, 700, 33])):
        cu_k = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
        cu_q = q_len * torch.arange(Hkv + 1, dtype=torch.int32)
        q = torch.randn(Hkv * q_len, G, D, generator=g).half()
        k = torch.randn(sum(lens), 1, D, generator=g).half()
        v = torch.randn(sum(lens), 1, D, generator=g).half()
        got = ops.flash_attn_varlen_func(q.to(DEV), k.to(DEV), v.to(DEV), cu_q.to(DEV), cu_k.to(DEV), q_len, max(lens),
                                         causal=True).cpu()
        want = orc.varlen_attn(q, k.view(-1, D), v.view(-1, D), cu_k[:-1].tolist(), lens, q_len)
        assert got.shape == q.shape and (got.float() - want.float()).abs().max() <= 1e-3


def test_tiny_api_cuda_alias_runs_the_reference_call():
    """The reference's import line (attention/kvcache.py:10) + call (kvcache.py:62-73) through the alias module."""
    from tiny_api_cuda import update_flatten_view
    g = torch.Generator().manual_seed(4)
    lens = torch.tensor([5, 0, 9], dtype=torch.int32)
    cu = torch.tensor([0, 5, 5, 14], dtype=torch.int32)
    cache = torch.randn(14, 64, generator=g).half()
    state = torch.randn(3 * 2, 64, generator=g).half()
    got = update_flatten_view(cache.to(DEV), state.to(DEV), lens.to(DEV), cu.to(DEV)).cpu()
    assert torch.equal(got, orc.update_flatten_view(cache, state, lens, cu))
    with pytest.raises(RuntimeError):
        update_flatten_view(cache.to(DEV), state.to(DEV), lens.long().to(DEV), cu.to(DEV))


def test_varlen_attn_compaction_identity():
    """compact + varlen attention == dense attention over the full KV with evicted keys masked (a13 anchor)."""
    Hkv, G, D, klen, q_len = 2, 7, 128, 777, 3
    g = torch.Generator().manual_seed(8)
    k = torch.randn(1, Hkv, klen, D, generator=g).half()
    v = torch.randn(1, Hkv, klen, D, generator=g).half()
    q = torch.randn(Hkv * q_len, G, D, generator=g).half()
    valid = (torch.rand(1, 1, Hkv, klen - q_len, generator=g) < 0.35)
    plan = ops().compact_plan(valid.to(DEV), 0, klen)
    tot = int(plan.len_k.sum())
    ko, vo = ops().compact_layer(k.to(DEV), v.to(DEV), plan, 0, tot)
    got = ops().varlen_attn(q.to(DEV), ko, vo, plan.seg_start[0], plan.len_k[0], q_len, int(plan.max_len_k[0])).cpu()
    full = orc.get_valid(valid[0], 0, klen)[0]
    want = orc.dense_masked_attn(q, k[0], v[0], full, q_len)
    assert (got.float() - want.float()).abs().max() <= 1e-3


# ------------------------------------------------------------------------------------------------
# a1 rounding chain: exhaustive over all 65536 16-bit inputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D", [64, 128])
def test_round_chain_exhaustive(dtype, D):
    """half(x / float32(sqrt(D))) for EVERY 16-bit x: the kernels' exact-reciprocal multiply and their IEEE-division
    fallback both reproduce torch's CPU result (tensor / python float, fp32 opmath) bit for bit."""
    import ctypes as C
    from kvzip_amd import _lib
    lib = _lib.load()
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dtype)
    want = (x / math.sqrt(D))
    finite = torch.isfinite(x.float())
    for force_div in (0, 1, 2, 3, 4):   # 2: the reciprocal through the four-logit chain of the two-pass kernels (quad_args); 3 / 4: the pruned call's (quad_round), low / high half
        out = torch.empty(65536, dtype=torch.int16, device=DEV)
        rcp = C.c_float(0)
        rc = lib.kvz_debug_round_chain(bits.to(DEV).data_ptr(), 65536, D, 0 if dtype == torch.float16 else 1, force_div,
                                       out.data_ptr(), C.byref(rcp), None)
        assert rc == 0
        torch.cuda.synchronize()
        got = out.cpu().view(dtype)
        # (the fused multiply path returns +0 for x = -0: equal as values, irrelevant for the softmax)
        same = (got.view(torch.int16) == want.view(torch.int16)) | ~finite | ((got.float() == 0) & (want.float() == 0))
        bad = torch.nonzero(~same).view(-1)[:8]
        assert same.all(), (dtype, D, force_div, int((~same).sum()), bad.tolist(), x[bad].tolist(),
                            got[bad].tolist(), want[bad].tolist())
        if force_div != 1:
            assert rcp.value != 0.0, "no exact reciprocal found: kernels would fall back to the slow division"


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("poison", ["inf", "nan"])
def test_score_chunk_propagates_nan_like_the_reference(poison, deferred):
    """An inf / NaN in one query row makes that row's softmax NaN in the reference, and amax over the rows carries it to EVERY ctx key
    of the KV head (attention/score.py:59-63; torch.amax propagates NaN).  Both output paths of the kernel - per-call finalize and
    the deferred log buffer with its unsigned-minimum merge - must do the same; the other heads stay finite and equal the oracle."""
    from kvzip_amd.kvcache import EvictCache
    H, Hkv, D, sink, N, q_len = 8, 2, 128, 16, 700, 333
    m = N
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, q_len, D, generator=g).half()
    k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g).half()
    q[0, 1, 200, 5] = float("inf") if poison == "inf" else float("nan")   # query head 1 -> KV head 0
    want = orc.get_score(q, k, sink, sink, sink + m)
    assert torch.isnan(want[0, 0].float()).all() and not torch.isnan(want[0, 1].float()).any()
    if deferred:
        cfg = types.SimpleNamespace(num_hidden_layers=1, num_attention_heads=H, num_key_value_heads=Hkv)
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=torch.float16, verbose=False)
        kv.update(k[:, :, :sink + N].to(DEV), k[:, :, :sink + N].to(DEV), 0)
        kv.init_score()
        assert kv.score_deferred
        k_all, _ = kv.update(k[:, :, sink + N:].to(DEV), k[:, :, sink + N:].to(DEV), 0)
        kv._get_score(q.to(DEV), k_all, 0)
        got = kv.score[0].cpu()
    else:
        got = ops().score_chunk(q.to(DEV), k.to(DEV), sink, sink, sink + m).cpu()
    assert torch.isnan(got[0, 0].float()).all(), "NaN of the poisoned head was dropped"
    assert not torch.isnan(got[0, 1].float()).any()
    d = ulp_diff(got[0, 1], want[0, 1])
    assert (d <= 1).float().mean() >= 0.99 and d.max() <= 16


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_varlen_attn_vs_an_independent_flash_implementation(dtype):
    """a13 has no reference-held vector (flash-attn 2.7.4.post1 is not in the image).  Besides the CPU restatement, the decode call is
    anchored on an INDEPENDENT implementation of the same published algorithm that does exist on the GPU box: PyTorch-ROCm's fused
    flash-attention backend of ``scaled_dot_product_attention`` (AOTriton), forced with ``sdpa_kernel(FLASH_ATTENTION)``.  Per KV head:
    q_len = 1 (the mask is irrelevant: every key is visible) and a 5-token query with the bottom-right aligned causal mask expressed
    as "the last 5 rows of a square causal problem" (is_causal over the last 5 keys ++ full visibility of the prefix is what
    flash-attn's varlen kernel computes).  Both results are 16-bit roundings of the same fp32 value: at most one output step apart."""
    import torch.nn.functional as F
    from torch.nn.attention import SDPBackend, sdpa_kernel
    Hkv, G, D = 4, 7, 128
    lens = [4097, 257, 1500, 33]
    g = torch.Generator(device=DEV).manual_seed(11)
    starts, tot = [], 0
    for ln in lens:
        starts.append(tot)
        tot += ln + 7
    k = torch.randn(tot, D, generator=g, device=DEV).to(dtype)
    v = torch.randn(tot, D, generator=g, device=DEV).to(dtype)
    ks = torch.tensor(starts, dtype=torch.int32, device=DEV)
    kl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    tol = 1e-3 if dtype == torch.float16 else 8e-3
    for q_len in (1, 5):
        q = torch.randn(Hkv * q_len, G, D, generator=g, device=DEV).to(dtype)
        got = ops().varlen_attn(q, k, v, ks, kl, q_len, max(lens)).view(Hkv, q_len, G, D)
        for h in range(Hkv):
            kh = k[starts[h]:starts[h] + lens[h]].view(1, 1, lens[h], D).expand(1, G, -1, -1)
            vh = v[starts[h]:starts[h] + lens[h]].view(1, 1, lens[h], D).expand(1, G, -1, -1)
            qh = q.view(Hkv, q_len, G, D)[h].permute(1, 0, 2).unsqueeze(0)          # [1, G, q_len, D]
            with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
                if q_len == 1:
                    want = F.scaled_dot_product_attention(qh, kh, vh)
                else:
                    # bottom-right alignment through a square causal problem: pad the query with lens-q_len leading rows (their
                    # outputs are discarded); row i of the padded problem sees keys 0..i, i.e. the real row j sees keys 0..j+lens-q_len
                    pad = torch.zeros(1, G, lens[h] - q_len, D, dtype=dtype, device=DEV)
                    want = F.scaled_dot_product_attention(torch.cat([pad, qh], dim=2), kh, vh, is_causal=True)[:, :, -q_len:]
            check_attn(f"vs_torch_flash/{dtype}/q{q_len}/h{h}", got[h].permute(1, 0, 2), want[0], tol, ulp_of=dtype)
