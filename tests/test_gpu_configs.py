"""GPU: the BASELINE.json configurations as parity cases.

C1  Qwen2.5-0.5B geometry, 2k context, ratio 0.3 — the reference's CPU-runnable case, run here IN FULL against the oracle.
C3  Llama-3.1-8B geometry (GQA, 8 KV heads, G=4): scoring vs oracle at a reduced context + full 128k-size properties.
C5  Qwen2.5-14B geometry, --level head, ratio 0.6 (known-answer head scores of the reference) + post-prune decode.
"""
import types

import numpy as np
import pytest
import torch

import kvzip_oracle as orc
from conftest import check_score_parity, from_bits, load_golden, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(L, H, Hkv):
    return types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)


def _chunks(sink, N, chunk=2000):
    out = []
    for c, st in enumerate(range(0, N, chunk)):
        m = min(chunk, N - st)
        out.append((sink + st, sink + st + m, m + (13 if c == 0 else 26)))
    return out


def test_config_c1_qwen05b_full_context():
    """Whole C1: 24 layers x 2 scoring chunks (2000 + 48 tokens) scored on the GPU and on the CPU oracle from the same
    inputs, then prune(0.3).  Scores within the rounding-chain tolerance; masks bit-exact given the oracle's scores;
    end-to-end mask Hamming distance reported."""
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N = 24, 14, 2, 64, 30, 2048
    dt = torch.float16
    g = torch.Generator().manual_seed(42)
    kv = EvictCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    K = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dt) for _ in range(L)]
    V = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dt) for _ in range(L)]
    for l in range(L):
        kv.update(K[l].to(DEV), V[l].to(DEV), l)
    kv.init_score()
    ref = [[] for _ in range(L)]
    for st, en, q_len in _chunks(sink, N):
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            q = torch.randn(1, H, q_len, D, generator=g).to(dt)
            kr = torch.randn(1, Hkv, q_len, D, generator=g).to(dt)
            k_all, _ = kv.update(kr.to(DEV), kr.to(DEV), l)
            kv._get_score(q.to(DEV), k_all, l)
            ref[l].append(orc.get_score(q, torch.cat([K[l], kr], dim=2), sink, st, en))
        kv.slice(seen)
    kv.start_idx, kv.get_score = sink, False
    ref = [torch.cat(r, dim=-1) for r in ref]
    d = torch.stack([ulp_diff(kv.score[l], ref[l]) for l in range(L)])
    exact, within1 = float((d == 0).float().mean()), float((d <= 1).float().mean())
    print(f"C1 scores: {exact:.4f} bit-identical, {within1:.5f} within 1 half-ulp, worst {int(d.max())} ulp")
    check_score_parity("c1_full", torch.stack([kv.score[l] for l in range(L)]), torch.stack(ref))
    assert exact >= 0.998 and within1 >= 0.9998 and d.max() <= 10  # measured 0.9992 / 0.99993 / 5 (profiles/r2_parity_headline.txt)
    # masks: identical scores -> identical masks (bit-exact), HIP scores -> Hamming distance
    v_ref, t_ref = orc.threshold(ref, 0.3)
    hip_scores = [s.clone() for s in kv.score]
    kv.score = [r.to(DEV) for r in ref]
    thres, r_real = kv.prune(0.3)
    assert thres == t_ref and torch.equal(kv.valid.cpu(), v_ref)
    assert abs(r_real - 0.3) < 2e-3
    v_hip, t_hip = orc.threshold([s.cpu() for s in hip_scores], 0.3)
    assert t_hip == t_ref
    from conftest import check_mask_flips
    check_mask_flips("c1_full", torch.stack([s.cpu() for s in hip_scores]), torch.stack(ref), v_hip, v_ref, t_ref, allowed=0)  # measured 0 of 98 304
    # compaction == oracle prepare_init on the same mask
    fk, fv, lens, cus, mxs = orc.prepare_init(K, V, v_ref, sink)
    for l in range(L):
        seg = kv.info["seg_start"][l].cpu()
        for h in range(Hkv):
            n, c = int(lens[l][h]), int(cus[l][h])
            assert torch.equal(kv.key_cache[l][int(seg[h]):int(seg[h]) + n].cpu(), fk[l][c:c + n])
            assert torch.equal(kv.value_cache[l][int(seg[h]):int(seg[h]) + n].cpu(), fv[l][c:c + n])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_config_c3_llama_geometry_scoring(dtype):
    """Llama-3.1-8B head geometry (32 query heads, 8 KV heads, D=128): one (layer, chunk) at full chunk size."""
    from kvzip_amd import ops
    H, Hkv, D, sink, N, m = 32, 8, 128, 32, 3000, 2000
    q_len = m + 26
    g = torch.Generator().manual_seed(3)
    q = torch.randn(1, H, q_len, D, generator=g).to(dtype)
    k = torch.randn(1, Hkv, sink + N + q_len, D, generator=g).to(dtype)
    start = sink + 700
    want = orc.get_score(q, k, sink, start, start + m)
    got = ops.score_chunk(q.to(DEV), k.to(DEV), sink, start, start + m).cpu()
    # (G = 4: the 16 384 rows of a KV head are cut into row tiles differently than at G = 7; same arithmetic, same distribution -
    # the worst case of 7-8 steps seen here and in bench.py's C3 parity_sample is ONE score whose winning logit flipped by an ulp)
    check_score_parity(f"c3_llama/{dtype}", got, want)


def test_config_c3_full_size_compaction_properties():
    """128k-token context, 8 KV heads: order-preserving gather checked through size-independent properties."""
    from kvzip_amd import ops
    Hkv, D, sink, N = 8, 128, 32, 131072
    klen = sink + N
    g = torch.Generator(device=DEV).manual_seed(9)
    k = torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half()
    v = torch.randn(1, Hkv, klen, D, generator=g, device=DEV).half()
    # tag every row with its own index so that order and identity can be verified after the move
    tag = torch.arange(klen, device=DEV, dtype=torch.int32)
    k.view(torch.int32)[0, :, :, 0] = tag.unsqueeze(0)  # first two halfs of each row = row index bits
    valid = torch.rand(1, 1, Hkv, N, generator=g, device=DEV) < 0.3
    plan = ops.compact_plan(valid, sink, klen, slack=64)
    len_k = plan.len_k[0]
    full = torch.cat([torch.ones(Hkv, sink, dtype=torch.bool, device=DEV), valid[0, 0]], dim=1)
    assert torch.equal(len_k.long(), full.sum(-1))
    total = int(len_k.sum()) + 64 * Hkv
    ko, vo = ops.compact_layer(k, v, plan, 0, total)
    seg = plan.seg_start[0].tolist()
    for h in range(Hkv):
        n = int(len_k[h])
        rows = ko[seg[h]:seg[h] + n]
        idx = rows.view(torch.int32)[:, 0].long()
        assert torch.equal(idx, torch.nonzero(full[h]).squeeze(-1))          # exactly the kept rows, in order
        assert torch.equal(rows.view(torch.int16), k[0, h].index_select(0, idx).view(torch.int16))  # bytes moved intact
        assert torch.equal(vo[seg[h]:seg[h] + n], v[0, h].index_select(0, idx))
    # idempotence: compacting the compacted cache with an all-ones mask is the identity
    ones = torch.ones(1, 1, 1, int(len_k[0]) - 0, dtype=torch.bool, device=DEV)
    plan2 = ops.compact_plan(ones, 0, int(len_k[0]))
    k2, _ = ops.compact_layer(ko[seg[0]:seg[0] + int(len_k[0])].view(1, 1, -1, D),
                              vo[seg[0]:seg[0] + int(len_k[0])].view(1, 1, -1, D), plan2, 0, int(len_k[0]))
    assert torch.equal(k2.view(torch.int16), ko[seg[0]:seg[0] + int(len_k[0])].view(torch.int16))


def test_config_c5_qwen14b_head_level_and_decode():
    """Qwen2.5-14B geometry (L48, 40 query heads, 8 KV heads): context-independent head-level eviction at ratio 0.6
    from the reference's own head-score file, then two decode steps checked against the oracle attention."""
    from kvzip_amd.kvcache import EvictCache
    g = load_golden("g4_head_score.npz")
    hs = from_bits(g["qwen2.5-14b/head_score"], True)  # [48, 8] bf16
    L, Hkv = hs.shape
    H, D, sink, N = 40, 128, 16, 4096
    G = H // Hkv
    dt = torch.bfloat16
    gen = torch.Generator(device=DEV).manual_seed(14)
    kv = EvictCache(_cfg(L, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    K0 = []
    for l in range(L):
        k = torch.randn(1, Hkv, sink + N, D, generator=gen, device=DEV).to(dt)
        v = torch.randn(1, Hkv, sink + N, D, generator=gen, device=DEV).to(dt)
        kv.update(k, v, l)
        K0.append((k, v))
    kv.score = hs.to(DEV).unsqueeze(-1).expand(-1, -1, N).unsqueeze(1)
    thres, r_real = kv.prune(0.6, "head")
    assert thres == 0.53515625
    kept = g["qwen2.5-14b/kept/1000/0.6"]
    assert int(kept.sum()) == 229
    assert abs(r_real - 229 / 384) < 1e-6
    for l in range(L):
        assert np.array_equal(kv.info["len_k"][l].cpu().numpy(), sink + kept[l].astype(np.int32) * N)
        seg = kv.info["seg_start"][l].tolist()
        for h in range(Hkv):
            n = sink + (N if kept[l, h] else 0)
            assert torch.equal(kv.key_cache[l][seg[h]:seg[h] + n], K0[l][0][0, h, :n])
    # decode two tokens on a few layers, compare with the oracle varlen attention on the same flattened tensors
    for step in range(2):
        for l in (0, 17, 47):
            q = torch.randn(1, H, 1, D, generator=gen, device=DEV).to(dt)
            kn = torch.randn(1, Hkv, 1, D, generator=gen, device=DEV).to(dt)
            vn = torch.randn(1, Hkv, 1, D, generator=gen, device=DEV).to(dt)
            kf, vf = kv.update(kn, vn, l)
            qf, kf2, vf2, info = kv.prepare(q, kf, vf, l)
            got = kv.attend(qf, kf2, vf2, info).cpu().float()
            lens = (info["k_len"].cpu() + info["k_len_offset"]).tolist()
            want = orc.varlen_attn(qf.cpu(), kf2.view(-1, D).cpu(), vf2.view(-1, D).cpu(), info["k_start"].tolist(),
                                   lens, 1).float()
            assert (got - want).abs().max() <= 8e-3  # bf16 output
    assert kv._seen_tokens == sink + N + 2
