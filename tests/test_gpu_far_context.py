"""GPU: the far end of long contexts (round 6).

G14  the reference's scores of the LAST TWO scoring chunks of one layer of a 131 072-token context (chunk starts 128 032 / 130 032, key
     length 133 k, the short last chunk of 1 072 tokens), Qwen2.5-7B head geometry, both dtypes (oracle/gen_golden.py:gen_far_context).
G15  ONE FULL LAYER of that context from the reference: all 66 chunks -> 524 288 scores, threshold and mask at ratio 0.3
     (gen_full_layer) - the measured count of non-identical scores and flipped mask entries at the headline size, per layer.
N = 524 288 (the reference's demo scale, README.md:21 / demo.py:31-50): a scoring call at the far end against the oracle, selection,
     compaction and decode attention through size-independent properties and the oracle on one layer.
"""
import types

import numpy as np
import pytest
import torch

import e2e_inputs as E
import kvzip_oracle as orc
from conftest import check_mask_flips, check_score_parity, from_bits, load_golden, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(L, H, Hkv):
    return types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_far_context_scores_match_the_reference(tag):
    """G14 through the cache object (update -> _get_score -> slice, as model/wrapper.py:223-249 drives it) and through the plain entry point."""
    from kvzip_amd import ops
    from kvzip_amd.kvcache import EvictCache
    g = load_golden("g14_far_context.npz")
    geom = E.GEOM_FAR
    assert [geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")] == g["geom"].tolist()
    dt = torch.float16 if tag == "f16" else torch.bfloat16
    K0, far = E.make_far(dt)
    assert E.checksum([K0], [[(q, kr)] for (_, _, _, q, kr) in far]) == int(g[f"{tag}/checksum"][0]), "seeded inputs differ from the fixture's"
    want = from_bits(g[f"{tag}/score"], tag == "bf16")            # [1, Hkv, 3072]
    sink, N, H, Hkv = geom["sink"], geom["N"], geom["H"], geom["Hkv"]
    kv = EvictCache(_cfg(1, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    K0d = K0.to(DEV)
    kv.update(K0d, K0d, 0)
    kv.init_score()
    direct = []
    for (st, en, q_len, q, kr) in far:
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        k_all, _ = kv.update(kr.to(DEV), kr.to(DEV), 0)
        assert k_all.shape[2] == sink + N + q_len
        kv._get_score(q.to(DEV), k_all, 0)
        direct.append(ops.score_chunk(q.to(DEV), k_all, sink, st, en).cpu())
        kv.slice(seen)
    got = kv.score[0].cpu()
    assert got.shape == want.shape
    check_score_parity(f"g14 far context/{tag}/cache object", got, want)
    check_score_parity(f"g14 far context/{tag}/kvz_score_chunk", torch.cat(direct, dim=-1), want)


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_full_layer_of_the_headline_context_against_the_reference(tag):
    """G15: every scoring chunk of one layer of the 131 072-token context; scores bounded like everywhere else, threshold EQUAL to the
    reference's, flipped mask entries counted (and each of them a non-identical score at the threshold)."""
    from kvzip_amd.kvcache import EvictCache
    g = load_golden("g15_full_layer.npz")
    geom = E.GEOM_FAR
    dt = torch.float16 if tag == "f16" else torch.bfloat16
    sink, N, H, Hkv = geom["sink"], geom["N"], geom["H"], geom["Hkv"]
    it = E.stream_full(dt)
    K0 = next(it)
    acc = E.checksum_update(0, K0)
    kv = EvictCache(_cfg(1, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    K0d = K0.to(DEV)
    kv.update(K0d, K0d, 0)
    kv.init_score()
    for (st, en, q_len, q, kr) in it:
        acc = E.checksum_update(E.checksum_update(acc, q), kr)
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        k_all, _ = kv.update(kr.to(DEV), kr.to(DEV), 0)
        kv._get_score(q.to(DEV), k_all, 0)
        kv.slice(seen)
    assert acc == int(g[f"{tag}/checksum"][0]), "seeded inputs differ from the fixture's"
    kv.start_idx, kv.get_score = sink, False
    want = from_bits(g[f"{tag}/score"], tag == "bf16").view(1, 1, Hkv, N)
    got = torch.stack([s for s in kv.score], 0).cpu()
    d = ulp_diff(got, want)
    check_score_parity(f"g15 full layer/{tag}", got, want)
    want_valid = torch.from_numpy(np.unpackbits(g[f"{tag}/valid"])[:want.numel()]).bool().view(want.shape)
    want_thres = float(g[f"{tag}/thres"][0])
    thres, r_real = kv.prune(0.3)
    ham = int((kv.valid.cpu() != want_valid).sum())
    print(f"\nG15 {tag}: {want.numel()} scores of one layer x 66 chunks: {int((d != 0).sum())} not bit-identical, {int((d > 1).sum())} beyond one "
          f"step, worst {int(d.max())}; thres {thres!r} vs reference {want_thres!r}; mask Hamming distance {ham}; kept ratio {r_real:.5f}")
    assert thres == want_thres
    # (one layer's scores share the threshold with nobody else: the count of entries AT the threshold value decides how many can flip)
    check_mask_flips(f"g15 full layer/{tag}", got, want, kv.valid.cpu(), want_valid, want_thres, allowed=3 if tag == "f16" else 2)
    assert np.array_equal(kv.valid.cpu().sum(-1).reshape(Hkv).numpy().astype(np.int32) - g[f"{tag}/kept"],
                          (kv.valid.cpu().int() - want_valid.int()).sum(-1).reshape(Hkv).numpy())


def test_half_million_token_context():
    """N = 524 288 at the Qwen2.5-7B head geometry, one layer (0.54 GB per K / V buffer; byte offsets beyond 2^29 inside a head, beyond 2^31
    across the buffer when V follows K): the last scoring chunk against the oracle, the global-threshold selection against torch's own
    sort, order / identity / idempotence of the compaction, a decode step on the pruned layer against the oracle."""
    from kvzip_amd import ops
    from kvzip_amd.kvcache import EvictCache
    H, Hkv, D, sink, N, chunk = 28, 4, 128, 32, 524288, 2000
    G = H // Hkv
    dt = torch.float16
    gen = torch.Generator(device=DEV).manual_seed(524288)
    K = torch.randn(1, Hkv, sink + N, D, generator=gen, device=DEV).to(dt)
    V = torch.randn(1, Hkv, sink + N, D, generator=gen, device=DEV).to(dt)
    tag = torch.arange(sink + N, device=DEV, dtype=torch.int32)
    V.view(torch.int32)[0, :, :, 0] = tag.unsqueeze(0)            # every V row carries its own index: order and identity after the move
    kv = EvictCache(_cfg(1, H, Hkv), (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    kv.update(K, V, 0)
    # ---- the last two chunks through the cache object, against the oracle on the same rows (sink ++ chunk ++ repeat: the oracle's
    # arithmetic only sees those keys, so it gets a compact key tensor instead of 0.5 GB)
    chunks = E.chunks(dict(sink=sink, N=N, chunk=chunk))
    kv.init_score()
    got, want = [], []
    for (st, en, q_len) in chunks[-2:]:
        q = torch.randn(1, H, q_len, D, generator=gen, device=DEV).to(dt)
        kr = torch.randn(1, Hkv, q_len, D, generator=gen, device=DEV).to(dt)
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        k_all, _ = kv.update(kr, kr, 0)
        kv._get_score(q, k_all, 0)
        kv.slice(seen)
        compact = torch.cat([K[:, :, :sink], K[:, :, st:en], kr], dim=2).cpu()
        want.append(orc.get_score(q.cpu(), compact, sink, sink, sink + (en - st)))
    got = kv.score[0].cpu()
    check_score_parity("n524288/last two chunks", got, torch.cat(want, dim=-1))
    kv.start_idx, kv.get_score = sink, False
    # ---- selection at full size on synthetic scores (softmax-max-like: many equal values), against torch's sort
    score = (torch.rand(1, 1, Hkv, N, generator=gen, device=DEV) ** 6).to(dt)
    n_all = score.numel()
    srt = torch.sort(score.reshape(-1).float(), descending=True).values
    want_thres = float(srt[max(int(n_all * 0.3) - 1, 0)])
    kv.score = [score[0]]
    thres, r_real = kv.prune(0.3)
    assert thres == want_thres
    want_valid = score.float() > want_thres
    assert torch.equal(kv.valid, want_valid)
    assert abs(r_real - float(want_valid.float().mean())) < 1e-9
    # ---- compaction: exactly the kept rows, in order, bytes intact (V rows are tagged with their index)
    full = torch.cat([torch.ones(Hkv, sink, dtype=torch.bool, device=DEV), want_valid[0, 0]], dim=1)
    len_k = kv.info["len_k"][0]
    seg = kv.info["seg_start"][0].tolist()
    assert torch.equal(len_k.long(), full.sum(-1))
    for h in range(Hkv):
        n = int(len_k[h])
        rows_v = kv.value_cache[0][seg[h]:seg[h] + n]
        idx = rows_v.view(torch.int32)[:, 0].long()
        assert torch.equal(idx, torch.nonzero(full[h]).squeeze(-1))
        assert torch.equal(rows_v.view(torch.int16), V[0, h].index_select(0, idx).view(torch.int16))
        assert torch.equal(kv.key_cache[0][seg[h]:seg[h] + n].view(torch.int16), K[0, h].index_select(0, idx).view(torch.int16))
    # ---- one decode step on the pruned layer (~157 k kept rows per head) against the oracle
    q1 = torch.randn(1, H, 1, D, generator=gen, device=DEV).to(dt)
    kn = torch.randn(1, Hkv, 1, D, generator=gen, device=DEV).to(dt)
    vn = torch.randn(1, Hkv, 1, D, generator=gen, device=DEV).to(dt)
    kf, vf = kv.update(kn, vn, 0)
    qf, kf2, vf2, info = kv.prepare(q1, kf, vf, 0)
    out = kv.attend(qf, kf2, vf2, info).cpu().float()
    lens = (info["k_len"].cpu() + info["k_len_offset"]).tolist()
    ref = orc.varlen_attn(qf.cpu(), kf2.view(-1, D).cpu(), vf2.view(-1, D).cpu(), info["k_start"].tolist(), lens, 1).float()
    # (the first two halfs of every V row hold its index bits - arbitrary 16-bit patterns, inf / NaN among them: output columns 0 and 1
    # are not numbers; the V columns are independent, so every other column is checked)
    assert (out[..., 2:] - ref[..., 2:]).abs().max() <= 1e-3
