"""CPU: the oracle restatement is pinned bit-for-bit to golden vectors produced by the reference's own Python
(oracle/gen_golden.py).  No GPU, no /root/reference access."""
import numpy as np
import pytest
import torch

import kvzip_oracle as orc
from conftest import from_bits, load_golden, to_bits, ulp_diff


def _cases(npz):
    return sorted({k.split("/")[0] for k in npz.files})


G1 = load_golden("g1_score.npz")


@pytest.mark.parametrize("name", _cases(G1))
def test_get_score_bit_exact(name):
    Hkv, G, D, sink, start, end, q_len, klen, bf = G1[name + "/meta"].tolist()
    q = from_bits(G1[name + "/q"], bf)
    k = from_bits(G1[name + "/k"], bf)
    want = from_bits(G1[name + "/score"], bf)
    got = orc.get_score(q, k, sink, start, end)
    assert got.shape == want.shape == (1, Hkv, end - start)
    assert np.array_equal(to_bits(got), to_bits(want))


@pytest.mark.parametrize("name", _cases(G1))
def test_get_score_fp32_chain_model(name):
    """The explicit fp32 rounding chain (what the HIP kernel implements) reproduces the reference up to rare
    1-ulp logit flips: >= 99% of scores bit-identical, everything within 8 half-ulps."""
    Hkv, G, D, sink, start, end, q_len, klen, bf = G1[name + "/meta"].tolist()
    q = from_bits(G1[name + "/q"], bf)
    k = from_bits(G1[name + "/k"], bf)
    want = from_bits(G1[name + "/score"], bf)
    got = orc.get_score_chain_fp32(q, k, sink, start, end)
    d = ulp_diff(got, want)
    assert (d == 0).float().mean() >= 0.99, (d == 0).float().mean()
    assert d.max() <= 8


def test_threshold_golden():
    g = load_golden("g2_threshold.npz")
    for tag, bf in (("bf16", True), ("f16", False), ("odd", False)):
        score = from_bits(g[f"{tag}/score"], bf)
        ratios = [k.split("/")[-1] for k in g.files if k.startswith(f"{tag}/thres/")]
        assert ratios
        for r in ratios:
            valid, thres = orc.threshold(score, float(r))
            want = np.unpackbits(g[f"{tag}/valid/{r}"])[:score.numel()].astype(bool)
            assert np.array_equal(valid.numpy().reshape(-1), want), (tag, r)
            assert thres == g[f"{tag}/thres/{r}"][0], (tag, r)


def test_threshold_edge_semantics():
    s = torch.tensor([[0.5, 0.25, 0.25, 0.125]], dtype=torch.float16)
    v, t = orc.threshold(s, 0.0)          # n = 0 -> thres = max -> nothing is > max
    assert not v.any() and t == 0.5
    v, t = orc.threshold(s, 0.75)         # n = 2 -> thres = 0.25 -> both ties evicted (strict >)
    assert v.tolist() == [[True, False, False, False]] and t == 0.25
    v, t = orc.threshold(s, 1.0)
    assert v.all() and t == 0.0


def test_threshold_uniform_golden():
    g = load_golden("g3_threshold_uniform.npz")
    score = from_bits(g["score"], False)
    for key in [k for k in g.files if k.startswith("valid/")]:
        r = float(key.split("/")[1])
        valid, thres = orc.threshold_uniform([score[i] for i in range(score.shape[0])], r)
        want = np.unpackbits(g[key])[:score.numel()].astype(bool)
        assert thres == 0
        assert np.array_equal(valid.numpy().reshape(-1), want), r


def test_head_score_known_answers():
    """utils/head_score/*.pt known-answer data: kept-head sets and thresholds."""
    g = load_golden("g4_head_score.npz")
    expect_06 = {"qwen2.5-14b": (229, 0.53515625), "qwen2.5-7b": (67, 0.890625), "llama3.1-8b": (153, 0.399169921875)}
    for name, (kept, thres) in expect_06.items():
        hs = from_bits(g[f"{name}/head_score"], bool(g[f"{name}/is_bf16"][0]))
        for ctx_len in (64, 1000):
            for r in (0.3, 0.6, 0.9):
                score = hs.unsqueeze(-1).expand(-1, -1, ctx_len).unsqueeze(1)
                valid, t = orc.threshold(score, r)
                want = g[f"{name}/kept/{ctx_len}/{r!r}"]
                assert np.array_equal(valid[:, 0, :, 0].numpy(), want)
                assert t == g[f"{name}/thres/{ctx_len}/{r!r}"][0]
        assert int(g[f"{name}/kept/1000/0.6"].sum()) == kept
        assert g[f"{name}/thres/1000/0.6"][0] == thres


def test_threshold_heads_equals_expanded_threshold():
    """oracle.threshold_heads (no expansion: rank idx // N on the [L,Hkv] values) is pinned by the reference's results on the
    expanded tensor (g4) and agrees with oracle.threshold on the expanded tensor for other N / ratios, ties included."""
    g = load_golden("g4_head_score.npz")
    for name in ("qwen2.5-14b", "qwen2.5-7b", "llama3.1-8b"):
        hs = from_bits(g[f"{name}/head_score"], bool(g[f"{name}/is_bf16"][0]))
        for ctx_len in (64, 1000):
            for r in (0.3, 0.6, 0.9):
                kept, t = orc.threshold_heads(hs, ctx_len, r)
                assert np.array_equal(kept.numpy(), g[f"{name}/kept/{ctx_len}/{r!r}"])
                assert t == g[f"{name}/thres/{ctx_len}/{r!r}"][0]
        for ctx_len in (1, 7, 333):
            for r in (0.0, 1e-9, 0.05, 0.5, 0.999, 1.0, 1.3):
                kept, t = orc.threshold_heads(hs, ctx_len, r)
                valid, t2 = orc.threshold(hs.unsqueeze(-1).expand(-1, -1, ctx_len).unsqueeze(1), r)
                assert torch.equal(valid[:, 0, :, 0], kept) and t == t2, (name, ctx_len, r)


@pytest.mark.parametrize("tag", ["f16_pair", "bf16_pair", "f16_uniform"])
def test_cache_life_cycle_golden(tag):
    """score -> prune -> prepare_init -> append/prepare x3 -> slice, against the reference's EvictCache run."""
    g = load_golden(f"g6_cache_{tag}.npz")
    L, H, Hkv, D, sink, N, bf = g["meta"].tolist()
    K = [from_bits(g[f"K0/{l}"], bf) for l in range(L)]
    V = [from_bits(g[f"V0/{l}"], bf) for l in range(L)]
    # scoring (two chunks)
    chunks = [(sink, sink + 50), (sink + 50, sink + 80)]
    scores = [[] for _ in range(L)]
    for ci, (st, en) in enumerate(chunks):
        for l in range(L):
            q = from_bits(g[f"sc/{ci}/{l}/q"], bf)
            kr = from_bits(g[f"sc/{ci}/{l}/k"], bf)
            kfull = torch.cat([K[l], kr], dim=2)
            scores[l].append(orc.get_score(q, kfull, sink, st, en))
    scores = [torch.cat(s, dim=-1) for s in scores]
    for l in range(L):
        assert np.array_equal(to_bits(scores[l]), g[f"score/{l}"])
    ratio = float(g["ratio"][0])
    if "uniform" in tag:
        valid, thres = orc.threshold_uniform(scores, ratio)
    else:
        valid, thres = orc.threshold(scores, ratio)
    assert np.array_equal(valid.numpy(), g["valid"])
    assert thres == g["thres"][0]
    r_real = 1 - (valid == False).float().mean().item()  # noqa: E712  (reference kvcache.py:133-134)
    assert r_real == g["r_real"][0]
    fk, fv, lens, cus, mxs = orc.prepare_init(K, V, valid, sink)
    for l in range(L):
        assert np.array_equal(to_bits(fk[l]), g[f"flatK/{l}"]) and np.array_equal(to_bits(fv[l]), g[f"flatV/{l}"])
        assert np.array_equal(lens[l].numpy(), g[f"len_k/{l}"])
        assert np.array_equal(cus[l].numpy(), g[f"cu_len_k/{l}"])
        assert int(mxs[l]) == int(g[f"max_len_k/{l}"][0])
    # generation: t = 7, 1, 1
    offset = [0] * L
    cu = [c.clone() for c in cus]
    cu_head = torch.arange(Hkv + 1, dtype=torch.int32)
    for si, t in enumerate((7, 1, 1)):
        for l in range(L):
            q = from_bits(g[f"gen/{si}/{l}/q"], bf)
            kn = from_bits(g[f"gen/{si}/{l}/k"], bf)
            vn = from_bits(g[f"gen/{si}/{l}/v"], bf)
            head_lens = lens[l] + offset[l]
            fk[l] = orc.update_flatten_view(fk[l], kn.contiguous().view(-1, D), head_lens, cu[l])
            fv[l] = orc.update_flatten_view(fv[l], vn.contiguous().view(-1, D), head_lens, cu[l])
            qo = orc.prepare_query(q, Hkv)
            offset[l] += t
            cu[l] = cu[l] + t * cu_head
            assert np.array_equal(to_bits(qo), g[f"gen/{si}/{l}/q_out"])
            assert np.array_equal(to_bits(fk[l]), g[f"gen/{si}/{l}/k_out"])
            assert np.array_equal(to_bits(fv[l]), g[f"gen/{si}/{l}/v_out"])
            assert np.array_equal(cu[l].numpy(), g[f"gen/{si}/{l}/cu_len_k"])
            assert np.array_equal((t * cu_head).numpy(), g[f"gen/{si}/{l}/cu_len_q"])
            assert [t, int(mxs[l]) + offset[l]] == g[f"gen/{si}/{l}/max_len"].tolist()
            att = orc.varlen_attn(qo, fk[l], fv[l], cu[l][:-1].tolist(), (cu[l][1:] - cu[l][:-1]).tolist(), t)
            assert np.array_equal(to_bits(att), g[f"gen/{si}/{l}/attn"])
    # slice back to the compressed context
    for l in range(L):
        sl = orc.slice_flat(fk[l], cu[l], lens[l])
        assert np.array_equal(to_bits(sl), g[f"sliced/K/{l}"])
        assert np.array_equal((cu[l] - offset[l] * cu_head).numpy(), g[f"sliced/cu_len_k/{l}"])
        assert np.array_equal(to_bits(sl), g[f"flatK/{l}"])


def test_varlen_attn_matches_dense_masked_identity():
    """a13 anchor: compacted varlen attention == dense attention over the full KV with evicted keys masked."""
    torch.manual_seed(3)
    Hkv, G, D, klen, q_len = 2, 4, 64, 90, 5
    k = torch.randn(Hkv, klen, D).half()
    v = torch.randn(Hkv, klen, D).half()
    q = torch.randn(Hkv * q_len, G, D).half()
    keep = torch.rand(Hkv, klen) < 0.4
    keep[:, -q_len:] = True
    flat_k = torch.cat([k[h][keep[h]] for h in range(Hkv)])
    flat_v = torch.cat([v[h][keep[h]] for h in range(Hkv)])
    lens = keep.sum(-1).tolist()
    starts = [0, lens[0]]
    a = orc.varlen_attn(q, flat_k, flat_v, starts, lens, q_len, causal=True)
    b = orc.dense_masked_attn(q, k, v, keep, q_len)
    assert torch.allclose(a.float(), b.float(), atol=1e-3, rtol=0)


def test_varlen_attn_matches_torch_sdpa():
    """a13, second anchor: an INDEPENDENT implementation of the published semantics of flash_attn_varlen_func(causal=True)
    (scaled dot-product attention per packed sequence, causal mask aligned to the bottom-right corner when seqlen_q <
    seqlen_k).  flash-attn itself is a third-party dependency that is absent from /root/reference (parity unpinned at that
    boundary); torch's math-backend SDPA with an explicit mask is what is available here."""
    import torch.nn.functional as F
    torch.manual_seed(11)
    Hkv, G, D = 3, 7, 128
    for q_len in (1, 4):
        lens = [37, 5 + q_len, 64]
        starts = [0, 40, 40 + 5 + q_len + 3]          # segments with gaps between them, as in the slack layout
        total = starts[-1] + lens[-1]
        k = torch.randn(total, D).half()
        v = torch.randn(total, D).half()
        q = torch.randn(Hkv * q_len, G, D).half()      # [Hkv*q_len, G, D] as handed over by prepare()
        got = orc.varlen_attn(q, k, v, starts, lens, q_len, causal=True).float()
        for h in range(Hkv):
            kh = k[starts[h]:starts[h] + lens[h]].float()
            vh = v[starts[h]:starts[h] + lens[h]].float()
            qh = q[h * q_len:(h + 1) * q_len].float().transpose(0, 1)          # [G, q_len, D]
            i = torch.arange(q_len).view(-1, 1)
            j = torch.arange(lens[h]).view(1, -1)
            mask = j <= i + (lens[h] - q_len)                                   # bottom-right aligned causal mask
            want = F.scaled_dot_product_attention(qh.unsqueeze(0), kh.expand(1, G, -1, -1), vh.expand(1, G, -1, -1),
                                                  attn_mask=mask).squeeze(0).transpose(0, 1)   # [q_len, G, D]
            assert torch.allclose(got[h * q_len:(h + 1) * q_len], want, atol=2e-3, rtol=0)


def test_oracle_pinned_on_e2e_d128_fixture_sample():
    """G9 (Qwen2.5-7B head geometry, reference-generated): the oracle restatement reproduces the reference's scores of one
    (layer, chunk) call bit for bit, and its threshold / mask on the reference's own scores (the whole fixture is checked on the
    GPU box by tests/test_gpu_e2e_parity.py; one call keeps the CPU suite short)."""
    import e2e_inputs as E
    g = load_golden("g9_e2e_d128.npz")
    geom = E.GEOM
    K0, per_chunk = E.make(torch.float16)
    assert E.checksum(K0, per_chunk) == int(g["f16/checksum"][0])
    want = from_bits(g["f16/score"], False)
    ci, l = 3, 1
    st, en, q_len = E.chunks()[ci]
    q, kr = per_chunk[ci][l]
    # a quarter of the query heads' worth of work would change the result (max over ALL rows): run the call in full
    got = orc.get_score(q, torch.cat([K0[l], kr], dim=2), geom["sink"], st, en)
    assert torch.equal(to_bits_t(got), to_bits_t(want[l][:, :, st - geom["sink"]:en - geom["sink"]]))
    valid, thres = orc.threshold([want[i] for i in range(geom["L"])], 0.3)
    assert thres == float(g["f16/thres"][0])
    assert np.array_equal(np.packbits(valid.numpy().reshape(-1)), g["f16/valid"])


def test_oracle_threshold_on_e2e_d128_512k_fixture():
    """G10 (8 layers x 8 chunks = 512 000 reference-generated scores per dtype): the oracle's global threshold on the reference's own
    scores reproduces the reference's threshold, mask and per-(layer, head) kept counts bit for bit (the scores themselves are
    compared on the GPU box by tests/test_gpu_e2e_parity.py; regenerating the 1.2 GB of seeded inputs here would double the CPU suite)."""
    import e2e_inputs as E
    g = load_golden("g10_e2e_d128_512k.npz")
    geom = E.GEOM_512K
    assert [geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")] == g["geom"].tolist()
    for tag, bf in (("f16", False), ("bf16", True)):
        want = from_bits(g[f"{tag}/score"], bf)
        assert tuple(want.shape) == (geom["L"], 1, geom["Hkv"], geom["N"])
        valid, thres = orc.threshold([want[i] for i in range(geom["L"])], 0.3)
        assert thres == float(g[f"{tag}/thres"][0])
        assert np.array_equal(np.packbits(valid.numpy().reshape(-1)), g[f"{tag}/valid"])
        assert np.array_equal(valid.sum(-1).reshape(geom["L"], geom["Hkv"]).numpy().astype(np.int32), g[f"{tag}/kept"])


def test_oracle_threshold_on_e2e_ratios_fixture():
    """G12 (round 5): the reference's _threshold at ratios 0.1 / 0.3 / 0.6 / 0.9 on its own 512 000 G10 scores per dtype: the oracle
    reproduces every threshold, mask and per-(layer, head) kept count bit for bit."""
    import e2e_inputs as E
    g10, g12 = load_golden("g10_e2e_d128_512k.npz"), load_golden("g12_e2e_ratios.npz")
    geom = E.GEOM_512K
    assert g12["ratios"].tolist() == [0.1, 0.3, 0.6, 0.9]
    for tag, bf in (("f16", False), ("bf16", True)):
        want = from_bits(g10[f"{tag}/score"], bf)
        for r in g12["ratios"].tolist():
            valid, thres = orc.threshold([want[i] for i in range(geom["L"])], r)
            assert thres == float(g12[f"{tag}/thres/{r!r}"][0]), (tag, r)
            assert np.array_equal(np.packbits(valid.numpy().reshape(-1)), g12[f"{tag}/valid/{r!r}"]), (tag, r)
            assert np.array_equal(valid.sum(-1).reshape(geom["L"], geom["Hkv"]).numpy().astype(np.int32), g12[f"{tag}/kept/{r!r}"])


def test_oracle_threshold_uniform_contract_on_rows_with_ties():
    """G13 (round 5): the reference's _threshold_uniform on rows WITH ties (bf16 scores, quantised fp16 scores, a row of one value).
    torch.topk's order among equal values is implementation-defined; the oracle (lowest index first) meets the documented contract
    against the REFERENCE's masks: k per row, same kept multiset, identical away from the boundary value."""
    from conftest import check_uniform_contract
    g = load_golden("g13_uniform_ties.npz")
    for tag, bf in (("bf16", True), ("f16q", False)):
        score = from_bits(g[f"{tag}/score"], bf)
        L, N = score.shape[0], score.shape[-1]
        for r in g["ratios"].tolist():
            ref = torch.from_numpy(np.unpackbits(g[f"{tag}/valid/{r!r}"])[:score.numel()]).bool().view(score.shape)
            valid, thres = orc.threshold_uniform([score[i] for i in range(L)], r)
            assert thres == 0
            check_uniform_contract(f"oracle/{tag}/{r}", score, valid, ref, int(N * r))
            kth = from_bits(g[f"{tag}/kth/{r!r}"], bf).float()
            assert torch.equal(torch.sort(score.float(), dim=-1, descending=True).values[..., int(N * r) - 1:int(N * r)], kth)


def test_oracle_threshold_on_e2e_llama_fixture():
    """G11 (Llama-3.1-8B head geometry, reference-generated): threshold, mask and kept counts of the reference reproduced by the oracle
    on the reference's scores, and one (layer, chunk) get_score call reproduced bit for bit (G = 4 path of the restatement)."""
    import e2e_inputs as E
    g = load_golden("g11_e2e_llama.npz")
    geom = E.GEOM_LLAMA
    assert [geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")] == g["geom"].tolist()
    for tag, bf in (("f16", False), ("bf16", True)):
        want = from_bits(g[f"{tag}/score"], bf)
        valid, thres = orc.threshold([want[i] for i in range(geom["L"])], 0.3)
        assert thres == float(g[f"{tag}/thres"][0])
        assert np.array_equal(np.packbits(valid.numpy().reshape(-1)), g[f"{tag}/valid"])
        assert np.array_equal(valid.sum(-1).reshape(geom["L"], geom["Hkv"]).numpy().astype(np.int32), g[f"{tag}/kept"])
    K0, per_chunk = E.make(torch.bfloat16, geom, E.SEED_LLAMA)
    assert E.checksum(K0, per_chunk) == int(g["bf16/checksum"][0])
    want = from_bits(g["bf16/score"], True)
    ci, l = 1, 0
    st, en, q_len = E.chunks(geom)[ci]
    q, kr = per_chunk[ci][l]
    got = orc.get_score(q, torch.cat([K0[l], kr], dim=2), geom["sink"], st, en)
    assert torch.equal(to_bits_t(got), to_bits_t(want[l][:, :, st - geom["sink"]:en - geom["sink"]]))


def to_bits_t(t):
    return t.contiguous().view(torch.int16)


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_oracle_pinned_at_the_far_end_of_the_headline_context(tag):
    """G14 (round 6, reference-generated): the last two scoring chunks of one layer of a 131 072-token context at the Qwen2.5-7B head
    geometry - chunk starts 128 032 / 130 032, key length 133 k, the short last chunk (1 072 tokens).  The oracle reproduces the
    reference's scores bit for bit there too."""
    import e2e_inputs as E
    g = load_golden("g14_far_context.npz")
    geom = E.GEOM_FAR
    dt = torch.float16 if tag == "f16" else torch.bfloat16
    K0, far = E.make_far(dt)
    assert E.checksum([K0], [[(q, kr)] for (_, _, _, q, kr) in far]) == int(g[f"{tag}/checksum"][0])
    want = from_bits(g[f"{tag}/score"], tag == "bf16")
    got = torch.cat([orc.get_score(q, torch.cat([K0, kr], dim=2), geom["sink"], st, en) for (st, en, _, q, kr) in far], dim=-1)
    assert torch.equal(to_bits_t(got), to_bits_t(want))


def test_oracle_threshold_on_the_full_layer_fixture():
    """G15 (round 6): the oracle's threshold / mask on the reference's own 524 288 scores of one full layer of the headline context."""
    g = load_golden("g15_full_layer.npz")
    L, H, Hkv, D, sink, N, chunk = [int(x) for x in g["geom"]]
    for tag, bf in (("f16", False), ("bf16", True)):
        want = from_bits(g[f"{tag}/score"], bf).view(1, 1, Hkv, N)
        valid, thres = orc.threshold([want[0]], 0.3)
        assert thres == float(g[f"{tag}/thres"][0])
        assert np.array_equal(np.packbits(valid.numpy().reshape(-1)), g[f"{tag}/valid"])
        assert np.array_equal(valid.sum(-1).reshape(Hkv).numpy().astype(np.int32), g[f"{tag}/kept"])
