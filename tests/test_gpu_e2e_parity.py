"""GPU: eviction-mask parity where north_star states it - D = 128, several layers and chunks, ONE global threshold.

Qwen2.5-7B head geometry (H28 Hkv4 D128), 2 layers x 4 scoring chunks of 2000 tokens (64 000 scores), ratio 0.3, fp16 and
bf16.  The expected scores, threshold and mask were produced by the REFERENCE's own KVScore._get_score / _threshold
(attention/score.py:36-65, :88-102) from the seeded inputs of tests/e2e_inputs.py (oracle/gen_golden.py:gen_e2e_d128) and are
committed as tests/golden/g9_e2e_d128.npz.  Here the same inputs go through the drop-in cache object (update -> _get_score
-> slice per chunk, then prune) and every number north_star names is printed and bounded by the FIXED per-dtype bounds of
conftest.check_score_parity (round 5; rounds 3-4 bounded them at 2x the last measurement).  G12 (round 5) adds the reference's
thresholds and masks at ratios 0.1 / 0.3 / 0.6 / 0.9 on the 512 000-score case.
"""
import types

import numpy as np
import pytest
import torch

import e2e_inputs as E
from conftest import check_score_parity, from_bits, load_golden, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# How many mask entries may flip end to end (the integer part - the mask GIVEN the scores - is asserted bit for bit below).
# An entry can only flip where a score that is not bit-identical to the reference's sits at the threshold: with a fraction f of
# non-identical scores (fp16 ~1e-3, bf16 ~1e-4, conftest.SCORE_BOUNDS) and T scores sharing the threshold value or the grid
# step next to it (fp16: a few hundred of 512 000; bf16: thousands, but its non-identical scores are ten times rarer), the
# expectation is ~ f * T <= 1 per ratio at this size.  Allowed: the largest count ever measured + 1 - 3 (fp16: 1 / 2 / 1 / 0 over the four
# ratios in round 5, 1 at ratio 0.3 in round 6) / 2 (bf16: 0 / 0 / 0 / 1, 0) per ratio on 512 000 scores; 0 on the smaller fixtures.
# Measured on ONE FULL LAYER of the headline context against the reference (G15, tests/test_gpu_far_context.py: 524 288 scores, 483 /
# 35 of them not bit-identical): 0 (fp16) / 1 (bf16) flipped entries - every one a last-bit difference of the fp32 accumulation order
# at the threshold.
FLIPS_512K = {"f16": 3, "bf16": 2}


def _drive(kv, K0, per_chunk, geom):
    """update -> _get_score -> slice per chunk, exactly as model/wrapper.py:223-249 drives the cache object"""
    L, sink = geom["L"], geom["sink"]
    for l in range(L):
        kv.update(K0[l].to(DEV), K0[l].to(DEV), l)     # (values are irrelevant for the scores)
    kv.init_score()
    for ci, (st, en, q_len) in enumerate(E.chunks(geom)):
        kv.start_idx, kv.end_idx = st, en
        seen = kv._seen_tokens
        for l in range(L):
            q, kr = per_chunk[ci][l]
            k_all, _ = kv.update(kr.to(DEV), kr.to(DEV), l)
            kv._get_score(q.to(DEV), k_all, l)
        kv.slice(seen)
    kv.start_idx, kv.get_score = sink, False


def _mask_parity(tag, fixture, geom, seed, case, hamming_allowed, ratios_fixture=None):
    from kvzip_amd.kvcache import EvictCache
    g = load_golden(fixture)
    assert [geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")] == g["geom"].tolist()
    dt = torch.float16 if tag == "f16" else torch.bfloat16
    K0, per_chunk = E.make(dt, geom, seed)
    assert E.checksum(K0, per_chunk) == int(g[f"{tag}/checksum"][0]), "seeded inputs differ from the ones the fixture was made from"
    L, H, Hkv, D, sink, N = (geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N"))
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    _drive(kv, K0, per_chunk, geom)
    want = from_bits(g[f"{tag}/score"], tag == "bf16")                       # [L, 1, Hkv, N] from the reference
    got = torch.stack([s for s in kv.score], 0).cpu()
    d = ulp_diff(got, want)
    exact, within1, worst = float((d == 0).float().mean()), float((d <= 1).float().mean()), int(d.max())
    want_valid = torch.from_numpy(np.unpackbits(g[f"{tag}/valid"])[:want.numel()]).bool().view(want.shape)
    want_thres = float(g[f"{tag}/thres"][0])
    thres, r_real = kv.prune(0.3)
    flips = (kv.valid.cpu() != want_valid)
    ham = int(flips.sum())
    n_chunks = len(E.chunks(geom))
    print(f"\nE2E D=128 {tag}: {want.numel()} scores of {L} layers x {n_chunks} chunks: {exact:.5f} bit-identical, {within1:.5f} within "
          f"one half-ulp, worst {worst}; thres {thres!r} vs reference {want_thres!r} ({'EQUAL' if thres == want_thres else 'DIFFERENT'}); "
          f"mask Hamming distance {ham} of {want.numel()}; kept ratio {r_real:.5f}")
    # what sat at the threshold: the scores equal to it (evicted by the strict >, score.py:95-96) and the ones one step above
    tb = torch.tensor([want_thres]).to(dt)
    at = int((want == tb).sum())
    print(f"   scores equal to the threshold value: {at}; scores that differ from the reference's AND lie within 8 steps of the "
          f"threshold: {int(((d > 0) & (ulp_diff(want, tb.expand_as(want)) <= 8)).sum())}")
    if ham:
        idx = flips.nonzero()[:16]
        for i in idx:
            i = tuple(int(x) for x in i)
            print(f"   flipped entry {i}: reference score {float(want[i])!r} (kept {bool(want_valid[i])}), ours {float(got[i])!r}")
    check_score_parity(f"{case}/{tag}", got, want)
    assert thres == want_thres, "the global threshold (one order statistic over all layers and chunks) must be the reference's"
    # The mask is an integer function of the scores: given the reference's scores it is reproduced bit for bit (end of this test).  End
    # to end an entry can only flip where a score that is NOT bit-identical to the reference's sits right at the threshold (the strict
    # > evicts the ties, attention/score.py:95-96): every flipped entry must be such a score, at most `worst` steps of the 16-bit grid
    # away from the threshold - nothing else may ever flip - and their number is bounded by the count the caller states.
    near = (d > 0) & (ulp_diff(want, tb.expand_as(want)) <= max(worst, 1))
    assert bool((flips & ~near).sum() == 0), "a mask entry flipped although its score equals the reference's or lies away from the threshold"
    assert ham <= hamming_allowed, f"eviction mask differs from the reference's in {ham} entries (allowed {hamming_allowed})"
    if f"{tag}/kept" in g.files:
        kept = torch.stack(kv.info["len_k"]).cpu() - sink
        assert torch.equal(kept.int(), torch.from_numpy(g[f"{tag}/kept"]).int()) or ham > 0
    if ratios_fixture is not None:
        # G12: the reference's _threshold at four ratios on ITS scores; ours: the selection kernels on OUR scores.  Threshold equal,
        # flips only at the threshold and bounded; and from the reference's scores the masks bit for bit.
        from kvzip_amd import ops
        g12 = load_golden(ratios_fixture)
        got_dev, want_dev = got.to(DEV), want.to(DEV)
        for r in g12["ratios"].tolist():
            want_valid_r = torch.from_numpy(np.unpackbits(g12[f"{tag}/valid/{r!r}"])[:want.numel()]).bool().view(want.shape)
            want_thres_r = float(g12[f"{tag}/thres/{r!r}"][0])
            v, t, _, _ = ops.select_threshold(got_dev, r, row_len=N)
            t = float(t.item())
            fl = (v.cpu().bool().view(want.shape) != want_valid_r)
            tbr = torch.tensor([want_thres_r]).to(dt)
            near_r = (d > 0) & (ulp_diff(want, tbr.expand_as(want)) <= max(worst, 1))
            print(f"   ratio {r}: thres {t!r} vs reference {want_thres_r!r} ({'EQUAL' if t == want_thres_r else 'DIFFERENT'}), mask Hamming "
                  f"{int(fl.sum())} of {want.numel()}, scores equal to the threshold value {int((want == tbr).sum())}")
            assert t == want_thres_r, (tag, r)
            assert bool((fl & ~near_r).sum() == 0), (tag, r, "a mask entry flipped away from the threshold")
            assert int(fl.sum()) <= hamming_allowed, (tag, r, int(fl.sum()))
            v2, t2, _, _ = ops.select_threshold(want_dev, r, row_len=N)
            assert float(t2.item()) == want_thres_r and torch.equal(v2.cpu().bool().view(want.shape), want_valid_r), (tag, r)
            assert np.array_equal(want_valid_r.sum(-1).reshape(L, Hkv).numpy().astype(np.int32), g12[f"{tag}/kept/{r!r}"])
    # identical scores -> identical mask, bit for bit (the integer part of the contract)
    kv2 = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=dt, verbose=False)
    for l in range(L):
        kv2.update(K0[l].to(DEV), K0[l].to(DEV), l)
    kv2.score = [want[l].to(DEV) for l in range(L)]
    t2, _ = kv2.prune(0.3)
    assert t2 == want_thres and torch.equal(kv2.valid.cpu(), want_valid)


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_e2e_mask_parity_d128_multilayer(tag):
    """G9: 2 layers x 4 chunks = 64 000 scores; the mask must be the reference's bit for bit (Hamming 0, round 4: no allowance)."""
    _mask_parity(tag, "g9_e2e_d128.npz", E.GEOM, 4242, "e2e_d128", 0)


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_e2e_mask_parity_d128_512k(tag):
    """G10 (round 4): 8 layers x 8 chunks = 512 000 scores under ONE global threshold, expected values from the REFERENCE's own
    _get_score / _threshold (oracle/gen_golden.py:gen_e2e_d128_512k) - the sentence of north_star ("eviction masks bit-exactly") on
    3.5 % of the headline context's scores: threshold EQUAL asserted, every flipped mask entry must be a non-identical score at the
    threshold, and their count is bounded by FLIPS_512K (measured in round 4 at ratio 0.3: fp16 2, bf16 0).  Round 5 (G12): the same
    for the reference's thresholds and masks at ratios 0.1 / 0.3 / 0.6 / 0.9."""
    # measured on MI355X (profiles/r4_parity_e2e.txt): fp16 threshold EQUAL, 2 of 512 000 entries flipped - two scores that differ from
    # the reference's by one step of the 16-bit grid and sit exactly at / one step above the threshold value (275 scores share that
    # value); bf16: 0.  That is the number north_star's "bit-exact masks" comes down to at this size: 3.9e-6 of the entries, each one
    # explained by a last-bit difference of the fp32 accumulation order in Q.K^T (the reference's own CPU / GPU builds differ the same way).
    _mask_parity(tag, "g10_e2e_d128_512k.npz", E.GEOM_512K, E.SEED_512K, "e2e_d128_512k", FLIPS_512K[tag], "g12_e2e_ratios.npz")


@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_e2e_mask_parity_llama_geometry(tag):
    """G11 (round 4): BASELINE config C3's head geometry (Llama-3.1-8B: H32 Hkv8 D128, G = 4 - another row partition of both scoring
    passes than Qwen2.5-7B's G = 7), 2 layers x 4 chunks = 128 000 scores per dtype from the REFERENCE's own _get_score / _threshold:
    threshold equal, mask identical."""
    _mask_parity(tag, "g11_e2e_llama.npz", E.GEOM_LLAMA, E.SEED_LLAMA, "e2e_llama", 0)


def test_full_size_masks_do_not_depend_on_streams():
    """The headline context in full (Qwen2.5-7B geometry, 131 072 tokens, 66 chunks x 28 layers = 14.68 M scores, ratio 0.3): scored on
    one stream and on three side streams, with update + _get_score as one library call or as two, the 16-bit scores, the threshold
    and the eviction mask must be BIT-IDENTICAL (the side-stream pipeline and the atomics of the deferred merge are order
    independent), and so must two runs of the same configuration.  (The scores taken from the forward's own QK^T - f2, off by
    default - differ in the summation order of the row sums by design, DESIGN.md 3.6: they are compared in
    test_score_forward_fused_statistics, not here.)"""
    from kvzip_amd.kvcache import EvictCache
    L, H, Hkv, D, sink, N, chunk = 28, 28, 4, 128, 32, 131072, 2000
    cfg = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)
    g = torch.Generator(device=DEV).manual_seed(99)
    chunks = []
    for c, st in enumerate(range(0, N, chunk)):
        m = min(chunk, N - st)
        chunks.append((sink + st, sink + st + m, m + (13 if c == 0 else 26)))
    q_max = max(c[2] for c in chunks)
    cap = sink + N + q_max + 8
    store = []
    for l in range(L):
        t = torch.empty((1, Hkv, cap, D), dtype=torch.float16, device=DEV)
        t[:, :, :sink + N] = torch.randn(1, Hkv, sink + N, D, generator=g, device=DEV, dtype=torch.float32).half()
        store.append(t)
    # two sets of repeat-pass inputs, alternating over the chunks (the context keys differ from chunk to chunk anyway)
    Q = [[torch.randn(1, H, q_max, D, generator=g, device=DEV, dtype=torch.float32).half() for _ in range(L)] for _ in range(2)]
    Kr = [[torch.randn(1, Hkv, q_max, D, generator=g, device=DEV, dtype=torch.float32).half() for _ in range(L)] for _ in range(2)]

    def run(nstreams, fused):
        kv = EvictCache(cfg, (sink, sink + N), device=DEV, dtype=torch.float16, verbose=False)
        kv.n_score_streams = nstreams
        kv.fuse_update_score = fused
        kv.adopt_dense(store, store, sink + N)   # (K doubles as V: only the scores matter here)
        kv.init_score()
        for c, (st, en, q_len) in enumerate(chunks):
            kv.start_idx, kv.end_idx = st, en
            seen = kv._seen_tokens
            for l in range(L):
                q, kr = Q[c % 2][l][:, :, :q_len], Kr[c % 2][l][:, :, :q_len]
                k_all, v_all = kv.update(kr, kr, l)
                kv._get_score(q, k_all, l)
            kv.slice(seen)
        kv.start_idx, kv.get_score = sink, False
        kv.valid = None
        thres, r_real = kv._select(0.3, "pair")          # (deferred path: the finalize launch carries the first histogram)
        score = torch.stack([s for s in kv.score]).clone()
        return score, thres, kv.valid.clone(), r_real

    s1, t1, v1, r1 = run(1, True)
    assert s1.numel() == L * Hkv * N
    for n, fused in ((3, True), (3, False), (1, False), (3, True)):
        s2, t2, v2, _ = run(n, fused)
        same = torch.equal(s1.view(torch.int16), s2.view(torch.int16))
        print(f"\nFULL SIZE {n} stream(s), update + _get_score {'one call' if fused else 'two calls'}: scores "
              f"{'bit-identical' if same else 'DIFFERENT'}, thres {t2!r} vs {t1!r}, mask Hamming {int((v1 != v2).sum())} of {v1.numel()}")
        assert same and t1 == t2 and torch.equal(v1, v2)
    # the selection through the plain three-launch path on the same scores: the same mask (the fused histogram is exactly the first pass)
    from kvzip_amd import ops
    v3, t3, k3, _ = ops.select_threshold(s1, 0.3, row_len=N)
    assert float(t3.item()) == t1 and torch.equal(v3, v1)
    print(f"\nFULL SIZE kept ratio {r1:.5f}, threshold {t1!r}, {int(v1.sum())} of {v1.numel()} entries kept")
