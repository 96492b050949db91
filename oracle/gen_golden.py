"""Generate golden vectors by IMPORTING the reference's own Python (runs only in the build container,
where /root/reference exists; nothing from the reference is copied — only inputs/outputs are stored).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Shims: ``tiny_api_cuda.update_flatten_view`` (the reference's only native op, CUDA-only) is
replaced by the CPU restatement of csrc/csrc/cuda_api.cu:29-39 from ``oracle/kvzip_oracle.py``;
``transformers.HybridCache`` (removed in transformers 5.x, used only by the out-of-scope Gemma3 cache) is a
dummy class; ``key_cache/value_cache/_seen_tokens`` (created by transformers 4.51.3's DynamicCache.__init__)
are seeded by hand.

Half tensors are stored as uint16 bit patterns (numpy has no bfloat16).
"""
from __future__ import annotations

import glob
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import kvzip_oracle as orc  # noqa: E402


def install_shims():
    mod = types.ModuleType("tiny_api_cuda")
    mod.update_flatten_view = orc.update_flatten_view
    sys.modules["tiny_api_cuda"] = mod
    import transformers
    if not hasattr(transformers, "HybridCache"):
        class HybridCache:  # noqa: D401 - dummy, only subclassed by the out-of-scope RetainHybridCache
            pass
        transformers.HybridCache = HybridCache
    sys.path.insert(0, REF)


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


class StubModel(torch.nn.Module):
    def __init__(self, L, H, Hkv, dtype):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1, dtype=dtype))
        self.config = types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)


def make_cache(cls, L, H, Hkv, dtype, evict_range):
    kv = cls(StubModel(L, H, Hkv, dtype), evict_range)
    kv.key_cache, kv.value_cache, kv._seen_tokens = [], [], 0
    return kv


def gen_score(KVScore):
    """G1: (Q, K, sink, start, end) -> score, straight from KVScore._get_score."""
    cases = [
        # name, dtype, Hkv, G, D, sink, ctx N, start, end, q_len
        ("f16_d128_g7_first", torch.float16, 2, 7, 128, 30, 100, 30, 70, 53),
        ("f16_d128_g7_later", torch.float16, 2, 7, 128, 30, 100, 70, 130, 66),
        ("bf16_d128_g7_first", torch.bfloat16, 2, 7, 128, 30, 100, 30, 70, 53),
        ("bf16_d128_g4_later", torch.bfloat16, 2, 4, 128, 17, 150, 87, 167, 96),
        ("f16_d64_g7", torch.float16, 2, 7, 64, 30, 140, 30, 170, 153),
        ("bf16_d64_g7_later", torch.bfloat16, 2, 7, 64, 8, 140, 70, 148, 91),
        ("f16_d128_g5", torch.float16, 3, 5, 128, 5, 60, 5, 65, 73),
        ("f16_d128_g1_tiny", torch.float16, 1, 1, 128, 0, 40, 0, 33, 37),
    ]
    out = {}
    for i, (name, dt, Hkv, G, D, sink, N, start, end, q_len) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        klen = sink + N + q_len
        q = torch.randn(1, Hkv * G, q_len, D, generator=g).to(dt)
        k = torch.randn(1, Hkv, klen, D, generator=g).to(dt)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = Hkv, dt, "cpu", 1
        sc.sink, sc.start_idx, sc.end_idx = sink, start, end
        sc.init_score()
        sc._get_score(q, k, 0)
        out[name + "/q"] = bits(q)
        out[name + "/k"] = bits(k)
        out[name + "/score"] = bits(sc.score[0])
        out[name + "/meta"] = np.array([Hkv, G, D, sink, start, end, q_len, klen, int(dt == torch.bfloat16)], dtype=np.int64)
    np.savez(os.path.join(OUT, "g1_score.npz"), **out)
    print("g1_score", len(cases), "cases")


RATIOS = [0.0, 1e-9, 0.1, 0.3, 0.6, 0.999999, 1.0, 1.5]


def gen_threshold(KVScore):
    """G2: _threshold with heavy ties (bf16) and with many distinct values (fp16); G3: _threshold_uniform (tie-free)."""
    sc = KVScore()
    out = {}
    g = torch.Generator().manual_seed(7)
    L, Hkv, N = 3, 4, 512
    # softmax-max-like scores in (0,1]
    base = torch.rand(L, 1, Hkv, N, generator=g) ** 4
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        score = base.to(dt)
        out[f"{tag}/score"] = bits(score)
        for r in RATIOS:
            valid, thres = sc._threshold(score, r)
            out[f"{tag}/valid/{r!r}"] = np.packbits(valid.numpy().reshape(-1))
            out[f"{tag}/thres/{r!r}"] = np.array([thres], dtype=np.float64)
        # list-of-layers form (what prune() passes after scoring)
        lst = [score[i] for i in range(L)]
        valid, thres = sc._threshold(lst, 0.3)
        assert torch.equal(valid, sc._threshold(score, 0.3)[0])
    # odd sizes: n not a multiple of 8, negative / zero scores, N not a multiple of 8
    odd = torch.randn(2, 1, 3, 157, generator=g).to(torch.float16)
    odd[0, 0, 0, :5] = 0.0
    odd[1, 0, 2, 3:9] = -0.0
    out["odd/score"] = bits(odd)
    for r in (0.0, 0.25, 0.5, 0.9):
        valid, thres = sc._threshold(odd, r)
        out[f"odd/valid/{r!r}"] = np.packbits(valid.numpy().reshape(-1))
        out[f"odd/thres/{r!r}"] = np.array([thres], dtype=np.float64)
    np.savez(os.path.join(OUT, "g2_threshold.npz"), **out)

    # G3: tie-free rows (distinct fp16 values per row)
    out = {}
    L, Hkv, N = 2, 3, 400
    vals = torch.arange(1, 20001, dtype=torch.float32) / 20001.0
    vals = torch.unique(vals.to(torch.float16))
    rows = []
    for i in range(L * Hkv):
        perm = torch.randperm(vals.numel(), generator=g)[:N]
        rows.append(vals[perm])
    score = torch.stack(rows).view(L, 1, Hkv, N)
    out["score"] = bits(score)
    for r in (0.0, 0.1, 0.3, 0.6, 0.9975, 1.0):
        valid, thres = sc._threshold_uniform([score[i] for i in range(L)], r)
        assert thres == 0
        out[f"valid/{r!r}"] = np.packbits(valid.numpy().reshape(-1))
    np.savez(os.path.join(OUT, "g3_threshold_uniform.npz"), **out)
    print("g2/g3 threshold ok")


def gen_head_scores(KVScore):
    """G4: the reference's own known-answer data (utils/head_score/*.pt) and the head-level selection it implies
    (model/wrapper.py:40-58 + attention/score.py:88-102), independent of the context length."""
    sc = KVScore()
    out = {}
    groups = {"qwen2.5-7b": 0.6, "qwen2.5-14b": 0.6, "llama3.1-8b": 0.6}
    for name, _ in groups.items():
        paths = sorted(glob.glob(os.path.join(REF, "utils", "head_score", f"{name}-*.pt")))
        attn = torch.stack([torch.load(p, map_location="cpu").squeeze() for p in paths], 0).amax(0)  # layer x head
        out[f"{name}/head_score"] = bits(attn)
        out[f"{name}/is_bf16"] = np.array([int(attn.dtype == torch.bfloat16)])
        for ctx_len in (64, 1000):
            score = attn.unsqueeze(-1).expand(-1, -1, ctx_len).unsqueeze(1)  # [L,1,Hkv,N]
            for r in (0.3, 0.6, 0.9):
                valid, thres = sc._threshold(score, r)
                kept_heads = valid[..., 0].squeeze(1)  # [L, Hkv]
                assert torch.equal(valid, kept_heads.unsqueeze(1).unsqueeze(-1).expand_as(valid))
                out[f"{name}/kept/{ctx_len}/{r!r}"] = kept_heads.numpy()
                out[f"{name}/thres/{ctx_len}/{r!r}"] = np.array([thres], dtype=np.float64)
    np.savez(os.path.join(OUT, "g4_head_score.npz"), **out)
    print("g4 head scores ok")


def gen_cache_sequence(EvictCache, RetainCache):
    """G5 + G6: a whole EvictCache life cycle on CPU:
    prefill update -> init_score -> 2 scoring chunks (update/_get_score/slice) -> prune -> query prefill (t=7)
    -> 2 decode steps (t=1) -> slice.  RetainCache is run beside it (reference's own cross-oracle)."""
    for tag, dt, level in (("f16_pair", torch.float16, "pair"), ("bf16_pair", torch.bfloat16, "pair"),
                           ("f16_uniform", torch.float16, "pair-uniform")):
        L, H, Hkv, D = 2, 8, 2, 64
        G = H // Hkv
        sink, N = 6, 80
        g = torch.Generator().manual_seed(len(tag))
        ev = make_cache(EvictCache, L, H, Hkv, dt, (sink, sink + N))
        rt = make_cache(RetainCache, L, H, Hkv, dt, (sink, sink + N))
        out = {"meta": np.array([L, H, Hkv, D, sink, N, int(dt == torch.bfloat16)], dtype=np.int64)}
        K0 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dt) for _ in range(L)]
        V0 = [torch.randn(1, Hkv, sink + N, D, generator=g).to(dt) for _ in range(L)]
        for l in range(L):
            ev.update(K0[l].clone(), V0[l].clone(), l)
            rt.update(K0[l].clone(), V0[l].clone(), l)
            out[f"K0/{l}"], out[f"V0/{l}"] = bits(K0[l]), bits(V0[l])
        # scoring: two chunks of 50 / 30 tokens, repeat prompts of q = m + 9 / m + 13 tokens
        chunks = [(sink, sink + 50, 59), (sink + 50, sink + 80, 43)]
        for kv in (ev, rt):
            kv.init_score()
        for ci, (st, en, q_len) in enumerate(chunks):
            for kv in (ev, rt):
                kv.start_idx, kv.end_idx = st, en
            seen = ev._seen_tokens
            for l in range(L):
                q = torch.randn(1, H, q_len, D, generator=g).to(dt)
                kr = torch.randn(1, Hkv, q_len, D, generator=g).to(dt)
                vr = torch.randn(1, Hkv, q_len, D, generator=g).to(dt)
                out[f"sc/{ci}/{l}/q"], out[f"sc/{ci}/{l}/k"], out[f"sc/{ci}/{l}/v"] = bits(q), bits(kr), bits(vr)
                for kv in (ev, rt):
                    kfull, _ = kv.update(kr.clone(), vr.clone(), l)
                    kv._get_score(q, kfull, l)
            for kv in (ev, rt):
                kv.slice(seen)
        for kv in (ev, rt):
            kv.start_idx, kv.get_score = sink, False
        for l in range(L):
            assert torch.equal(ev.score[l], rt.score[l])
            out[f"score/{l}"] = bits(ev.score[l])
        ratio = 0.4
        import io
        from contextlib import redirect_stdout
        with redirect_stdout(io.StringIO()):
            thres, r_ = ev.prune(ratio, level)
            thres2, r2 = rt.prune(ratio, level)
        assert thres == thres2 and r_ == r2 and torch.equal(ev.valid, rt.valid)
        out["ratio"] = np.array([ratio]); out["thres"] = np.array([thres], dtype=np.float64)
        out["r_real"] = np.array([r_], dtype=np.float64)
        out["valid"] = ev.valid.numpy()
        for l in range(L):
            out[f"flatK/{l}"], out[f"flatV/{l}"] = bits(ev.key_cache[l]), bits(ev.value_cache[l])
            out[f"len_k/{l}"] = ev.info["len_k"][l].numpy()
            out[f"cu_len_k/{l}"] = ev.info["cu_len_k"][l].numpy().copy()
            out[f"max_len_k/{l}"] = np.array([int(ev.info["max_len_k"][l])])
        # generation: query prefill of 7 tokens, then two decode steps
        seen = ev._seen_tokens
        for si, t in enumerate((7, 1, 1)):
            for l in range(L):
                q = torch.randn(1, H, t, D, generator=g).to(dt)
                kn = torch.randn(1, Hkv, t, D, generator=g).to(dt)
                vn = torch.randn(1, Hkv, t, D, generator=g).to(dt)
                out[f"gen/{si}/{l}/q"], out[f"gen/{si}/{l}/k"], out[f"gen/{si}/{l}/v"] = bits(q), bits(kn), bits(vn)
                ke, ve = ev.update(kn.clone(), vn.clone(), l)
                qe, ke2, ve2, ie = ev.prepare(q, ke, ve, l)
                kr_, vr_ = rt.update(kn.clone(), vn.clone(), l)
                qr, kr2, vr2, ir = rt.prepare(q, kr_, vr_, l)
                # the reference's own cross-check: Evict (compact + append) == Retain (mask gather)
                assert torch.equal(qe, qr) and torch.equal(ke2, kr2) and torch.equal(ve2, vr2)
                assert torch.equal(ie["cu_len_k"], ir["cu_len_k"]) and int(ie["max_len_k"]) == int(ir["max_len_k"])
                out[f"gen/{si}/{l}/q_out"] = bits(qe)
                out[f"gen/{si}/{l}/k_out"] = bits(ke2.view(-1, D))
                out[f"gen/{si}/{l}/v_out"] = bits(ve2.view(-1, D))
                out[f"gen/{si}/{l}/cu_len_q"] = ie["cu_len_q"].numpy().copy()
                out[f"gen/{si}/{l}/cu_len_k"] = ie["cu_len_k"].numpy().copy()
                out[f"gen/{si}/{l}/max_len"] = np.array([int(ie["max_len_q"]), int(ie["max_len_k"])])
                # attention output of the CPU restatement on exactly these tensors (a13, unpinned)
                cu = ie["cu_len_k"]
                att = orc.varlen_attn(qe, ke2.view(-1, D), ve2.view(-1, D), cu[:-1].tolist(),
                                      (cu[1:] - cu[:-1]).tolist(), t, causal=True)
                out[f"gen/{si}/{l}/attn"] = bits(att)
        out["seen_after_gen"] = np.array([ev._seen_tokens, ev.get_seq_length()])
        ev.slice(seen)
        for l in range(L):
            out[f"sliced/K/{l}"] = bits(ev.key_cache[l])
            out[f"sliced/cu_len_k/{l}"] = ev.info["cu_len_k"][l].numpy().copy()
        out["seen_after_slice"] = np.array([ev._seen_tokens])
        out["mem_gb"] = np.array([ev._mem()])
        np.savez(os.path.join(OUT, f"g6_cache_{tag}.npz"), **out)
        print("g6", tag, "thres", thres, "r_real", r_)


def gen_attn():
    """G7: varlen attention vectors from the oracle restatement (a13; parity unpinned, see module header)."""
    out = {}
    cases = [("f16_decode", torch.float16, 4, 7, 128, 1, [37, 1, 260, 129]),
             ("bf16_decode", torch.bfloat16, 2, 4, 128, 1, [300, 33]),
             ("f16_prefill", torch.float16, 2, 7, 128, 9, [64, 140]),
             ("f16_d64", torch.float16, 2, 7, 64, 3, [50, 131]),
             ("f16_short", torch.float16, 2, 5, 128, 4, [2, 9])]
    for i, (name, dt, Hkv, G, D, q_len, lens) in enumerate(cases):
        g = torch.Generator().manual_seed(900 + i)
        slack = 5
        starts, tot = [], 0
        for ln in lens:
            starts.append(tot)
            tot += ln + slack
        q = torch.randn(Hkv * q_len, G, D, generator=g).to(dt)
        k = torch.randn(tot, D, generator=g).to(dt)
        v = torch.randn(tot, D, generator=g).to(dt)
        o = orc.varlen_attn(q, k, v, starts, lens, q_len, causal=True)
        out[f"{name}/q"], out[f"{name}/k"], out[f"{name}/v"], out[f"{name}/out"] = bits(q), bits(k), bits(v), bits(o)
        out[f"{name}/k_start"] = np.array(starts, dtype=np.int32)
        out[f"{name}/k_len"] = np.array(lens, dtype=np.int32)
        out[f"{name}/meta"] = np.array([Hkv, G, D, q_len, int(dt == torch.bfloat16)], dtype=np.int64)
    np.savez(os.path.join(OUT, "g7_attn.npz"), **out)
    print("g7 attn ok")


def gen_templates():
    """G8: outputs of the reference's template(model_name, task) (model/template.py:5-33) for the families on the path -
    the strings decide sink = len(sys_prompt_ids) and the prompts once a real tokenizer is used."""
    import contextlib
    import io
    import json
    import importlib.util
    # (the file is imported on its own: the reference's `model` package pulls in the HF model wrappers)
    spec = importlib.util.spec_from_file_location("ref_template", os.path.join(REF, "model", "template.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    template = mod.template
    out = {}
    names = ["Llama-3.1-8B-Instruct", "duo", "Qwen2.5-7B-Instruct-1M", "Qwen2.5-14B-Instruct-1M", "Qwen3-8B",
             "gemma3-12b", "gemma-3-4b-it", "LlamaForCausalLM", "Qwen2ForCausalLM", "mistral-7b"]
    with contextlib.redirect_stdout(io.StringIO()):  # the fallback branch prints a warning
        for n in names:
            for task in ("qa", "gsm8k", "squad"):
                out[f"{n}|{task}"] = list(template(n, task))
    with open(os.path.join(OUT, "g8_templates.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("g8 templates ok")


def gen_e2e_d128(KVScore):
    """G9: where north_star states mask parity - Qwen2.5-7B head geometry (H28 Hkv4 D128), 2 layers x 4 scoring chunks of 2000
    tokens, ONE global threshold at ratio 0.3 over all 64 000 scores (attention/score.py:36-65 + :88-102).  Inputs are seeded
    (tests/e2e_inputs.py) and regenerated on the GPU box; only scores, threshold, mask and an input checksum are stored."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_inputs as E
    geom = E.GEOM
    out = {}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        K0, per_chunk = E.make(dt)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = geom["Hkv"], dt, "cpu", geom["L"]
        sc.sink = geom["sink"]
        sc.init_score()
        for ci, (st, en, q_len) in enumerate(E.chunks()):
            sc.start_idx, sc.end_idx = st, en
            for l in range(geom["L"]):
                q, kr = per_chunk[ci][l]
                kfull = torch.cat([K0[l], kr], dim=2)     # what update() returns during the scoring pass (kvcache.py:75-78)
                sc._get_score(q, kfull, l)
        score = torch.stack(sc.score, 0)                  # [L, 1, Hkv, N]
        valid, thres = sc._threshold(sc.score, 0.3)
        out[f"{tag}/score"] = bits(score)
        out[f"{tag}/thres"] = np.array([thres], dtype=np.float64)
        out[f"{tag}/valid"] = np.packbits(valid.numpy().reshape(-1))
        out[f"{tag}/checksum"] = np.array([E.checksum(K0, per_chunk)], dtype=np.int64)
        print("g9", tag, "thres", thres, "kept", int(valid.sum()), "of", valid.numel())
    out["geom"] = np.array([geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g9_e2e_d128.npz"), **out)


def gen_e2e_llama(KVScore):
    """G11 (round 4): G9's procedure at BASELINE config C3's head geometry (Llama-3.1-8B: H32 Hkv8 D128, G = 4): 2 layers x 4 scoring
    chunks, 128 000 scores per dtype under one global threshold, from the REFERENCE's _get_score / _threshold."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_inputs as E
    geom = E.GEOM_LLAMA
    out = {}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        K0, per_chunk = E.make(dt, geom, E.SEED_LLAMA)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = geom["Hkv"], dt, "cpu", geom["L"]
        sc.sink = geom["sink"]
        sc.init_score()
        for ci, (st, en, q_len) in enumerate(E.chunks(geom)):
            sc.start_idx, sc.end_idx = st, en
            for l in range(geom["L"]):
                q, kr = per_chunk[ci][l]
                sc._get_score(q, torch.cat([K0[l], kr], dim=2), l)
        score = torch.stack(sc.score, 0)
        valid, thres = sc._threshold(sc.score, 0.3)
        out[f"{tag}/score"] = bits(score)
        out[f"{tag}/thres"] = np.array([thres], dtype=np.float64)
        out[f"{tag}/valid"] = np.packbits(valid.numpy().reshape(-1))
        out[f"{tag}/kept"] = valid.sum(-1).reshape(geom["L"], geom["Hkv"]).numpy().astype(np.int32)
        out[f"{tag}/checksum"] = np.array([E.checksum(K0, per_chunk)], dtype=np.int64)
        print("g11", tag, "thres", thres, "kept", int(valid.sum()), "of", valid.numel(), flush=True)
    out["geom"] = np.array([geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g11_e2e_llama.npz"), **out)


def gen_e2e_d128_512k(KVScore):
    """G10 (round 4): the same as G9 at 8 layers x 8 scoring chunks = 512 000 scores per dtype under ONE global threshold - 3.5 % of
    the 14.68 M scores of the headline context, produced by the REFERENCE's own _get_score / _threshold (attention/score.py:36-65,
    :88-102).  Stored: score bits, threshold, packed mask, kept count per (layer, head), input checksum."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_inputs as E
    geom = E.GEOM_512K
    out = {}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        K0, per_chunk = E.make(dt, geom, E.SEED_512K)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = geom["Hkv"], dt, "cpu", geom["L"]
        sc.sink = geom["sink"]
        sc.init_score()
        for ci, (st, en, q_len) in enumerate(E.chunks(geom)):
            sc.start_idx, sc.end_idx = st, en
            for l in range(geom["L"]):
                q, kr = per_chunk[ci][l]
                kfull = torch.cat([K0[l], kr], dim=2)
                sc._get_score(q, kfull, l)
            print("g10", tag, "chunk", ci, flush=True)
        score = torch.stack(sc.score, 0)                  # [L, 1, Hkv, N]
        valid, thres = sc._threshold(sc.score, 0.3)
        out[f"{tag}/score"] = bits(score)
        out[f"{tag}/thres"] = np.array([thres], dtype=np.float64)
        out[f"{tag}/valid"] = np.packbits(valid.numpy().reshape(-1))
        out[f"{tag}/kept"] = valid.sum(-1).reshape(geom["L"], geom["Hkv"]).numpy().astype(np.int32)
        out[f"{tag}/checksum"] = np.array([E.checksum(K0, per_chunk)], dtype=np.int64)
        print("g10", tag, "thres", thres, "kept", int(valid.sum()), "of", valid.numel(), flush=True)
        del K0, per_chunk
    out["geom"] = np.array([geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g10_e2e_d128_512k.npz"), **out)


E2E_RATIOS = (0.1, 0.3, 0.6, 0.9)


def gen_e2e_ratios(KVScore):
    """G12 (round 5): the REFERENCE's _threshold (attention/score.py:88-102) at ratios 0.1 / 0.3 / 0.6 / 0.9 on the reference's own
    G10 scores (8 layers x 8 chunks, 512 000 scores per dtype; the scores are read back from g10_e2e_d128_512k.npz, where the
    reference's _get_score put them).  Stored per dtype and ratio: threshold, packed mask, kept count per (layer, head)."""
    g10 = np.load(os.path.join(OUT, "g10_e2e_d128_512k.npz"))
    L, H, Hkv, D, sink, N, chunk = [int(x) for x in g10["geom"]]
    sc = KVScore()
    out = {"ratios": np.array(E2E_RATIOS, dtype=np.float64)}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        score = torch.from_numpy(g10[f"{tag}/score"].astype(np.int16)).view(dt).reshape(L, 1, Hkv, N)
        lst = [score[i] for i in range(L)]                # the list-of-layers form prune() passes after scoring
        for r in E2E_RATIOS:
            valid, thres = sc._threshold(lst, r)
            if r == 0.3:                                  # must reproduce what G10 itself stored
                assert thres == float(g10[f"{tag}/thres"][0]) and np.array_equal(np.packbits(valid.numpy().reshape(-1)), g10[f"{tag}/valid"])
            out[f"{tag}/thres/{r!r}"] = np.array([thres], dtype=np.float64)
            out[f"{tag}/valid/{r!r}"] = np.packbits(valid.numpy().reshape(-1))
            out[f"{tag}/kept/{r!r}"] = valid.sum(-1).reshape(L, Hkv).numpy().astype(np.int32)
            print("g12", tag, r, "thres", thres, "kept", int(valid.sum()), "of", valid.numel(), flush=True)
    np.savez_compressed(os.path.join(OUT, "g12_e2e_ratios.npz"), **out)


def gen_uniform_ties(KVScore):
    """G13 (round 5): _threshold_uniform (attention/score.py:104-120) on rows WITH ties - bf16 softmax-max-like scores (a few hundred
    distinct values per row) and fp16 rows quantised to 64 levels.  torch.topk's choice among equal values is
    implementation-defined, so the fixture pins the CONTRACT the build documents for pair-uniform, not positions among equals:
    exactly k = int(N * ratio) kept per (layer, head), the kept multiset, and the mask wherever a score differs from the row's
    boundary value (the k-th largest)."""
    sc = KVScore()
    g = torch.Generator().manual_seed(13)
    out = {}
    L, Hkv, N = 3, 4, 1000
    base = torch.rand(L, 1, Hkv, N, generator=g) ** 4
    cases = {"bf16": base.to(torch.bfloat16), "f16q": (torch.round(base * 64) / 64).to(torch.float16)}
    cases["f16q"][0, 0, 0, :] = 0.25                      # a row of ONE value: every position ties
    ratios = (0.1, 0.3, 0.6, 0.95)
    out["ratios"] = np.array(ratios, dtype=np.float64)
    for tag, score in cases.items():
        out[f"{tag}/score"] = bits(score)
        for r in ratios:
            valid, thres = sc._threshold_uniform([score[i] for i in range(L)], r)
            assert thres == 0
            k = int(N * r)
            assert bool((valid.sum(-1) == k).all())
            # boundary value of every row = its k-th largest; ties exist there in (almost) every row of these inputs
            kth = torch.sort(score.float(), dim=-1, descending=True).values[..., k - 1:k]
            n_tied = int(((score.float() == kth).sum(-1) > 1).sum())
            out[f"{tag}/valid/{r!r}"] = np.packbits(valid.numpy().reshape(-1))
            out[f"{tag}/kth/{r!r}"] = bits(kth.to(score.dtype))
            print("g13", tag, r, "k", k, "rows with a tie at the boundary:", n_tied, "of", L * Hkv, flush=True)
    np.savez_compressed(os.path.join(OUT, "g13_uniform_ties.npz"), **out)


def gen_far_context(KVScore):
    """G14 (round 6): the REFERENCE's _get_score (attention/score.py:36-65) on the LAST TWO scoring chunks of one layer of a full
    131 072-token context at the Qwen2.5-7B head geometry - chunk starts 128 032 / 130 032, key length 133 k: pins the far end of the
    cache (64-bit offsets, the short last chunk of 1 072 tokens) against the reference itself.  Only the 3 072 x Hkv scores per dtype
    are stored; the inputs come from a seed (tests/e2e_inputs.py:make_far) and carry a checksum."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_inputs as E
    geom = E.GEOM_FAR
    out = {}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        K0, far = E.make_far(dt)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = geom["Hkv"], dt, "cpu", 1
        sc.sink = geom["sink"]
        sc.init_score()
        # the score buffer of the reference grows by concatenation (score.py:30-34): start it at the first stored chunk
        for (st, en, q_len, q, kr) in far:
            sc.start_idx, sc.end_idx = st, en
            sc._get_score(q, torch.cat([K0, kr], dim=2), 0)
        score = sc.score[0]                                # [1, Hkv, sum of the two chunk lengths]
        assert score.shape[-1] == far[-1][1] - far[0][0]
        out[f"{tag}/score"] = bits(score)
        out[f"{tag}/checksum"] = np.array([E.checksum([K0], [[(q, kr)] for (_, _, _, q, kr) in far])], dtype=np.int64)
        print("g14", tag, "chunks", [(st, en, ql) for (st, en, ql, _, _) in far], "scores", tuple(score.shape), flush=True)
    out["geom"] = np.array([geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g14_far_context.npz"), **out)


def gen_full_layer(KVScore):
    """G15 (round 6): ONE FULL LAYER of the headline context from the REFERENCE - all 66 scoring chunks of a 131 072-token context at the
    Qwen2.5-7B head geometry through _get_score (attention/score.py:36-65), then _threshold at ratio 0.3 (:88-102): 524 288 scores per
    dtype, threshold, packed mask, kept per head.  Replaces the extrapolated flip count at the headline size by a measured one (per
    layer).  ~4 minutes of CPU per dtype."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_inputs as E
    geom = E.GEOM_FAR
    out = {}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        it = E.stream_full(dt)
        K0 = next(it)
        acc = E.checksum_update(0, K0)
        sc = KVScore()
        sc.n_heads_kv, sc.dtype, sc.device, sc.n_layers = geom["Hkv"], dt, "cpu", 1
        sc.sink = geom["sink"]
        sc.init_score()
        for ci, (st, en, q_len, q, kr) in enumerate(it):
            acc = E.checksum_update(E.checksum_update(acc, q), kr)
            sc.start_idx, sc.end_idx = st, en
            sc._get_score(q, torch.cat([K0, kr], dim=2), 0)
            if ci % 8 == 0:
                print("g15", tag, "chunk", ci, flush=True)
        score = sc.score[0]
        assert score.shape[-1] == geom["N"]
        valid, thres = sc._threshold(sc.score, 0.3)
        out[f"{tag}/score"] = bits(score)
        out[f"{tag}/thres"] = np.array([thres], dtype=np.float64)
        out[f"{tag}/valid"] = np.packbits(valid.numpy().reshape(-1))
        out[f"{tag}/kept"] = valid.sum(-1).reshape(geom["Hkv"]).numpy().astype(np.int32)
        out[f"{tag}/checksum"] = np.array([acc], dtype=np.int64)
        print("g15", tag, "thres", thres, "kept", int(valid.sum()), "of", valid.numel(), flush=True)
        del K0
    out["geom"] = np.array([geom[k] for k in ("L", "H", "Hkv", "D", "sink", "N", "chunk")], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g15_full_layer.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    install_shims()
    from attention.score import KVScore
    from attention.kvcache import EvictCache, RetainCache
    if "--only-e2e" in sys.argv:
        gen_e2e_d128(KVScore)
        return
    if "--only-e2e-512k" in sys.argv:
        gen_e2e_d128_512k(KVScore)
        return
    if "--only-e2e-llama" in sys.argv:
        gen_e2e_llama(KVScore)
        return
    if "--only-round6" in sys.argv:
        gen_far_context(KVScore)
        gen_full_layer(KVScore)
        return
    if "--only-full-layer" in sys.argv:
        gen_full_layer(KVScore)
        return
    if "--only-round5" in sys.argv:
        gen_e2e_ratios(KVScore)
        gen_uniform_ties(KVScore)
        return
    gen_score(KVScore)
    gen_threshold(KVScore)
    gen_head_scores(KVScore)
    gen_cache_sequence(EvictCache, RetainCache)
    gen_attn()
    gen_templates()
    gen_e2e_d128(KVScore)
    gen_e2e_d128_512k(KVScore)
    gen_e2e_llama(KVScore)
    gen_e2e_ratios(KVScore)
    gen_uniform_ties(KVScore)
    gen_far_context(KVScore)
    gen_full_layer(KVScore)
    total = sum(os.path.getsize(p) for p in glob.glob(os.path.join(OUT, "*.npz")))
    print(f"wrote {OUT}: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
