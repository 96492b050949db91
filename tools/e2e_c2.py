#!/usr/bin/env python
"""BASELINE config C2 through the front door: a random-init Qwen2 model at the Qwen2.5-7B geometry
(L28 H28 Hkv4 D128, hidden 3584, bf16 - no download), then exactly the reference's quick-start (README.md:43-57):

    ModelKVzip.prefill(32 768 ids)  ->  scoring  ->  kv.prune(0.3)  ->  generate(query, kv=kv)

through kvzip_amd.wrapper / kvzip_amd.attn (reference model/wrapper.py:169-195, :223-249, :251-284).  Checks: the evicting
and the non-evicting cache generate the same tokens; the scores of one sampled (layer, chunk) call equal the CPU oracle's on the
very tensors the forward pass handed over.  Reports seconds for prefill / scoring (forward + kernels) / prune / generation,
the library's own kernel times and the peak HBM use.

    python tools/e2e_c2.py [--ctx 32768] [--layers 28] [--json out.json] [--no-oracle]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402


def build_model(layers, dtype, dev, vocab=152064):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    cfg = Qwen2Config(vocab_size=vocab, hidden_size=3584, intermediate_size=18944, num_hidden_layers=layers,
                      num_attention_heads=28, num_key_value_heads=4, max_position_embeddings=1 << 20, rope_theta=1e6,
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)   # weights are created in bf16 directly on the device (7.6 B parameters: 15 GB)
    try:
        with torch.device(dev):
            model = Qwen2ForCausalLM(cfg)
    finally:
        torch.set_default_dtype(old)
    return model.eval()


def run(ctx_len=32768, layers=28, dev="cuda:0", oracle=True, max_new_tokens=8, ratio=0.3, verbose=True, fuse_forward=False):
    from kvzip_amd import _lib
    from kvzip_amd.wrapper import ModelKVzip
    lib = _lib.load()
    dtype = torch.bfloat16
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    model = build_model(layers, dtype, dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator().manual_seed(7)
    V = model.config.vocab_size
    sys_ids, post_ids = torch.randint(0, V, (1, 32), generator=g), torch.randint(0, V, (1, 5), generator=g)
    rep = (torch.randint(0, V, (1, 13), generator=g), torch.randint(0, V, (1, 26), generator=g))
    ctx = torch.randint(0, V, (1, ctx_len), generator=g)
    query = torch.randint(0, V, (1, 24), generator=g)
    out = {"statistics_from_the_forward_kernel": fuse_forward, "config": f"Qwen2 random-init L{layers} H28 Hkv4 D128 hidden 3584 bf16, ctx {ctx_len}, ratio {ratio}", "build_s": round(t_build, 2)}
    gens, scores, logits = {}, {}, {}
    for kv_type in ("evict", "retain"):
        m = ModelKVzip(model, kv_type=kv_type, name="Qwen2.5-7B-random", max_new_tokens=max_new_tokens,
                       cache_kwargs=dict(verbose=False))
        m.set_prompt_ids(sys_ids, post_ids)
        m.fuse_forward_score = fuse_forward
        torch.cuda.synchronize(); t0 = time.perf_counter()
        kv = m.prefill(ctx, prefill_chunk_size=16000, do_score=False)      # model/wrapper.py:169-195
        torch.cuda.synchronize(); t_prefill = time.perf_counter() - t0
        captured = {}
        if kv_type == "evict" and oracle:
            orig = kv._get_score
            want_call = (min(3, layers - 1), 1)  # (layer, chunk)
            seen = {"chunk": -1, "last_start": None}

            def spy(q, k, layer_idx):
                if kv.start_idx != seen["last_start"]:
                    seen["chunk"] += 1
                    seen["last_start"] = kv.start_idx
                r = orig(q, k, layer_idx)
                if (layer_idx, seen["chunk"]) == want_call:  # (after the call: the fused update + score call has appended the rows)
                    captured.update(q=q.cpu().clone(), k=k.cpu().clone(), layer=layer_idx, st=kv.start_idx, en=kv.end_idx, sink=kv.sink)
                return r
            kv._get_score = spy
            orig_fwd = kv._score_forward

            def spy_fwd(q, k, v, layer_idx, softmax_scale=None):   # (f2: the forward's attention kernel emits the statistics)
                if kv.start_idx != seen["last_start"]:
                    seen["chunk"] += 1
                    seen["last_start"] = kv.start_idx
                r = orig_fwd(q, k, v, layer_idx, softmax_scale=softmax_scale)
                if r is not None and (layer_idx, seen["chunk"]) == want_call:
                    captured.update(q=q.cpu().clone(), k=k.cpu().clone(), layer=layer_idx, st=kv.start_idx, en=kv.end_idx, sink=kv.sink,
                                    fused_forward=True)
                return r
            kv._score_forward = spy_fwd
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.scoring(kv, ctx, repeat_prompt_ids=rep)                             # model/wrapper.py:223-249
        torch.cuda.synchronize(); t_scoring = time.perf_counter() - t0
        scores[kv_type] = torch.stack([s.clone() for s in kv.score], 0)
        t_forward = None
        if kv_type == "evict":
            # the same scoring pass with the scoring kernels switched off: what the repeat-prompt forward passes cost on their own
            real_get_score = kv._get_score
            kv.get_score = True
            saved = (kv._score_buf, kv._score_log, list(kv._score_fill), kv._log_dirty)
            kv._get_score = lambda q, k, layer_idx: None
            kv._score_forward = lambda q, k, v, layer_idx, softmax_scale=None: None   # (falls back to the no-op _get_score + plain forward)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            start_tmp = kv.start_idx
            kv.end_idx = 0
            for prefill_ids_p, repeat_ids_p in m.self_task(ctx, repeat_prompt_ids=rep):
                kv.end_idx = kv.start_idx + prefill_ids_p.shape[1]
                m(repeat_ids_p, kv, update_cache=False)
                kv.start_idx = kv.end_idx
            kv.start_idx = start_tmp
            torch.cuda.synchronize(); t_forward = time.perf_counter() - t0
            kv._get_score, kv.get_score = real_get_score, False
            kv._score_forward = orig_fwd if (kv_type == "evict" and oracle) else type(kv)._score_forward.__get__(kv)
            kv._score_buf, kv._score_log, kv._score_fill, kv._log_dirty = saved[0], saved[1], saved[2], saved[3]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        thres, r_real = kv.prune(ratio)
        torch.cuda.synchronize(); t_prune = time.perf_counter() - t0
        q_ids = m.apply_template(query)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gens[kv_type] = m.generate(q_ids, kv=kv, return_ids=True).cpu()
        torch.cuda.synchronize(); t_gen = time.perf_counter() - t0
        logits[kv_type] = m(q_ids, kv, return_logits=True).logits[0, -1].float().cpu()
        out[kv_type] = {"prefill_s": round(t_prefill, 3), "scoring_s": round(t_scoring, 3), "prune_s": round(t_prune, 4),
                        "generate_s": round(t_gen, 3), "new_tokens": int(gens[kv_type].shape[1]) + 1, "thres": thres, "real_ratio": r_real,
                        "kv_gb_after_prune": kv._mem(), "scoring_forward_only_s": round(t_forward, 3) if t_forward else None,
                        "note": "scoring_s = repeat-prompt forward passes WITH the scoring kernels on the side streams; "
                                "scoring_forward_only_s = the same passes with _get_score switched off (evict run only)"}
        if captured:
            import kvzip_oracle as orc
            t0 = time.perf_counter()
            want = orc.get_score(captured["q"], captured["k"], captured["sink"], captured["st"], captured["en"])
            lo = captured["st"] - captured["sink"]
            got = scores["evict"][captured["layer"]][:, :, lo:lo + want.shape[-1]].cpu()
            a = got.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
            b = want.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
            key = lambda x: torch.where(x >= 0x8000, 0x8000 - (x - 0x8000) - 1, x + 0x8000)
            d = (key(a) - key(b)).abs()
            out["sampled_call_vs_oracle"] = {"layer": captured["layer"], "window": [captured["st"], captured["en"]], "q_len": captured["q"].shape[2],
                                             "bit_identical": float((d == 0).float().mean()), "within_one_half_ulp": float((d <= 1).float().mean()),
                                             "worst_half_ulps": int(d.max()), "oracle_s": round(time.perf_counter() - t0, 1),
                                             "statistics_from_the_forward_kernel": bool(captured.get("fused_forward"))}
        del kv, m
    out["evict_equals_retain_tokens"] = bool(torch.equal(gens["evict"], gens["retain"]))
    out["evict_equals_retain_scores"] = bool(torch.equal(scores["evict"], scores["retain"]))
    out["last_logit_max_abs_diff_rel"] = float((logits["evict"] - logits["retain"]).abs().max() / logits["retain"].abs().max())
    out["generated_ids"] = gens["evict"][0].tolist()
    out["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    if verbose:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--fused-forward", action="store_true", help="f2: row statistics of the scores from the forward's attention kernel")
    a = ap.parse_args()
    res = run(a.ctx, a.layers, oracle=not a.no_oracle, fuse_forward=a.fused_forward)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)
