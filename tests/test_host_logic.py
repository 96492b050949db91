"""CPU: host-side logic of the drop-in classes that does not need a kernel — dense cache bookkeeping, score buffer
management, scoring-chunk construction, sharding — mirrors of reference model/wrapper.py:18-37,197-221 and
attention/kvcache.py:41-121, attention/score.py:25-34."""
import types

import pytest
import torch

import kvzip_oracle as orc
from kvzip_amd.kvcache import EvictCache, RetainCache
from kvzip_amd.wrapper import ModelKVzip, chunk_fn
from kvzip_amd.template import template


def _cfg(L=2, H=4, Hkv=2):
    return types.SimpleNamespace(num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hkv)


@pytest.mark.parametrize("cls", [EvictCache, RetainCache])
def test_dense_update_slice_get_seq_length(cls):
    """update appends in place (amortised growth), slice drops the query/answer rows, _seen_tokens is the logical length."""
    kv = cls(_cfg(), (3, 13), device="cpu", dtype=torch.float16, reserve=4, verbose=False)
    assert kv.get_seq_length() == 0 and kv.sink == 3 and kv.ctx_len == 10 and not kv.pruned and not kv.get_score
    g = torch.Generator().manual_seed(0)
    parts = [[torch.randn(1, 2, t, 8, generator=g).half() for t in (13, 5, 9)] for _ in range(2)]
    for step in range(3):
        for l in range(2):
            K, V = kv.update(parts[l][step], parts[l][step] * 2, l)
            want = torch.cat(parts[l][:step + 1], dim=2)
            assert torch.equal(K, want) and torch.equal(V, want * 2)  # survives two re-allocations (reserve = 4)
    assert kv.get_seq_length() == 27 and kv._seen_tokens == 27
    kv.slice(13)
    assert kv.get_seq_length() == 13 and kv.key_cache[1].shape == (1, 2, 13, 8)
    assert torch.equal(kv.key_cache[0], parts[0][0])
    # a later update continues right after the kept rows
    K, _ = kv.update(parts[0][1], parts[0][1], 0)
    assert torch.equal(K, torch.cat([parts[0][0], parts[0][1]], dim=2))
    assert kv._mem() == 0.0


def test_get_valid_pads_sink_and_tail():
    kv = EvictCache(_cfg(L=1), (2, 7), device="cpu", dtype=torch.float16, verbose=False)
    kv.valid = torch.tensor([[[[1, 0, 0, 1, 1], [0, 0, 0, 0, 1]]]], dtype=torch.bool)
    full = kv._get_valid(0, 10)
    assert full.shape == (1, 2, 10)
    assert full[0, 0].tolist() == [True, True, True, False, False, True, True, True, True, True]
    assert torch.equal(full, orc.get_valid(kv.valid[0], 2, 10))


def test_score_buffer_bookkeeping():
    """init_score / _update_score keep L views into one [L,1,Hkv,N] buffer; _stacked_score returns it without a copy."""
    kv = EvictCache(_cfg(L=2), (1, 9), device="cpu", dtype=torch.float16, verbose=False)
    kv.init_score()
    assert kv.get_score and len(kv.score) == 2 and kv.score[0].shape == (1, 2, 0)
    a = [torch.rand(1, 2, 5).half(), torch.rand(1, 2, 3).half()]
    for l in range(2):
        for blk in a:
            kv._update_score(l, blk)
    for l in range(2):
        assert kv.score[l].shape == (1, 2, 8) and torch.equal(kv.score[l], torch.cat(a, dim=-1))
    stacked = kv._stacked_score(kv.score)
    assert stacked.data_ptr() == kv._score_buf.data_ptr() and stacked.shape == (2, 1, 2, 8)
    # scores assigned from outside (head-level tensor or a list of foreign tensors) are stacked / passed through
    foreign = [torch.rand(1, 2, 8).half() for _ in range(2)]
    assert torch.equal(kv._stacked_score(foreign), torch.stack(foreign))
    t = torch.rand(2, 1, 2, 8).half()
    assert kv._stacked_score(t) is t
    # capacity growth when more scores arrive than ctx_len announced
    kv._update_score(0, torch.rand(1, 2, 4).half())
    assert kv.score[0].shape[-1] == 12 and torch.equal(kv.score[1], torch.cat(a, dim=-1))


def test_chunk_fn_matches_reference_chunking():
    ids = torch.arange(4500).view(1, -1)
    chunks = chunk_fn(ids, 2000)
    assert [c.shape[1] for c in chunks] == [2000, 2000, 500]
    assert torch.equal(torch.cat(chunks, dim=1), ids)
    assert len(chunk_fn(ids[:, :2000], 2000)) == 1 and len(chunk_fn(ids[:, :2001], 2000)) == 2


def test_self_task_builds_repeat_prompts():
    """(chunk ids, repeat prompt ++ postfix ++ chunk ids); later chunks quote the last 8 tokens of the previous chunk
    (reference model/wrapper.py:197-221)."""
    m = ModelKVzip.__new__(ModelKVzip)  # no model needed for the prompt construction
    m.device = torch.device("cpu")
    m.postfix_ids = torch.tensor([[900, 901]])
    first, later = torch.tensor([[1, 2, 3]]), torch.tensor([[4, 5, 6, 7]])
    ctx = torch.arange(100, 100 + 250).view(1, -1)
    out = m.self_task(ctx, chunk_size=100, prev_postfix_size=8, repeat_prompt_ids=(first, later))
    assert len(out) == 3
    a0, r0 = out[0]
    assert torch.equal(a0, ctx[:, :100]) and torch.equal(r0, torch.cat([first, m.postfix_ids, a0], dim=1))
    a1, r1 = out[1]
    assert torch.equal(r1, torch.cat([later, ctx[:, 92:100], m.postfix_ids, a1], dim=1))
    a2, r2 = out[2]
    assert a2.shape[1] == 50 and torch.equal(r2[:, 4:12], ctx[:, 192:200])
    # repeat prompt overhead: q_len - m = len(prompt) + len(postfix) (+ 8 quoted tokens for later chunks)
    assert r0.shape[1] - 100 == 5 and r1.shape[1] - 100 == 14


def test_template_strings_equal_reference_outputs():
    """template(name, task) returns exactly what the reference's model/template.py:5-33 returns (fixture g8 holds its outputs,
    generated by oracle/gen_golden.py): with a real tokenizer these strings decide the sink length and the prompts."""
    import json
    import os
    from conftest import GOLDEN
    want = json.load(open(os.path.join(GOLDEN, "g8_templates.json")))
    assert len(want) == 30
    for key, (prefix, postfix) in want.items():
        name, task = key.split("|")
        assert template(name, task) == (prefix, postfix), key


def test_load_head_score_is_a_stride0_view_and_save_layout(tmp_path):
    """load_head_score: element-wise maximum over a model's files, expanded over the context WITHOUT materialising it
    (reference model/wrapper.py:40-58); the file stem follows the reference's name mapping."""
    from kvzip_amd.wrapper import head_score_name, load_head_score
    assert head_score_name("Qwen2.5-7B-Instruct-1M") == "qwen2.5-7b"
    assert head_score_name("Llama-3.1-8B-Instruct") == "llama3.1-8b" and head_score_name("tiny") == "tiny"
    a = torch.rand(3, 2).half()
    b = torch.rand(3, 2).half()
    torch.save(a, tmp_path / "qwen2.5-14b-squad-0.pt")
    torch.save(b.unsqueeze(0), tmp_path / "qwen2.5-14b-scbench_kv-3.pt")  # the reference squeezes
    s = load_head_score("Qwen2.5-14B-Instruct-1M", 1 << 20, str(tmp_path), "cpu")
    assert s.shape == (3, 1, 2, 1 << 20) and s.stride(-1) == 0
    assert torch.equal(s[:, 0, :, 0], torch.maximum(a, b))
    with pytest.raises(FileNotFoundError):
        load_head_score("nothing", 8, str(tmp_path), "cpu")


def test_tiny_api_cuda_alias_is_the_native_op():
    """`from tiny_api_cuda import update_flatten_view` (reference attention/kvcache.py:10) resolves to the HIP-backed op."""
    import tiny_api_cuda
    from kvzip_amd import ops
    assert tiny_api_cuda.update_flatten_view is ops.update_flatten_view


def test_template_strings():
    for name in ("Qwen2.5-7B-Instruct-1M", "Llama-3.1-8B-Instruct", "Qwen3-8B", "tiny"):
        prefix, postfix = template(name, "qa")
        assert isinstance(prefix, str) and isinstance(postfix, str)
    assert "<|im_start|>" in template("Qwen2.5-7B", "qa")[0]
    assert "<|start_header_id|>" in template("Llama-3.1-8B", "qa")[0]


def test_unsupported_kv_types_are_refused():
    m = ModelKVzip.__new__(ModelKVzip)
    m.kv_type, m.model, m.cache_kwargs = "int4static", None, {}
    with pytest.raises(NotImplementedError):
        m._init_kv()


def test_product_path_fails_loudly_without_a_device():
    """No CPU fallback: handing CPU tensors to a kernel wrapper raises instead of computing something else."""
    from kvzip_amd import ops
    from kvzip_amd._lib import KvzError
    with pytest.raises(KvzError):
        ops.select_threshold(torch.rand(1, 1, 1, 16).half(), 0.3)
    with pytest.raises((KvzError, AssertionError)):
        ops.score_chunk(torch.rand(1, 2, 4, 64).half(), torch.rand(1, 1, 20, 64).half(), 0, 0, 8)


# ------------------------------------------------------------------------------------------------
# static partition of the row-statistics pass (kvz_score.hip: PaPlan) - host code, no GPU needed
# ------------------------------------------------------------------------------------------------
def _score_plan(sink, m, q_len, G, Hkv):
    import ctypes as C
    from kvzip_amd import _lib
    lib = _lib.load()
    unit, tile = (C.c_uint16 * 257)(), (C.c_uint16 * 257)()
    nb, max_seg, rows = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib.kvz_debug_score_plan(sink, m, q_len, G, Hkv, unit, tile, C.byref(nb), C.byref(max_seg), C.byref(rows))
    assert rc == 0
    return list(unit), list(tile), nb.value, max_seg.value, rows.value


def _ntiles(rt, rows, R, q_len, sink, m):
    r0, r1 = rt * rows, min(R - 1, rt * rows + rows - 1)
    qmax = (r1 % q_len) if r0 // q_len == r1 // q_len else q_len - 1
    return (sink + m + qmax + 1 + 127) // 128


@pytest.mark.parametrize("geom", [(32, 2000, 2026, 7, 4), (32, 2000, 2013, 7, 4), (32, 2000, 2026, 4, 8), (16, 600, 626, 4, 2),
                                  (4, 32, 40, 7, 4), (0, 1, 1, 1, 1), (30, 700, 713, 7, 2), (32, 2000, 2026, 8, 1),
                                  (8, 100, 3000, 2, 1), (32, 1072, 1098, 7, 4), (0, 5000, 5026, 16, 16), (40, 129, 17, 5, 3)])
def test_score_plan_is_a_partition(geom):
    """Every key tile of every (row tile, head) unit belongs to exactly one block, blocks are ordered and non-empty, the last cut
    is the end of the sequence, ranges differ by a bounded amount, and `max_seg` bounds the partial statistics per unit."""
    sink, m, q_len, G, Hkv = geom
    unit, tile, nb, max_seg, rows = _score_plan(sink, m, q_len, G, Hkv)
    R = G * q_len
    RT = (R + rows - 1) // rows
    U = RT * Hkv
    nt = [_ntiles(u // Hkv, rows, R, q_len, sink, m) for u in range(U)]
    total = sum(nt)
    assert 1 <= nb <= 256 and nb == min(256, total)
    assert (unit[0], tile[0]) == (0, 0) and (unit[nb], tile[nb]) == (U, 0)
    pref = [0]
    for x in nt:
        pref.append(pref[-1] + x)
    cuts = []
    for b in range(nb + 1):
        assert unit[b] <= U and (unit[b] == U or tile[b] < nt[unit[b]])
        cuts.append(pref[unit[b]] + tile[b])
    sizes = [cuts[b + 1] - cuts[b] for b in range(nb)]
    assert all(s >= 1 for s in sizes) and sum(sizes) == total
    assert max(sizes) <= 2 * (total / nb) + 2            # (tiles that hold a causal limit weigh 1.5: ranges are equal up to that)
    touched = [0] * U                                      # blocks per unit
    for b in range(nb):
        u0, u1 = unit[b], unit[b + 1] if tile[b + 1] > 0 else unit[b + 1] - 1
        for u in range(u0, min(u1, U - 1) + 1):
            touched[u] += 1
    assert max(touched) <= max_seg


def test_fastdiv_matches_integer_division():
    """The kernels divide by launch-invariant integers (q_len, Hkv, items per slice) with a multiply-shift (Granlund-Montgomery
    round-up, exact for 31-bit numerators): compare with Python's // and % on edge and random values."""
    import ctypes as C
    import random
    from kvzip_amd import _lib
    lib = _lib.load()
    rng = random.Random(5)
    q, r = C.c_int(0), C.c_int(0)
    divisors = [1, 2, 3, 4, 5, 7, 8, 13, 26, 28, 31, 32, 33, 127, 128, 129, 224, 626, 713, 1000, 2013, 2026, 4096, 65535, 65536,
                65537, 1 << 20, (1 << 30) - 1, 1 << 30, (1 << 31) - 1] + [rng.randint(1, (1 << 31) - 1) for _ in range(40)]
    for d in divisors:
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - 2, 1 << 30] + [rng.randint(0, (1 << 31) - 1) for _ in range(200)]
        ns += [k * d + off for k in (1, 7, 1000, ((1 << 31) - 1) // d) for off in (-1, 0, 1)]
        for n in ns:
            if not 0 <= n < (1 << 31):
                continue
            assert lib.kvz_debug_fastdiv(d, n, C.byref(q), C.byref(r)) == 0
            assert (q.value, r.value) == (n // d, n % d), (d, n, q.value, r.value)


def test_next_token_follows_hf_generate_knobs():
    """ModelKVzip.generate keeps the reference's gen_kwargs (model/wrapper.py:81-94: do_sample / temperature / top_k / top_p, handed to
    HF generate there).  The token choice is restated in ModelKVzip.next_token: greedy = argmax, and with sampling the SUPPORT and the
    probabilities after temperature -> top-k -> top-p must be those of transformers' own logits warpers."""
    import torch
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from kvzip_amd.wrapper import ModelKVzip as M
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 200, generator=g) * 3
    assert torch.equal(M.next_token(logits, {"do_sample": False, "temperature": 0.3, "top_k": 5}), logits.argmax(-1, keepdim=True))
    for kw in ({"temperature": 0.7, "top_k": 12, "top_p": 0.8}, {"temperature": 1.0, "top_k": None, "top_p": 0.5},
               {"temperature": 1.3, "top_k": 3, "top_p": 1}, {"temperature": 1.0, "top_k": None, "top_p": 1}):
        x = logits.clone()
        ids = torch.zeros(3, 1, dtype=torch.long)
        if kw["temperature"] != 1.0:
            x = TemperatureLogitsWarper(kw["temperature"])(ids, x)
        if kw["top_k"]:
            x = TopKLogitsWarper(kw["top_k"])(ids, x)
        if kw["top_p"] < 1:
            x = TopPLogitsWarper(kw["top_p"])(ids, x)
        want = torch.softmax(x, -1)
        # empirical check of the support: many draws never leave it, and the most probable token dominates as it should
        draws = torch.cat([M.next_token(logits, dict(kw, do_sample=True), g) for _ in range(400)], dim=1)   # [3, 400]
        for b in range(3):
            support = set(torch.nonzero(want[b] > 0).view(-1).tolist())
            assert set(draws[b].tolist()) <= support, kw
            top = int(want[b].argmax())
            freq = float((draws[b] == top).float().mean())
            assert abs(freq - float(want[b, top])) < 0.12, (kw, freq, float(want[b, top]))
    # degenerate knobs reduce to greedy
    assert torch.equal(M.next_token(logits, {"do_sample": True, "top_k": 1}, g), logits.argmax(-1, keepdim=True))
    assert torch.equal(M.next_token(logits, {"do_sample": True, "top_p": 1e-9}, g), logits.argmax(-1, keepdim=True))


def test_bench_parity_gate_uses_the_test_suite_bounds():
    """bench.py turns its parity sample into an assertion (exit code 3): its bounds are the ones of conftest.SCORE_BOUNDS, and the
    checker flags exactly what they forbid."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from conftest import SCORE_BOUNDS
    assert bench.PARITY_BOUNDS == {"f16": SCORE_BOUNDS[torch.float16], "bf16": SCORE_BOUNDS[torch.bfloat16]}
    n = 20000
    d = torch.zeros(n, dtype=torch.int64)
    assert bench.parity_violations("f16", d) == []
    d[:40] = 1                                          # 0.2 % not identical: inside the fp16 bound, outside the bf16 one
    assert bench.parity_violations("f16", d) == [] and len(bench.parity_violations("bf16", d)) == 1
    d[0] = 9                                            # worst step count
    assert len(bench.parity_violations("f16", d)) == 1
    # masks: a flip is tolerated only where a non-identical score sits at the threshold
    want = torch.linspace(0.01, 0.99, n).half()
    got = want.clone()
    d = torch.zeros(n, dtype=torch.int64)
    v_ref = want.float() > 0.5
    v_got = v_ref.clone()
    j = int(torch.nonzero(v_ref)[0])                    # first entry above the threshold
    v_got[j] = False
    assert len(bench.parity_violations("f16", d, want, got, v_ref, v_got, 0.5)) == 1      # identical score, flipped entry: never
    d[j] = 1
    assert bench.parity_violations("f16", d, want, got, v_ref, v_got, float(want[j - 1])) == []
