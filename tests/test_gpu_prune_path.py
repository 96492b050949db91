"""The exact-pruning variant of the scoring call (round 5; round 6: both dtypes, candidates at key granularity; knob ``score_prune``: 3 = the chained call (6, the default, pipelines its three small launches over the calls of a stream: tests/test_gpu_tail_pipeline.py),
0 = the two-pass call, 5 = round 5's candidate pairs): a key-per-lane pass A that also writes per-group maxima, merged statistics + group bounds, per
row group the list of ctx keys whose column maximum the group can hold, a pass that recomputes 32 gathered candidate keys per MFMA tile.

What is asserted: against the CPU oracle the same fixed bounds as the product path (``conftest.SCORE_BOUNDS``); the sparse pass B returns
THE SAME BITS as the full pass B on the same statistics (knob 1 vs 3) and as the sparse pass over every pair (knob 4) - the bounds are
exact; NaN inputs poison exactly the heads the product path poisons; logits beyond the reach of the fixed references are redone in-kernel."""
import pytest
import torch

import kvzip_oracle as orc
from conftest import check_score_parity

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _score_log(q, k, sink, start, end, prune):
    """kvz_score_chunk_log + kvz_score_finalize_log with the knob set for the duration of the call."""
    from kvzip_amd import _lib, ops
    lib = _lib.load()
    _, H, q_len, D = q.shape
    _, Hkv, klen, _ = k.shape
    G, m = H // Hkv, end - start
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink), dtype=torch.uint8, device=q.device)
    log = torch.empty(Hkv, m, dtype=torch.int32, device=q.device)
    out = torch.empty(Hkv, m, dtype=q.dtype, device=q.device)
    lib.kvz_debug_set_tunable(b"score_prune", prune)
    try:
        ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), st), "kvz_score_log_fill")
        ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, end, q_len, Hkv, G, D,
                                          ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), st), "kvz_score_chunk_log")
        ops.check(lib.kvz_score_finalize_log(log.data_ptr(), log.numel(), out.data_ptr(), ops._dtype_code(q.dtype), st), "kvz_score_finalize_log")
        torch.cuda.synchronize()
    finally:
        lib.kvz_debug_set_tunable(b"score_prune", -1)   # (back to the default)
    return out.cpu()


def _same_bits(a, b):
    return torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0))


def _inputs(H, Hkv, D, sink, N, q_len, seed, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    klen = sink + N + q_len
    return torch.randn(1, H, q_len, D, generator=g).to(dtype), torch.randn(1, Hkv, klen, D, generator=g).to(dtype)


@pytest.mark.parametrize("H,Hkv,D,sink,N,start,m,q_len", [
    (28, 4, 128, 32, 2100, 32, 2000, 2013),        # first chunk of the headline geometry (q = m + 13)
    (14, 2, 128, 32, 5000, 32 + 2000, 2000, 2026),  # a later chunk (q = m + 26), rows of a group in two query heads
    (8, 2, 128, 4, 1800, 4 + 1000, 777, 790),       # sink not a multiple of 32: a 32-key block of the stream straddles two ctx blocks
    (8, 8, 128, 0, 700, 5, 33, 40),                 # G = 1, two ctx key blocks, one row tile
    (4, 2, 64, 16, 500, 16 + 100, 300, 310),        # D = 64
    (16, 2, 64, 30, 1200, 30 + 400, 513, 77),       # G = 8, rows spanning several query heads per tile
])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_pruned_scoring_call(H, Hkv, D, sink, N, start, m, q_len, dtype):
    q, k = _inputs(H, Hkv, D, sink, N, q_len, seed=H * 1000 + m, dtype=dtype)
    want = orc.get_score(q, k, sink, start, start + m)[0]
    qd, kd = q.to(DEV), k.to(DEV)
    got = {p: _score_log(qd, kd, sink, start, start + m, p) for p in (1, 3, 4, 5)}
    check_score_parity(f"prune/{H}x{Hkv}x{D}/m{m}/{dtype}", got[3], want)
    assert _same_bits(got[3], got[1]), "candidate-key pass differs from the full pass B on the same statistics"
    assert _same_bits(got[3], got[4]), "candidate keys differ from all (group, key) items"
    assert _same_bits(got[3], got[5]), "candidate keys (round 6) differ from candidate pairs (round 5)"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_pruned_scoring_call_on_copy_like_logits(dtype):
    """peaky rows (the repeat prompt's queries resemble the keys they repeat): few candidate pairs, large spread of the row statistics."""
    H, Hkv, D, sink, N, start, m, q_len = 14, 2, 128, 32, 3000, 32 + 500, 2000, 2026
    q, k = _inputs(H, Hkv, D, sink, N, q_len, seed=77, dtype=dtype)
    kk = k[:, :, start:start + m].repeat_interleave(H // Hkv, dim=1)
    q[:, :, :m] = (q[:, :, :m].float() * 0.5 + kk.float() * 0.35).to(dtype)
    want = orc.get_score(q, k, sink, start, start + m)[0]
    qd, kd = q.to(DEV), k.to(DEV)
    got1, got3 = _score_log(qd, kd, sink, start, start + m, 1), _score_log(qd, kd, sink, start, start + m, 3)
    check_score_parity("prune/copy-like", got3, want)
    assert _same_bits(got3, got1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_pruned_scoring_call_propagates_nan_like_the_product_path(dtype):
    H, Hkv, D, sink, N, start, m, q_len = 8, 4, 128, 32, 1500, 32 + 300, 600, 610
    q, k = _inputs(H, Hkv, D, sink, N, q_len, seed=5, dtype=dtype)
    q[0, 3, 77, 5] = float("nan")        # query head 3 -> KV head 1
    k[0, 2, start + 11, 7] = float("inf")  # a ctx key of KV head 2
    qd, kd = q.to(DEV), k.to(DEV)
    ref, got = _score_log(qd, kd, sink, start, start + m, 0), _score_log(qd, kd, sink, start, start + m, 3)
    assert torch.equal(ref.isnan(), got.isnan())
    assert ref[1].isnan().all() and ref[2].isnan().all() and not ref[0].isnan().any() and not ref[3].isnan().any()
    want = orc.get_score(q, k, sink, start, start + m)[0]
    check_score_parity("prune/nan/clean heads", got[[0, 3]], want[[0, 3]])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_logits_outside_the_range_of_the_fixed_references_are_redone(dtype):
    """the key-per-lane pass keeps the reference of a row where its first key block put it; a later logit ~88 above it makes the row's sum
    non-finite, and the kernel redoes such an item at its end with references that follow the logits: same bounds against the oracle, and
    the sparse pass still returns the bits of the full pass B on those statistics."""
    H, Hkv, D, sink, N, start, m, q_len = 8, 4, 128, 32, 1500, 32 + 300, 600, 610
    q, k = _inputs(H, Hkv, D, sink, N, q_len, seed=9, dtype=dtype)
    k[0, 1, start + 400] *= 40.0
    k[0, 2, -100] *= 60.0
    k[0, 3, 5] *= 50.0
    want = orc.get_score(q, k, sink, start, start + m)[0]
    assert not want.isnan().any()
    qd, kd = q.to(DEV), k.to(DEV)
    got1, got3 = _score_log(qd, kd, sink, start, start + m, 1), _score_log(qd, kd, sink, start, start + m, 3)
    check_score_parity("prune/spikes", got3, want)
    assert _same_bits(got3, got1)
    # ... and the next call with ordinary inputs is right as well (the per-block redo masks clear themselves)
    q2, k2 = _inputs(H, Hkv, D, sink, N, q_len, seed=10, dtype=dtype)
    want2 = orc.get_score(q2, k2, sink, start, start + m)[0]
    check_score_parity("prune/after a redo", _score_log(q2.to(DEV), k2.to(DEV), sink, start, start + m, 3), want2)


@pytest.mark.parametrize("dtype,q_len", [(torch.float16, 20), (torch.bfloat16, 20)])
def test_knob_is_ignored_where_the_path_does_not_apply(dtype, q_len):
    """chunks of fewer than 32 query positions take the two-pass kernels whatever the knob says."""
    H, Hkv, D, sink, N, start, m = 4, 2, 128, 8, 900, 8 + 100, 300
    q, k = _inputs(H, Hkv, D, sink, N, q_len, seed=3, dtype=dtype)
    qd, kd = q.to(DEV), k.to(DEV)
    assert torch.equal(_score_log(qd, kd, sink, start, start + m, 0), _score_log(qd, kd, sink, start, start + m, 3))


def test_reference_fixtures_end_to_end_with_the_pruning_path_on():
    """the reference-generated end-to-end fixtures (scores -> threshold -> mask of 64 000 / 512 000 / 128 000 scores, both dtypes) through the whole
    cache object with the knob preset by the environment (a fresh process: the knob is read when the library is loaded)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVZIP_SCORE_PRUNE="3")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_e2e_parity.py"), "-x", "-q", "-m", "gpu"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-1000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-500:]
