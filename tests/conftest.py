import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when collected on a machine without a GPU and without -m gpu
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def from_bits(arr, is_bf16):
    """uint16 bit patterns -> torch half tensor"""
    t = torch.from_numpy(arr.astype(np.int16) if arr.dtype == np.uint16 else arr)
    return t.view(torch.bfloat16 if is_bf16 else torch.float16)


def to_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def ulp_diff(a, b):
    """|a-b| in units of 16-bit ulps via the monotone integer order of the bit patterns"""
    def key(t):
        x = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(x >= 0x8000, 0x8000 - (x - 0x8000) - 1, x + 0x8000)
    return (key(a) - key(b)).abs()


# ---- score parity bookkeeping -------------------------------------------------------------------------------------------
# The scoring kernels follow the reference's rounding chain; what differs from the CPU oracle is the accumulation order of the
# fp32 dot products (and exp / log implementations), so a small fraction of the 16-bit scores lands one or a few steps of the
# 16-bit grid away.  Every GPU test that compares scores goes through check_score_parity: it prints the measured distribution,
# records it (KVZ_RECORD_PARITY=1 -> gpurun_out/score_parity_measured.json) and bounds it at TWICE the values measured on MI355X
# and committed in tests/golden/score_parity_measured.json (+2 elements of slack for the tiny cases); a case without a record
# falls back to the structural bound (>= 97 % identical, >= 99.5 % within one step, worst 16 = one ulp of the winning logit).
_MEASURED = None


def _measured():
    global _MEASURED
    if _MEASURED is None:
        import json
        path = os.path.join(GOLDEN, "score_parity_measured.json")
        _MEASURED = json.load(open(path)) if os.path.exists(path) else {}
    return _MEASURED


def check_score_parity(case: str, got, want):
    import json
    d = ulp_diff(got, want)
    n = d.numel()
    n_diff, n_far, worst = int((d != 0).sum()), int((d > 1).sum()), int(d.max()) if n else 0
    print(f"\nPARITY {case}: {n} scores, {1 - n_diff / max(n, 1):.5f} bit-identical, {1 - n_far / max(n, 1):.5f} within one step, worst {worst}")
    if os.environ.get("KVZ_RECORD_PARITY"):
        path = os.path.join(ROOT, "gpurun_out", "score_parity_measured.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec[case] = {"n": n, "not_identical": n_diff, "beyond_one_step": n_far, "worst": worst}
        json.dump(rec, open(path, "w"), indent=0, sort_keys=True)
    m = _measured().get(case)
    if m is None:
        assert n_diff <= 0.03 * n + 2 and n_far <= 0.005 * n + 1 and worst <= 16, (case, n, n_diff, n_far, worst)
    else:
        assert n_diff <= 2 * m["not_identical"] + 2 and n_far <= 2 * m["beyond_one_step"] + 1 and worst <= max(2 * m["worst"], 2), \
            (case, n, n_diff, n_far, worst, m)
    return 1 - n_diff / max(n, 1), 1 - n_far / max(n, 1), worst


def grid_step(x: torch.Tensor, dtype) -> torch.Tensor:
    """One step of the 16-bit grid of `dtype` at magnitude |x| (fp32 tensor): 2^(floor(log2 |x|) - mantissa bits), the smallest
    normal binade for tiny values."""
    mant, emin = (10, -14) if dtype == torch.float16 else (7, -126)
    e = torch.floor(torch.log2(x.abs().clamp_min(2.0 ** emin)))
    return torch.pow(2.0, e - mant)


def check_attn(case: str, got, want, tol: float, rel: float = 0.0, ulp_of=None):
    """Attention outputs vs the oracle / an fp32 reference (a13: flash-attn itself is not in the image, so this boundary is anchored,
    not pinned): print the ACHIEVED maximum error next to the bound and record it with KVZ_RECORD_PARITY=1
    (gpurun_out/attn_error_measured.json).  The bound is north_star's absolute tolerance (1e-3 in fp16) wherever one step of the
    16-bit output grid is finer than that - every |output| < 2 in fp16, i.e. every decode case - and, with ``ulp_of=dtype`` (round 4,
    replaces the blanket relative term 2^-10 x |want| of round 3), exactly ONE step of that grid where the grid itself is coarser
    (rows of the multi-row kernels that see a handful of keys return values of magnitude 2-4, where fp16 steps by 1.95e-3): an
    error of two output steps fails."""
    import json
    g, w = got.detach().float().cpu(), want.detach().float().cpu()
    err = (g - w).abs()
    bound = tol + rel * w.abs()
    if ulp_of is not None:
        bound = torch.maximum(torch.full_like(w, tol), grid_step(torch.maximum(w.abs(), g.abs()), ulp_of) * (1 + 2.0 ** -12))
    worst = float(err.max()) if err.numel() else 0.0
    margin = float((err - bound).max()) if err.numel() else 0.0
    coarse = int((bound > tol).sum()) if ulp_of is not None else 0
    print(f"\nATTN {case}: max |err| {worst:.3e} (absolute bound {tol:g}" + (f" + {rel:g} x |want|" if rel else "")
          + (f"; one output step where the {ulp_of} grid is coarser: {coarse} of {err.numel()} values" if ulp_of is not None else "")
          + f"), worst margin {margin:.2e}")
    if os.environ.get("KVZ_RECORD_PARITY"):
        path = os.path.join(ROOT, "gpurun_out", "attn_error_measured.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec[case] = {"max_abs_err": worst, "tol": tol, "rel": rel, "one_step_bound": ulp_of is not None, "values_on_coarser_grid": coarse,
                     "n": int(err.numel()), "max_abs_want": float(w.abs().max()) if w.numel() else 0.0, "worst_margin": margin}
        json.dump(rec, open(path, "w"), indent=0, sort_keys=True)
    assert (err <= bound).all(), (case, worst)


def check_mask_flips(case: str, got_scores, want_scores, got_valid, want_valid, want_thres: float, allowed: int = 0):
    """End-to-end eviction mask vs the mask of the reference's / oracle's scores.  The mask is an integer function of the scores, so an
    entry may only flip where a score that is NOT bit-identical sits right at the threshold (the strict > evicts ties,
    attention/score.py:95-96): every flipped entry must be such a score, at most the worst score deviation away from the threshold -
    nothing else may ever flip - and their number must not exceed `allowed` (what was measured on MI355X: 0 in every small case).
    Replaces the Hamming fractions (1e-3 .. 1e-4) of rounds 1-3."""
    d = ulp_diff(got_scores, want_scores).reshape(-1)
    flips = (got_valid.reshape(-1).cpu() != want_valid.reshape(-1).cpu())
    ham = int(flips.sum())
    tb = torch.tensor([want_thres]).to(want_scores.dtype)
    near = (d > 0) & (ulp_diff(want_scores.reshape(-1), tb.expand(d.numel())) <= max(int(d.max()), 1))
    print(f"\nMASK {case}: Hamming distance {ham} of {d.numel()} (allowed {allowed}); non-identical scores at the threshold: {int(near.sum())}")
    assert int((flips & ~near).sum()) == 0, (case, "a mask entry flipped although its score equals the expected one or lies away from the threshold")
    assert ham <= allowed, (case, ham)
    return ham
