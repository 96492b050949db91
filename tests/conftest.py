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


# ---- score parity: fixed structural bounds --------------------------------------------------------------------------------
# The scoring kernels follow the reference's rounding chain (attention/score.py:57-61); what differs from the CPU reference is the
# accumulation order of the fp32 dot products inside the MFMA (and exp / log implementations), so a small fraction of the 16-bit
# scores lands a few steps of the 16-bit grid away.  Round 5: the bounds are FIXED per dtype - they no longer follow the last
# measurement (rounds 2-4 asserted "no worse than 2x what was recorded", which a regression could ratchet):
#   fp16   >= 99.8 %  bit-identical, >= 99.95 % within one step, worst <= 8 steps
#   bf16   >= 99.95 % bit-identical, >= 99.99 % within one step, worst <= 8 steps
# (+ a few elements of slack so that cases of a few hundred scores are not decided by one element).  Why 8 is structural: when the
# accumulation order flips the 16-bit rounding of the WINNING logit x (|x| < 8 for the inputs of every test: one step of x is at
# most 2^-8 in fp16, 2^-5 in bf16) the score exp(x - m - log l) moves by that relative amount = 8 steps of its own grid.  Every
# comparison is still printed and, with KVZ_RECORD_PARITY=1, recorded (gpurun_out/score_parity_measured.json; the records of
# round 4 stay in tests/golden/score_parity_measured.json as documentation: worst case there 0.64 % / 0.022 % / 8).
SCORE_BOUNDS = {  # dtype: (not identical <= a * n + b, beyond one step <= c * n + d, worst)
    torch.float16: (0.002, 8, 0.0005, 2, 8),
    torch.bfloat16: (0.0005, 4, 0.0001, 2, 8),
}


def check_score_parity(case: str, got, want, worst_allowed=None):
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
    a, b, c, e, w = SCORE_BOUNDS[want.dtype]
    w = w if worst_allowed is None else worst_allowed
    assert n_diff <= a * n + b and n_far <= c * n + e and worst <= w, (case, str(want.dtype), n, n_diff, n_far, worst)
    return 1 - n_diff / max(n, 1), 1 - n_far / max(n, 1), worst


def check_uniform_contract(case: str, score, valid, ref_valid, k: int):
    """pair-uniform selection (attention/score.py:104-120) on rows WITH ties.  torch.topk's choice among equal values is
    implementation-defined, so what the build promises - and what this asserts against the REFERENCE's mask - is:
    exactly k kept per (layer, head) row, the same multiset of kept values, and the identical mask wherever a score differs from the
    row's boundary value (the k-th largest): only WHICH of the entries equal to the boundary value are kept may differ."""
    N = score.shape[-1]
    s = score.detach().cpu().float().reshape(-1, N)
    v = valid.detach().cpu().bool().reshape(-1, N)
    r = ref_valid.detach().cpu().bool().reshape(-1, N)
    assert bool((v.sum(-1) == k).all()) and bool((r.sum(-1) == k).all()), (case, "kept per row != k")
    if k == 0:
        return 0
    kth = torch.sort(s, dim=-1, descending=True).values[:, k - 1:k]
    off = s != kth
    assert torch.equal(v & off, r & off), (case, "mask differs away from the boundary value")
    assert torch.equal(torch.sort(torch.where(v, s, torch.full_like(s, -1e30)), dim=-1).values,
                       torch.sort(torch.where(r, s, torch.full_like(s, -1e30)), dim=-1).values), (case, "kept multiset differs")
    moved = int((v != r).sum())
    print(f"\nUNIFORM {case}: k = {k}, rows {s.shape[0]}, rows with a tie at the boundary {int(((s == kth).sum(-1) > 1).sum())}, "
          f"entries placed differently among equals {moved}")
    return moved


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
