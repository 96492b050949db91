"""GPU: BASELINE config C2 through the front door - a random-init Qwen2 at the Qwen2.5-7B geometry (L28 H28 Hkv4 D128, bf16),
ModelKVzip.prefill(32 768 ids) -> scoring -> kv.prune(0.3) -> generate(query, kv=kv)  (reference model/wrapper.py:169-195,
:223-249, :251-284; README.md:43-57).  The driver lives in tools/e2e_c2.py (it also writes profiles/r3_e2e_c2.json)."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c2_qwen7b_geometry_32k_through_modelkvzip():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_c2
    out = e2e_c2.run(ctx_len=32768, layers=28, verbose=False)
    print("\nC2 e2e:", {k: out[k] for k in ("evict", "sampled_call_vs_oracle", "peak_hbm_gb", "evict_equals_retain_tokens")})
    assert out["evict_equals_retain_scores"], "both cache types run the same scoring kernels on the same forward pass"
    assert out["evict_equals_retain_tokens"], "compact + varlen attention must generate what mask-gather attention generates"
    assert out["last_logit_max_abs_diff_rel"] <= 2e-2
    assert abs(out["evict"]["real_ratio"] - 0.3) < 5e-3 and out["evict"]["thres"] == out["retain"]["thres"]
    s = out["sampled_call_vs_oracle"]
    # (bf16, the workload's own (layer, chunk) shape; same bounds as the headline-shape parity test)
    assert s["bit_identical"] >= 0.999 and s["within_one_half_ulp"] >= 0.9999 and s["worst_half_ulps"] <= 2
    assert out["evict"]["kv_gb_after_prune"] < 0.35 * 2 * 28 * 4 * 32800 * 128 * 2 / 1e9 + 0.3
