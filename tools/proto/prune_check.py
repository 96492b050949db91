#!/usr/bin/env python
"""PROTOTYPE ONLY - needs tools/proto/score_prune_proto.patch applied to kvzip_amd/csrc (git apply) and the library rebuilt.
Exact pruning of pass B (knob score_prune): results and times of the three variants at the headline shape and a few odd ones.
0 = two full passes (product), 1 = key-per-lane pass A + full pass B, 2 = + bounds + pruned pass B.  1 and 2 must agree bit for bit."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import _lib, ops

def times(lib, fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
    for _ in range(n): fn()
    torch.cuda.synchronize(); lib.kvz_prof_enable(0)
    r = {}
    for kn in ("score_rowstat", "score_bounds", "score_colmax"):
        t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(kn.encode(), C.byref(t), C.byref(c))
        if c.value: r[kn] = round(t.value / c.value * 1e3, 1)
    return r

def main():
    lib = _lib.load(); dev = "cuda:0"
    shapes = [(4, 7, 2000, 128, 32, 2026, 60000, "gauss"), (4, 7, 2000, 128, 32, 2026, 60000, "copy"), (2, 4, 777, 128, 4, 790, 1000, "gauss"),
              (8, 4, 2000, 128, 32, 2026, 3000, "gauss"), (1, 1, 33, 128, 0, 40, 5, "gauss"), (2, 2, 300, 64, 16, 310, 100, "gauss")]
    for (Hkv, G, m, D, sink, q_len, s0, kind) in shapes:
        N = s0 + m + 1000
        klen = sink + N + q_len
        g = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).half()
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
        start = sink + s0
        if kind == "copy":   # repeat-prompt-like: the queries of position i resemble the key of ctx position i
            kk = k[:, :, start:start + m].repeat_interleave(G, dim=1)
            q[:, :, :m] = (q[:, :, :m] * 0.5 + kk * 1.5).half()
        out = {}
        for pr in (0, 1, 2):
            lib.kvz_debug_set_tunable(b"score_prune", pr)
            out[pr] = ops.score_chunk(q, k, sink, start, start + m).float().clone()
            t = times(lib, lambda: ops.score_chunk(q, k, sink, start, start + m)) if m >= 777 else {}
            print(f"  prune={pr} {json.dumps(t)}")
        lib.kvz_debug_set_tunable(b"score_prune", 0)
        d01 = (out[0] - out[1]).abs()
        print(f"shape Hkv{Hkv} G{G} m{m} D{D} sink{sink} q{q_len} {kind}: 1==2 bitwise {bool((out[1] == out[2]).all())}  "
              f"n_diff(1,2) {int((out[1] != out[2]).sum())}  |0-1| max {float(d01.max()):.3e} n_diff {int((d01 > 0).sum())} of {d01.numel()}  nan {int(out[2].isnan().sum())}", flush=True)

if __name__ == "__main__":
    main()
