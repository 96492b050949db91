#!/usr/bin/env python
"""A/B driver for the scoring kernels on the GPU box: for every library variant (tools/ab/lib_*.so or the in-tree build)
   * parity against the CPU oracle at the headline shape and a small multi-tile D = 64 shape (computed once, cached in /tmp),
   * run-to-run bit identity,
   * kernel times (hipEvent brackets of the library) and wall time per call at the bench geometry, fp16 and bf16.
   python tools/ab_score.py [lib.so ...]        (no argument: all of tools/ab + the in-tree library)"""
import ctypes as C, glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

CASES = {  # name: (H, Hkv, D, sink, m, q_len, start_off, dtype)
    "headline_f16": (28, 4, 128, 32, 2000, 2026, 0, torch.float16),
    "d64_f16": (14, 2, 64, 30, 700, 713, 300, torch.float16),
    "d128_bf16": (8, 2, 128, 16, 600, 626, 128, torch.bfloat16),
    # large logits / spikes / everything far below zero: the cold paths (reference moves up, reference moves down)
    "spiky_f16": (8, 2, 128, 16, 700, 713, 128, torch.float16),
    "negative_f16": (8, 2, 128, 16, 700, 713, 128, torch.float16),
    "spiky_bf16": (8, 2, 128, 16, 700, 713, 128, torch.bfloat16),
}

def inputs(name):
    H, Hkv, D, sink, m, q_len, off, dt = CASES[name]
    g = torch.Generator().manual_seed(7)
    q = torch.randn(1, H, q_len, D, generator=g)
    k = torch.randn(1, Hkv, sink + off + m + q_len + 64, D, generator=g)
    if name.startswith("spiky"):      # logits with std ~6 plus a few keys that every query loves (+40) at scattered positions
        q, k = q * 2.5, k * 2.5
        for pos in (3, sink + off + 17, sink + off + 400, k.shape[2] - 70):
            k[:, :, pos] = q[:, ::H // Hkv, min(pos, q_len - 1)] * 0.35
    if name.startswith("negative"):   # every logit far below zero: k = -(mean direction of q) scaled
        u = torch.ones(D) / D ** 0.5
        q = q * 0.3 + 22.0 * u
        k = k * 0.3 - 22.0 * u
    q, k = q.to(dt), k.to(dt)
    return q, k, sink, sink + off, sink + off + m

def oracle_cache():
    import kvzip_oracle as orc
    out = {}
    for name in CASES:
        path = f"/tmp/ab_oracle_{name}.pt"
        if not os.path.exists(path):
            q, k, sink, st, en = inputs(name)
            torch.save(orc.get_score(q, k, sink, st, en), path)
        out[name] = path
    return out

def ulp_keys(t):
    x = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.where(x >= 0x8000, 0x8000 - (x - 0x8000) - 1, x + 0x8000)

def child():
    from kvzip_amd import _lib, ops
    lib = _lib.load(); dev = "cuda:0"
    res = {"lib": os.path.basename(_lib.LIB_PATH)}
    for name in CASES:
        q, k, sink, st, en = inputs(name)
        want = torch.load(f"/tmp/ab_oracle_{name}.pt")
        qd, kd = q.to(dev), k.to(dev)
        got = ops.score_chunk(qd, kd, sink, st, en).cpu()
        again = [ops.score_chunk(qd, kd, sink, st, en).cpu() for _ in range(3)]
        d = (ulp_keys(got) - ulp_keys(want)).abs()
        res[name] = {"exact": round(float((d == 0).float().mean()), 5), "le1": round(float((d <= 1).float().mean()), 5), "worst": int(d.max()),
                     "deterministic": all(torch.equal(got.view(torch.int16), a.view(torch.int16)) for a in again),
                     "nan": bool(torch.isnan(got.float()).any())}
    Hkv, G, m, D, sink, N = 4, 7, 2000, 128, 32, 131072
    q_len = m + 26; klen = sink + N + q_len
    g = torch.Generator(device=dev).manual_seed(0)
    for dt, tag, D in ((torch.float16, "f16", 128), (torch.bfloat16, "bf16", 128), (torch.float16, "f16_d64", 64)):
        q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).to(dt); k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        start = sink + 60000
        for _ in range(5): ops.score_chunk(q, k, sink, start, start + m)
        torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(60): ops.score_chunk(q, k, sink, start, start + m)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 60 * 1e6
        lib.kvz_prof_enable(0)
        r = {"wall_us": round(wall, 1)}
        for kn in ("score_rowstat", "score_colmax"):
            t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(kn.encode(), C.byref(t), C.byref(c))
            r[kn + "_us"] = round(t.value / max(c.value, 1) * 1e3, 1)
        # without the brackets (back-to-back launches on one stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): ops.score_chunk(q, k, sink, start, start + m)
        torch.cuda.synchronize(); r["wall_nobracket_us"] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
        res["time_" + tag] = r
    print("AB " + json.dumps(res))

if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child(); sys.exit(0)
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tools/ab/lib_*.so"))) + [os.path.join(ROOT, "kvzip_amd/libkvzip_hip.so")]
    oracle_cache()
    for rnd in range(2):   # two interleaved rounds: box-to-box and warm-up effects show up as differences between the rounds
        for lib in libs:
            env = dict(os.environ, AB_CHILD="1", KVZIP_HIP_LIB=lib)
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=240)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
            print(f"round {rnd} {os.path.basename(lib)}: " + (line[0][3:] if line else "FAILED rc=%d %s" % (p.returncode, p.stderr[-800:])), flush=True)
