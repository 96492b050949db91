#!/usr/bin/env python
"""Kernel times of library variants (tools/ab/lib_*.so) at the bench geometry - for the time-attribution builds (KVZ_ABL), whose
results are garbage by construction.  python tools/abl_time.py [lib.so ...]"""
import ctypes as C, glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def child():
    import torch
    from kvzip_amd import _lib, ops
    lib = _lib.load(); dev = "cuda:0"
    Hkv, G, m, D, sink, N = 4, 7, 2000, 128, 32, 131072
    q_len = m + 26; klen = sink + N + q_len
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).half(); k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
    start = sink + 60000
    for _ in range(5): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
    for _ in range(40): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_enable(0)
    r = {}
    for kn in ("score_rowstat", "score_colmax"):
        t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(kn.encode(), C.byref(t), C.byref(c))
        r[kn] = round(t.value / max(c.value, 1) * 1e3, 1)
    print("ABL " + json.dumps(r))
if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child(); sys.exit(0)
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tools/ab/lib_*.so")))
    for rnd in range(2):
        for lib in libs:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, AB_CHILD="1", KVZIP_HIP_LIB=lib), capture_output=True, text=True, timeout=200)
            line = [l for l in p.stdout.splitlines() if l.startswith("ABL ")]
            print(f"round {rnd} {os.path.basename(lib):24s} " + (line[0][4:] if line else "FAILED " + p.stderr[-300:]), flush=True)
