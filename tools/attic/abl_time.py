#!/usr/bin/env python
"""Times of the scoring kernels for a list of library builds (tools/ab/lib_abl_*.so = time-attribution builds, KVZ_ABL masks):
hipEvent brackets of the library at the headline shape, fp16, interleaved rounds.  python tools/abl_time.py [rounds] [lib ...]"""
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
    if os.environ.get("ABL_CHILD"):
        child(); sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    libs = sys.argv[2:] or [os.path.join(ROOT, "kvzip_amd/libkvzip_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "tools/ab/lib_abl_*.so")), key=lambda p: int(p.split("_")[-1][:-3]))
    for rnd in range(rounds):
        for lib in libs:
            env = dict(os.environ, ABL_CHILD="1", KVZIP_HIP_LIB=lib)
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=120)
            line = [l for l in p.stdout.splitlines() if l.startswith("ABL ")]
            print(f"round {rnd} {os.path.basename(lib):24s} " + (line[0][4:] if line else "FAILED rc=%d %s" % (p.returncode, p.stderr[-300:])), flush=True)
