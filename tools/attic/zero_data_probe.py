import ctypes as C, os, sys
import torch
sys.path.insert(0, os.environ.get("R", "."))
from kvzip_amd import _lib, ops
lib = _lib.load(); dev = "cuda:0"
def run(kind):
    Hkv, G, m, D, sink, N = 4, 7, 2000, 128, 32, 131072
    q_len = m + 26; klen = sink + N + q_len
    g = torch.Generator(device=dev).manual_seed(0)
    if kind == "zeros":
        q = torch.zeros(1, Hkv * G, q_len, D, device=dev).half(); k = torch.zeros(1, Hkv, klen, D, device=dev).half()
    else:
        q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).half(); k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
    start = sink + 60000
    for _ in range(5): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
    for _ in range(50): ops.score_chunk(q, k, sink, start, start + m)
    torch.cuda.synchronize(); lib.kvz_prof_enable(0)
    out = []
    for name in ("score_rowstat", "score_colmax"):
        t, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(name.encode(), C.byref(t), C.byref(c)); out.append(t.value / c.value * 1e3)
    print(kind, "rowstat %.1f us  colmax %.1f us" % tuple(out))
run("random"); run("zeros"); run("random")
