#!/usr/bin/env python
"""40 scoring calls at the headline shape with the pruning path on (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import _lib, ops
lib = _lib.load(); dev = "cuda:0"
Hkv, G, m, D, sink, N = 4, 7, 2000, 128, 32, 131072
q_len = m + 26; klen = sink + N + q_len
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).half(); k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).half()
start = sink + 60000
st = torch.cuda.current_stream().cuda_stream
need = lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink)
ws = torch.empty(need, dtype=torch.uint8, device=dev); log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
lib.kvz_debug_set_tunable(b"score_prune", int(os.environ.get("PRUNE", "3")))
for _ in range(45):
    ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), st), "fill")
    ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, start + m, q_len, Hkv, G, D,
                                      ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), st), "score_chunk_log")
torch.cuda.synchronize()
