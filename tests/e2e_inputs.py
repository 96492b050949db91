"""Seeded inputs of the multi-layer, multi-chunk D = 128 parity case (tests/golden/g9_e2e_d128.npz).

Shared by oracle/gen_golden.py (which feeds them to the REFERENCE's KVScore._get_score in the build container and stores the
resulting scores) and by the GPU parity test (which regenerates them on the GPU box from the same seeds): only the scores, the
threshold and the mask travel as fixtures - the inputs are 60 MB per dtype.  The fixture carries a checksum of the inputs so
that a generator mismatch fails loudly instead of comparing different problems.
"""
import torch

GEOM = dict(L=2, H=28, Hkv=4, D=128, sink=32, N=8000, chunk=2000)   # Qwen2.5-7B head geometry, 2 layers x 4 scoring chunks
# round 4 (tests/golden/g10_e2e_d128_512k.npz): 8 layers x 8 scoring chunks = 512 000 scores under ONE global threshold
GEOM_512K = dict(L=8, H=28, Hkv=4, D=128, sink=32, N=16000, chunk=2000)
SEED_512K = 777
# round 4 (tests/golden/g11_e2e_llama.npz): BASELINE config C3's head geometry (Llama-3.1-8B: H32 Hkv8 D128, G = 4), 2 layers x 4 chunks
GEOM_LLAMA = dict(L=2, H=32, Hkv=8, D=128, sink=32, N=8000, chunk=2000)
SEED_LLAMA = 3131


# round 6 (tests/golden/g14_far_context.npz): the LAST TWO scoring chunks of one layer of the headline context (Qwen2.5-7B head geometry,
# N = 131 072 = 65 x 2000 + 1072): chunk starts 128 032 and 130 032, key length 133 k - the far end of the cache, from the reference itself
GEOM_FAR = dict(L=1, H=28, Hkv=4, D=128, sink=32, N=131072, chunk=2000)
SEED_FAR = 1414


def make_far(dtype, geom=GEOM_FAR, seed=SEED_FAR, n_last=2):
    """-> (K0 [1,Hkv,sink+N,D], [(start, end, q_len, q [1,H,q,D], k_rep [1,Hkv,q,D])] for the last ``n_last`` chunks) as CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    H, Hkv, D = geom["H"], geom["Hkv"], geom["D"]
    K0 = torch.randn(1, Hkv, geom["sink"] + geom["N"], D, generator=g).to(dtype)
    out = []
    for (st, en, q_len) in chunks(geom)[-n_last:]:
        q = torch.randn(1, H, q_len, D, generator=g).to(dtype)
        kr = torch.randn(1, Hkv, q_len, D, generator=g).to(dtype)
        out.append((st, en, q_len, q, kr))
    return K0, out


# round 6 (tests/golden/g15_full_layer.npz): ONE FULL LAYER of the headline context - all 66 scoring chunks, 524 288 scores per dtype,
# threshold and mask at ratio 0.3 from the reference; the inputs are streamed chunk by chunk from one generator (1.2 GB per dtype otherwise)
SEED_FULL = 1515


def stream_full(dtype, geom=GEOM_FAR, seed=SEED_FULL):
    """generator: first K0 [1,Hkv,sink+N,D], then (start, end, q_len, q [1,H,q,D], k_rep [1,Hkv,q,D]) for every scoring chunk in order"""
    g = torch.Generator().manual_seed(seed)
    H, Hkv, D = geom["H"], geom["Hkv"], geom["D"]
    yield torch.randn(1, Hkv, geom["sink"] + geom["N"], D, generator=g).to(dtype)
    for (st, en, q_len) in chunks(geom):
        q = torch.randn(1, H, q_len, D, generator=g).to(dtype)
        kr = torch.randn(1, Hkv, q_len, D, generator=g).to(dtype)
        yield st, en, q_len, q, kr


def checksum_update(acc: int, t) -> int:
    """one step of ``checksum`` (streamed inputs)"""
    v = t.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    w = torch.arange(1, v.numel() + 1, dtype=torch.int64) % 1000003
    return (acc * 1000003 + int((v.view(-1) * w).sum().item())) % (1 << 63)


def chunks(geom=GEOM):
    """(start, end, q_len) per scoring chunk exactly as model/wrapper.py:197-221 cuts them (13 / 26 repeat-prompt tokens)."""
    out = []
    for c, st in enumerate(range(0, geom["N"], geom["chunk"])):
        m = min(geom["chunk"], geom["N"] - st)
        out.append((geom["sink"] + st, geom["sink"] + st + m, m + (13 if c == 0 else 26)))
    return out


def make(dtype, geom=GEOM, seed=4242):
    """-> (K0[L] [1,Hkv,sink+N,D], per chunk per layer (q [1,H,q,D], k_rep [1,Hkv,q,D])) as CPU tensors of `dtype`."""
    g = torch.Generator().manual_seed(seed)
    L, H, Hkv, D = geom["L"], geom["H"], geom["Hkv"], geom["D"]
    K0 = [torch.randn(1, Hkv, geom["sink"] + geom["N"], D, generator=g).to(dtype) for _ in range(L)]
    per_chunk = []
    for (st, en, q_len) in chunks(geom):
        per_layer = []
        for _ in range(L):
            q = torch.randn(1, H, q_len, D, generator=g).to(dtype)
            kr = torch.randn(1, Hkv, q_len, D, generator=g).to(dtype)
            per_layer.append((q, kr))
        per_chunk.append(per_layer)
    return K0, per_chunk


def checksum(K0, per_chunk) -> int:
    """order-sensitive 63-bit checksum of the 16-bit patterns of every input tensor"""
    acc = 0
    for t in list(K0) + [x for pl in per_chunk for pair in pl for x in pair]:
        v = t.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
        w = torch.arange(1, v.numel() + 1, dtype=torch.int64) % 1000003
        acc = (acc * 1000003 + int((v.view(-1) * w).sum().item())) % (1 << 63)
    return acc
