"""CPU: the C restatement (oracle/oracle_int.c) agrees with the golden vectors and with the Python oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import kvzip_oracle as orc
from conftest import ROOT, from_bits, load_golden, to_bits

LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_int.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return C.CDLL(LIB)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_c_threshold_golden(lib):
    g = load_golden("g2_threshold.npz")
    for tag, bf in (("bf16", 1), ("f16", 0), ("odd", 0)):
        score = np.ascontiguousarray(g[f"{tag}/score"]).reshape(-1)
        for key in [k for k in g.files if k.startswith(f"{tag}/thres/")]:
            r = key.split("/")[-1]
            valid = np.zeros(score.size, dtype=np.uint8)
            thres = C.c_double(0)
            rc = lib.orc_threshold(_p(score), C.c_int64(score.size), C.c_double(float(r)), bf, _p(valid), C.byref(thres))
            assert rc == 0
            want = np.unpackbits(g[f"{tag}/valid/{r}"])[:score.size]
            assert np.array_equal(valid, want), (tag, r)
            assert thres.value == g[f"{tag}/thres/{r}"][0]


@pytest.mark.parametrize("tag", ["f16_pair", "bf16_pair", "f16_uniform"])
def test_c_compact_and_append_golden(lib, tag):
    g = load_golden(f"g6_cache_{tag}.npz")
    L, H, Hkv, D, sink, N, bf = g["meta"].tolist()
    klen, rb = sink + N, D * 2
    lens, cus, flats = [], [], []
    for l in range(L):
        k = np.ascontiguousarray(g[f"K0/{l}"]).reshape(Hkv, klen, D)
        v = np.ascontiguousarray(g[f"V0/{l}"]).reshape(Hkv, klen, D)
        valid = np.ascontiguousarray(g["valid"][l, 0]).astype(np.uint8)
        ko, vo = np.zeros((Hkv * klen, D), np.uint16), np.zeros((Hkv * klen, D), np.uint16)
        len_k, cu, mx = np.zeros(Hkv, np.int32), np.zeros(Hkv + 1, np.int32), C.c_int32(0)
        lib.orc_compact.restype = C.c_int64
        rows = lib.orc_compact(_p(k), _p(v), _p(valid), Hkv, N, sink, klen, rb, _p(ko), _p(vo), _p(len_k), _p(cu), C.byref(mx))
        assert np.array_equal(ko[:rows], g[f"flatK/{l}"]) and np.array_equal(vo[:rows], g[f"flatV/{l}"])
        assert np.array_equal(len_k, g[f"len_k/{l}"]) and np.array_equal(cu, g[f"cu_len_k/{l}"])
        assert mx.value == int(g[f"max_len_k/{l}"][0])
        lens.append(len_k); cus.append(cu); flats.append(ko[:rows].copy())
    # first append of the generation phase (t = 7)
    for l in range(L):
        state = np.ascontiguousarray(g[f"gen/0/{l}/k"]).reshape(Hkv * 7, D)
        out = np.zeros((flats[l].shape[0] + Hkv * 7, D), np.uint16)
        lib.orc_update_flatten(_p(flats[l]), _p(state), _p(lens[l]), _p(cus[l]), Hkv, 7, rb, _p(out))
        assert np.array_equal(out, g[f"gen/0/{l}/k_out"])


def test_c_full_mask(lib):
    valid = np.array([1, 0, 0, 1, 1], dtype=np.uint8)
    full = np.zeros(10, dtype=np.uint8)
    lib.orc_full_mask(_p(valid), 2, 5, 10, _p(full))
    assert full.tolist() == [1, 1, 1, 0, 0, 1, 1, 1, 1, 1]
    want = orc.get_valid(torch.from_numpy(valid.astype(bool)).view(1, 1, 5), 2, 10)
    assert want.view(-1).tolist() == [bool(x) for x in full]
