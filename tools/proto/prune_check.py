#!/usr/bin/env python
"""Round-5 pruned scoring call (knob score_prune: 3 = fp16 default, 0 = two-pass call): checks and times of its variants.
Exact pruning of pass B (knob score_prune) through the deferred-log entry points: results and times of the variants.
0 = two full passes (product), 1 = key-per-lane pass A + full pass B, 2 = + bounds + pruned dense pass B, 3 = + compacted candidate list +
queue-style sparse pass B.  1, 2 and 3 must agree bit for bit."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from kvzip_amd import _lib, ops

def main():
    lib = _lib.load(); dev = "cuda:0"
    dt = torch.bfloat16 if os.environ.get("PRUNE_DTYPE") == "bf16" else torch.float16
    variants = [int(x) for x in os.environ.get("PRUNE_VARIANTS", "0,1,2,3").split(",")]
    shapes = [(4, 7, 2000, 128, 32, 2026, 60000, "gauss"), (4, 7, 2000, 128, 32, 2026, 60000, "copy"), (4, 7, 2000, 128, 32, 2013, 0, "gauss"),
              (2, 4, 777, 128, 4, 790, 1000, "gauss"), (8, 4, 2000, 128, 32, 2026, 3000, "gauss"), (1, 1, 33, 128, 0, 40, 5, "gauss"),
              (2, 2, 300, 64, 16, 310, 100, "gauss"), (4, 7, 2000, 128, 32, 2026, 60000, "nan"), (4, 7, 2000, 128, 32, 2026, 60000, "const"),
              (4, 7, 2000, 128, 32, 2026, 60000, "spike"), (4, 7, 2000, 128, 32, 2026, 60000, "neg")]
    st = torch.cuda.current_stream().cuda_stream
    for (Hkv, G, m, D, sink, q_len, s0, kind) in shapes:
        N = s0 + m + 1000
        klen = sink + N + q_len
        g = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(1, Hkv * G, q_len, D, generator=g, device=dev).to(dt)
        k = torch.randn(1, Hkv, klen, D, generator=g, device=dev).to(dt)
        start = sink + s0
        if kind == "copy":   # repeat-prompt-like: the queries of position i resemble the key of ctx position i
            kk = k[:, :, start:start + m].repeat_interleave(G, dim=1)
            q[:, :, :m] = (q[:, :, :m] * 0.5 + kk * 1.5).to(dt)
        if kind == "nan":
            q[0, 3, 77, 5] = float("nan")      # poisons KV head 0 only
        if kind == "spike":                    # a few keys with logits far above everything before them: the fallback must take over
            k[0, 1, start + 700] *= 40.0; k[0, 2, klen - 100] *= 60.0
        if kind == "neg":                      # all logits very negative except late ones
            k[0, :, :sink + N] = -k[0, :, :1].abs() * 0 + k[0, :, :sink + N]; q[:] = q.abs(); k[0, :, : start + m] = -k[0, :, : start + m].abs() * 3.0
        if kind == "const":
            q[:] = 0.25; k[:] = 0.5            # every logit equal: every pair is a candidate
        need = lib.kvz_score_workspace_bytes(Hkv, G, q_len, m, sink)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        log = torch.empty(Hkv, m, dtype=torch.int32, device=dev)
        out = {}
        def call():
            ops.check(lib.kvz_score_log_fill(log.data_ptr(), log.numel(), st), "fill")
            ops.check(lib.kvz_score_chunk_log(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), klen, sink, start, start + m, q_len, Hkv, G, D,
                                              ops._dtype_code(q.dtype), log.data_ptr(), m, ws.data_ptr(), ws.numel(), st), "score_chunk_log")
        for pr in variants:
            if os.environ.get("PRUNE_FORCE_REDO"): ws.fill_(255)   # every block of the key-per-lane pass redoes its items in the slow loop
            lib.kvz_debug_set_tunable(b"score_prune", pr)
            call()
            o = torch.empty(Hkv, m, dtype=dt, device=dev)
            ops.check(lib.kvz_score_finalize_log(log.data_ptr(), log.numel(), o.data_ptr(), ops._dtype_code(q.dtype), st), "finalize")
            out[pr] = o.float().clone()
            t = {}
            if m >= 777 and not os.environ.get("PRUNE_NOTIME"):
                for _ in range(3): call()
                torch.cuda.synchronize(); lib.kvz_prof_reset(); lib.kvz_prof_enable(1)
                for _ in range(30): call()
                torch.cuda.synchronize(); lib.kvz_prof_enable(0)
                for kn in ("score_rowstat", "score_bounds", "score_colmax"):
                    tt, c = C.c_double(0), C.c_int64(0); lib.kvz_prof_read(kn.encode(), C.byref(tt), C.byref(c))
                    if c.value: t[kn] = round(tt.value / c.value * 1e3, 1)
            extra = ""
            ng = (G * q_len + 255) // 256 * 8; nkb = (m + 31) // 32
            eb = ((Hkv * nkb * ng + 128 + 256) * 4 + 255) // 256 * 256       # per-head counters, redo words, pair entries
            kb_ = (Hkv * ng * (1 + nkb * 32) * 4 + 255) // 256 * 256          # per-group counters + key lists (round 6), behind them
            if pr == 5:   # pair-level candidates: [counter per head, ..., entries]
                cnt = int(ws[need - kb_ - eb:need - kb_ - eb + 4 * Hkv].view(torch.int32).sum())
                extra = f"  candidate pairs {cnt} of {Hkv * nkb * ng} ({100.0 * cnt / (Hkv * nkb * ng):.1f} %)"
            if pr in (3, 4):   # key-level candidates: one counter per (head, row group)
                gc = ws[need - kb_:need - kb_ + 4 * Hkv * ng].view(torch.int32)
                extra = (f"  candidate (group, key) items {int(gc.sum())} = {float(gc.sum()) / (Hkv * m):.2f} per key, {float(gc.float().mean()):.1f} per group (max {int(gc.max())}); "
                         f"32x32 blocks {int(((gc + 31) // 32).sum())} (pair level: {Hkv * nkb * ng} pairs in all)")
            if False:
                if os.environ.get("PRUNE_DEBUG_OLD_LAYOUT"):
                    nb = (Hkv * ng * 32 * 4 + 255) // 256 * 256
                    nrow = ws[need - eb - nb:need - eb].view(torch.float32)[:Hkv * ng * 32].view(Hkv, ng * 32)
                    R = G * q_len
                    qh = q[0, :G].reshape(R, D).float()                      # rows of KV head 0, g-major
                    kv = torch.cat([k[0, 0, :sink], k[0, 0, start:start + m], k[0, 0, klen - q_len:]]).float()
                    x = ((qh @ kv.t()).half().float() / (D ** 0.5)).half().float()
                    qi = torch.arange(R, device=dev) % q_len
                    vis = torch.arange(kv.shape[0], device=dev)[None, :] <= (sink + m + qi)[:, None]
                    c = torch.logsumexp(x.masked_fill(~vis, float("-inf")), dim=1)
                    dn = (nrow[0, :R] + c).abs()
                    bad = torch.nonzero(~(dn <= 1e-2)).flatten()
                    extra += f"\n    nrow vs torch (head 0): max |diff| {float(dn.max()):.3e}, rows off by > 1e-2: {bad.numel()} first {bad[:16].tolist()}; their diffs {[round(float(x), 3) for x in (nrow[0, :R] + c)[bad[:16]]]}; histogram of bad rows % 32: {torch.bincount(bad % 32, minlength=32).tolist()}; bad rows // 256 (first): {sorted(set((bad // 256).tolist()))[:20]}"
                    t_ref = (x[:, sink:sink + m] - c[:, None]).amax(0)
                    d3 = (out[3][0] - t_ref).abs(); d1 = (out[1][0] - t_ref).abs() if 1 in out else d3
                    extra += f"\n    scores vs torch (head 0): variant 3 max {float(d3.max()):.3e}, variant 1 max {float(d1.max()):.3e}"
                    lg = log[0].clone().view(torch.float32)   # log-scores of head 0 as the kernels left them
                    tt = x[:, sink:sink + m] + nrow[0, :R][:, None]
                    t_emul = tt.amax(0)
                    badk = torch.nonzero((lg - t_emul).abs() > 1e-4).flatten()
                    extra += f"\n    log-scores vs emulation (x + nrow): {badk.numel()} keys differ; "
                    rows_true = tt.argmax(0)
                    info = []
                    for jb in badk[:10].tolist():
                        hit = torch.nonzero((tt[:, jb] - lg[jb]).abs() < 2e-6).flatten().tolist()
                        info.append((jb, int(rows_true[jb]), int(rows_true[jb]) % 32, round(float(t_emul[jb]), 4), round(float(lg[jb]), 4), [(r, r % 32) for r in hit[:3]]))
                    extra += f"(key, true row, row % 32, true t, got t, rows whose value was returned) {info}"
                    extra += f"\n    true-row % 32 histogram of the differing keys: {torch.bincount(rows_true[badk] % 32, minlength=32).tolist()}"
                    # which group holds the true maximum of the first bad key, and is it in the list?
                    ent = ws[need - eb + 512 + 1024:need - eb + 512 + 1024 + 4 * int(ws[need - eb:need - eb + 4].view(torch.int32)[0])].view(torch.int32).long() & 0xFFFFFFFF
                    jb = int(torch.nonzero(d3 > 1e-3).flatten()[0]) if (d3 > 1e-3).any() else -1
                    if jb >= 0:
                        rstar = int((x[:, sink + jb] - c).argmax()); gstar = rstar // 32
                        code = gstar | ((jb // 32) << 11) | (0 << 25)
                        extra += f"\n    key {jb}: true max at row {rstar} (group {gstar}), value {float(t_ref[jb]):.4f}, variant 3 {float(out[3][0, jb]):.4f}; pair in the list: {bool((ent == code).any())}"
            print(f"  prune={pr} {json.dumps(t)}{extra}")
        lib.kvz_debug_set_tunable(b"score_prune", -1)
        ref = out[variants[0]]
        msg = []
        for pr in variants[1:]:
            same = torch.equal(torch.nan_to_num(out[pr], nan=7.0), torch.nan_to_num(ref, nan=7.0))
            d = (torch.nan_to_num(out[pr], nan=7.0) - torch.nan_to_num(ref, nan=7.0)).abs()
            msg.append(f"{variants[0]}vs{pr}: equal {same} n_diff {int((d > 0).sum())} max {float(d.max()):.2e}")
        if 1 in out and 3 in out:
            d13 = torch.nan_to_num(out[3], nan=7.0) - torch.nan_to_num(out[1], nan=7.0)
            msg.append(f"3-1: lower {int((d13 < 0).sum())} higher {int((d13 > 0).sum())} first bad keys {torch.nonzero(d13[0] != 0)[:12].flatten().tolist()}")
            msg.append(f"1==3 bitwise {torch.equal(torch.nan_to_num(out[1], nan=7.0), torch.nan_to_num(out[3], nan=7.0))}")
        print(f"shape Hkv{Hkv} G{G} m{m} D{D} sink{sink} q{q_len} s0{s0} {kind}: " + "; ".join(msg) + f"; nan rows/keys {int(ref.isnan().sum())}", flush=True)

if __name__ == "__main__":
    main()
