#!/usr/bin/env python
"""How many 32x32 blocks of pass B could an EXACT epilogue skip?  (runs on the CPU.)

Pass B computes t_j = max_r (x_rj - lse_r) for the ctx keys.  The rounding chain and the subtraction are monotone per row, so a
lane (one query row, 16 keys of a 32-key group) can bound its 16 values by chain(max acc): if that bound is <= the smallest of
the 16 running maxima it compares against, nothing in the lane can raise a maximum, and if that holds for all 64 lanes of the
wave the block's epilogue can be skipped without changing a bit.  What the running maxima ARE decides the skip rate:
  lane    the maxima a lane has accumulated itself (its own rows only: what the kernel holds today)
  wave    per-key maxima over all rows the wave has seen, refreshed every P row tiles (needs a cross-lane reduction per refresh)
  global  the same, shared between the 8 row slices of a key tile through memory (slices assumed to advance in lockstep)
Geometry of the bench: one KV head, G*q = 14182 query rows in 111 tiles of 128 (4 steps of 32 rows), 8 row slices, m = 2000 keys.
Inputs: gaussian logits (bench.py's synthetic data) and a "copy" pattern (query row i strongly attends ctx key i mod m, as the
repeat chunk of the real scoring forward does)."""
import math, sys
import torch

torch.manual_seed(0)
D, m, sink, G, q_len = 128, 2000, 32, 7, 2026
R = G * q_len
TILE, SLICES = 128, 8

def logits(kind):
    k = torch.randn(sink + m + q_len, D)
    q = torch.randn(R, D)
    if kind == "copy":
        idx = (torch.arange(R) % q_len).clamp(max=m - 1)
        q = 0.35 * q + 0.65 * k[sink + idx] * 3.0
    x = (q @ k.t()) / math.sqrt(D)
    # causal mask inside the repeat part
    qi = (torch.arange(R) % q_len).view(R, 1)
    kj = torch.arange(sink + m + q_len).view(1, -1)
    x = x.masked_fill(kj > sink + m + qi, float("-inf"))
    lse = torch.logsumexp(x, dim=1, keepdim=True)
    return (x[:, sink:sink + m] - lse)   # [R, m] log-softmax of the ctx keys

def simulate(t, mode, P):
    ntiles = (R + TILE - 1) // TILE
    per = (ntiles + SLICES - 1) // SLICES
    nk = m // 32 * 32
    t = t[:, :nk]
    ngroups = nk // 32
    skipped = total = 0
    # thresholds [slice or 1][key]; lane-local: [slice][32 row residues][key]
    neg = float("-inf")
    if mode == "lane":
        thr = torch.full((SLICES, 32, nk), neg)
    else:
        thr = torch.full((SLICES if mode == "wave" else 1, nk), neg)
        pend = torch.full_like(thr, neg)   # maxima seen since the last refresh (not yet visible to the test)
    for step_tile in range(per):
        for s in range(SLICES):
            tile = s * per + step_tile
            if tile >= ntiles: continue
            r0 = tile * TILE
            for b in range(4):
                rows = t[r0 + b * 32: r0 + b * 32 + 32]           # [<=32, nk]
                if rows.shape[0] == 0: continue
                n = rows.shape[0]
                v = rows.view(n, ngroups, 2, 4, 4)                 # keys of a group: (i>>2) , half , (i&3) -> 8*(i>>2)+4*half+(i&3)
                # lane (row, half) of group g holds keys 8*a + 4*half + c, a in 0..3, c in 0..3
                lane_max = rows.view(n, ngroups, 4, 2, 4).amax(dim=(2, 4))          # [n, groups, half]
                if mode == "lane":
                    th = thr[s, :n].view(n, ngroups, 4, 2, 4).amin(dim=(2, 4))
                else:
                    th = thr[s if mode == "wave" else 0].view(ngroups, 4, 2, 4).amin(dim=(1, 3)).unsqueeze(0).expand(n, -1, -1)
                need = (lane_max > th).any(dim=2).any(dim=0)                          # [groups]: some lane of the wave must run
                total += ngroups
                skipped += int((~need).sum())
                if mode == "lane":
                    thr[s, :n] = torch.maximum(thr[s, :n], rows)
                else:
                    i = s if mode == "wave" else 0
                    pend[i] = torch.maximum(pend[i], rows.amax(dim=0))
        if mode != "lane" and (step_tile + 1) % P == 0:
            thr = torch.maximum(thr, pend)
    return skipped / total

for kind in ("gaussian", "copy"):
    t = logits(kind)
    print(f"== {kind} logits: max_r log-softmax per key: median {float(t.amax(0).median()):.2f}")
    print(f"   lane-local maxima                      : {simulate(t, 'lane', 1):6.1%} of the blocks skippable")
    for P in (1, 2, 4):
        print(f"   wave maxima, refresh every {P} tile(s)     : {simulate(t, 'wave', P):6.1%}")
    for P in (1, 2, 4):
        print(f"   shared by the 8 slices, refresh every {P}  : {simulate(t, 'global', P):6.1%}")
