#!/usr/bin/env python
"""How much of pass B's work survives an EXACT group-bound pruning?  (round 5, what-comes-next study)
For every ctx key j and 32-row group g let u_gj = max_{r in g} x_rj (a by-product of a key-per-lane pass A), c_r = m_r + log l_r.
   UB_gj = u_gj - min_{r in g} c_r   >=  every x_rj - c_r of the group;      LB_j = max_g (u_gj - max_{r in g} c_r)  <=  t_j.
Only (g, j) with UB_gj >= LB_j can hold the column maximum t_j = max_r (x_rj - c_r); pass B would recompute Q.K^T only for the
32x32 blocks (group g, 32-key block) that contain such a pair.  Prints the surviving fraction of blocks for the bench's Gaussian inputs
and for copy-like prompts (row i attends key i)."""
import sys, torch
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
H, Hkv, D, sink, m = 28, 4, 128, 32, 2000
G = H // Hkv
q_len = m + 26
def study(tag, q, k):   # q [G*q_len, D] rows of one KV head (g-major), k [sink + m + q_len, D]
    R = q.shape[0]
    x = (q.float() @ k.float().t()).half().float() / (D ** 0.5)
    x = x.half().float()                                            # the reference's rounding chain, approximately
    qi = torch.arange(R, device=dev) % q_len
    key = torch.arange(k.shape[0], device=dev)
    vis = key[None, :] <= (sink + m + qi)[:, None]
    xm = x.masked_fill(~vis, float("-inf"))
    c = torch.logsumexp(xm, dim=1)                                  # m_r + log l_r
    t_true = (x[:, sink:sink + m] - c[:, None]).amax(0)
    ng = (R + 31) // 32
    pad = ng * 32 - R
    xc = torch.cat([x[:, sink:sink + m], torch.full((pad, m), float("-inf"), device=dev)]).view(ng, 32, m)
    cc = torch.cat([c, torch.full((pad,), float("inf"), device=dev)]).view(ng, 32)
    u = xc.amax(1)                                                  # [ng, m]
    cmin = cc.amin(1)
    cmax = torch.where(torch.isinf(cc), torch.full_like(cc, float("-inf")), cc).amax(1)
    ub = u - cmin[:, None]
    lb = (u - cmax[:, None]).amax(0)
    cand = ub >= lb[None, :]                                        # [ng, m]
    assert bool((torch.where(cand, ub, torch.full_like(ub, float("-inf"))).amax(0) >= t_true).all())
    nb = (m + 31) // 32
    # the same test with a lower bound taken over the groups of the own row slice only (pass B cuts the rows into SPLITS slices; a slice
    # that tests against its own groups needs nothing from the other slices - exact as well, weaker)
    SPLITS = 8
    per = ((R + 127) // 128 + SPLITS - 1) // SPLITS * 4
    frac = []
    for s0 in range(0, ng, per):
        sl = slice(s0, min(ng, s0 + per))
        lbs = (u[sl] - cmax[sl, None]).amax(0)
        cs = ub[sl] >= lbs[None, :]
        frac.append(torch.cat([cs, torch.zeros(cs.shape[0], nb * 32 - m, dtype=torch.bool, device=dev)], 1).view(cs.shape[0], nb, 32).any(-1).float().sum())
    print(f"{tag:28s}: slice-local lower bound ({SPLITS} slices of {per} groups): blocks to recompute {float(sum(frac)) / (ng * nb) * 100:5.1f} %")
    cb = torch.cat([cand, torch.zeros(ng, nb * 32 - m, dtype=torch.bool, device=dev)], 1).view(ng, nb, 32).any(-1)
    print(f"{tag:28s}: candidate (group, key) pairs {float(cand.float().mean()) * 100:6.3f} %  ({float(cand.sum(0).float().mean()):.1f} groups per key of {ng}); "
          f"32x32 blocks that must be recomputed {float(cb.float().mean()) * 100:5.1f} %; spread of c within a group (mean) {float((cmax - cmin).mean()):.3f}")
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(G * q_len, D, generator=g, device=dev).half(); k = torch.randn(sink + m + q_len, D, generator=g, device=dev).half()
study("gaussian (the bench)", q, k)
# copy-like: the repeat chunk's query i is close to the ctx key it repeats (plus noise), with a realistic logit scale
base = torch.randn(m, D, generator=g, device=dev)
kc = torch.cat([torch.randn(sink, D, generator=g, device=dev), base, torch.randn(q_len, D, generator=g, device=dev)])
for s in (0.5, 1.0, 2.0):
    qrows = torch.randn(G, q_len, D, generator=g, device=dev)
    qrows[:, 26:26 + m] += s * base[None]
    study(f"copy-like, strength {s}", qrows.view(-1, D).half(), kc.half())
