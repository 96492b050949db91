"""Multi-GPU driver for the eviction path: independent contexts, one per GPU.

The reference is batch-1 and single-GPU (no collective anywhere); contexts never interact, so the
path shards by context with NO data-path collective.  The only exchange is the gather of a fixed-size result
record per context over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in CPU tests):
``{thres f64, real_ratio f64, n_kept i64, len_k int32[L*Hkv]}`` — 24 B + 4*L*Hkv B (472 B for Qwen2.5-7B):
latency-bound on any of the 7 xGMI links, which is why a plain all_gather (no ring all-reduce) is used.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_contexts(n_contexts: int, rank: int, world: int) -> List[int]:
    """Context ids owned by ``rank``: contiguous blocks, remainder spread over the first ranks."""
    base, rem = divmod(n_contexts, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def pack_record(thres: float, r_real: float, len_k: torch.Tensor) -> torch.Tensor:
    """-> float64 tensor [3 + L*Hkv] on len_k's device (int32 lengths are exact in float64)."""
    flat = len_k.reshape(-1).to(torch.float64)
    head = torch.tensor([float(thres), float(r_real), float(flat.sum().item())], dtype=torch.float64, device=flat.device)
    return torch.cat([head, flat])


def unpack_record(rec: torch.Tensor, layers: int, Hkv: int) -> Dict:
    rec = rec.cpu()
    return {"thres": float(rec[0]), "real_ratio": float(rec[1]), "n_kept": int(rec[2]),
            "len_k": rec[3:].to(torch.int32).view(layers, Hkv)}


def gather_results(records: Sequence[torch.Tensor], n_contexts: int, layers: int, Hkv: int,
                   group: Optional[dist.ProcessGroup] = None, force_collective: bool = False) -> List[Dict]:
    """Every rank contributes the records of the contexts it owns (``shard_contexts`` order); every rank gets
    the full list back in context order.  One all_gather of ``ceil(n_contexts/world)`` fixed-size slots.
    ``force_collective``: issue the all_gather even in a process group of ONE rank (a 1-GPU box still exercises RCCL:
    ``bench.py --force-dist``, ``tests/test_gpu_rccl.py``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_contexts(n_contexts, rank, world)
    assert len(records) == len(mine)
    width = 3 + layers * Hkv
    slots = -(-n_contexts // world)
    dev = records[0].device if records else torch.device("cpu")
    buf = torch.full((slots, width), float("nan"), dtype=torch.float64, device=dev)
    for i, r in enumerate(records):
        buf[i] = r
    if world == 1 and not (force_collective and dist.is_initialized()):
        gathered = [buf]
    else:
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
    out: List[Optional[Dict]] = [None] * n_contexts
    for rk in range(world):
        for i, ctx in enumerate(shard_contexts(n_contexts, rk, world)):
            out[ctx] = unpack_record(gathered[rk][i], layers, Hkv)
    return out


def backend_version() -> str:
    """``"rccl x.y.z"`` of the collective library behind torch.distributed's "nccl" backend on ROCm ("" without a GPU build)."""
    try:
        v = torch.cuda.nccl.version()
        return "rccl " + ".".join(str(x) for x in (v if isinstance(v, tuple) else (v,)))
    except Exception:  # CPU-only build / no nccl module
        return ""
