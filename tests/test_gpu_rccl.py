"""GPU: RCCL executes once on the MI355X box.  A 1-GPU box cannot run the 8-rank leg of BASELINE config C4,
but it can run everything that leg adds to the single-rank path: ``init_process_group("nccl")`` (= RCCL on ROCm) bound to the device,
the all_gather of the per-context result record (``kvzip_amd.dist.gather_results``) and the MAX all_reduce of ``bench.Ranks`` -
in a process group of ONE rank, in a child process (a process group is process-wide state)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import bench
from kvzip_amd.dist import backend_version
for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
    os.environ.pop(var, None)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ranks = bench.Ranks(1, "nccl", dev, force=True)
assert ranks.forced and ranks.dist.get_backend() == "nccl" and ranks.dist.get_world_size() == 1
ranks.barrier(torch.cuda.synchronize)
len_k = (torch.arange(28 * 4, dtype=torch.int32, device=dev).view(28, 4) * 1000 + 7)
recs = ranks.gather_records(0.1259765625, 0.2998, len_k, 28, 4)       # all_gather over RCCL, record on the device
t = ranks.max_over_ranks(0.75, dev)                                   # all_reduce(MAX) over RCCL
ranks.barrier(torch.cuda.synchronize)
out = {"thres": recs[0]["thres"], "ratio": recs[0]["real_ratio"], "n_kept": recs[0]["n_kept"], "n": len(recs),
       "len_ok": bool(torch.equal(recs[0]["len_k"], len_k.cpu())), "t": t, "version": backend_version()}
ranks.close()
print("RCCL " + json.dumps(out))
"""


@pytest.mark.gpu
def test_rccl_single_rank_gather_and_max():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL ")]
    assert p.returncode == 0 and line, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads(line[0][5:])
    print("RCCL on this box:", d["version"])
    assert d["n"] == 1 and d["thres"] == 0.1259765625 and d["ratio"] == 0.2998 and d["len_ok"] and d["t"] == 0.75
    assert d["n_kept"] == sum(1000 * i + 7 for i in range(28 * 4))
    assert d["version"].startswith("rccl ")
