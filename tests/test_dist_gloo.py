"""CPU, world_size 2 over gloo: the N>1 path = context sharding + the result-record gather (the only
exchange the path has).  Compute is GPU-only, so records are synthetic here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kvzip_amd.dist import gather_results, pack_record, shard_contexts, unpack_record


def test_shard_contexts_partition():
    for n, w in ((8, 8), (8, 2), (5, 2), (3, 4), (0, 2)):
        parts = [shard_contexts(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_record_round_trip():
    len_k = torch.arange(28 * 4, dtype=torch.int32).view(28, 4) * 1000 + 7
    rec = pack_record(0.1259765625, 0.2998, len_k)
    out = unpack_record(rec, 28, 4)
    assert out["thres"] == 0.1259765625 and out["real_ratio"] == 0.2998
    assert torch.equal(out["len_k"], len_k) and out["n_kept"] == int(len_k.sum())


def _worker(rank, world, port, n_ctx, L, Hkv, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_contexts(n_ctx, rank, world)
    recs = [pack_record(0.5 + ctx, 0.25 + 0.01 * ctx, torch.full((L, Hkv), 100 * ctx + 3, dtype=torch.int32))
            for ctx in mine]
    out = gather_results(recs, n_ctx, L, Hkv)
    ok = all(o["thres"] == 0.5 + c and abs(o["real_ratio"] - (0.25 + 0.01 * c)) < 1e-12 and
             int(o["len_k"][0, 0]) == 100 * c + 3 and o["n_kept"] == L * Hkv * (100 * c + 3) for c, o in enumerate(out))
    q.put((rank, ok, len(out)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_ctx", [2, 5])
def test_gather_results_world2_gloo(n_ctx):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_ctx, 3, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and n == n_ctx for _, ok, n in res)


# ---- bench.py's own rank logic (Ranks + launch) on gloo, world_size 2 ---------------------------------------------
def _bench_entry(argv):
    """What bench.main does around the kernels: Ranks from the environment, barrier, timed region, the library gather of the
    per-context record inside it, max over ranks - with CPU tensors and the gloo backend."""
    import json
    import time
    import bench
    out_dir, L, Hkv = argv[0], int(argv[1]), int(argv[2])
    world = int(argv[3]) if len(argv) > 3 else 2
    ranks = bench.Ranks(world, "gloo")
    ranks.barrier()
    t0 = time.perf_counter()
    len_k = torch.full((L, Hkv), 1000 * (ranks.rank + 1), dtype=torch.int32)
    recs = ranks.gather_records(0.25 + ranks.rank, 0.3 + 0.01 * ranks.rank, len_k, L, Hkv)
    ranks.barrier()
    mine = time.perf_counter() - t0 + (0.5 if ranks.rank == 1 else 0.0)   # rank 1 pretends to be slower
    elapsed = ranks.max_over_ranks(mine)
    with open(os.path.join(out_dir, f"rank{ranks.rank}.json"), "w") as f:
        json.dump({"rank": ranks.rank, "world": ranks.world, "local_rank": ranks.local_rank, "elapsed": elapsed, "mine": mine,
                   "thres": [r["thres"] for r in recs], "n_kept": [r["n_kept"] for r in recs],
                   "len00": [int(r["len_k"][0, 0]) for r in recs]}, f)
    ranks.close()


def test_bench_self_spawn_and_rank_logic_world2_gloo(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` without a launcher: bench.launch spawns the ranks itself; every rank sees the full gather
    (kvzip_amd.dist.gather_results) and the max-over-ranks time."""
    import json
    import bench
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(var, raising=False)
    bench.launch([str(tmp_path), "3", "2"], _bench_entry, 2)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, d in enumerate(res):
        assert d["rank"] == r == d["local_rank"] and d["world"] == 2
        assert d["thres"] == [0.25, 1.25] and d["len00"] == [1000, 2000] and d["n_kept"] == [6000, 12000]
    assert res[0]["elapsed"] == res[1]["elapsed"] >= res[1]["mine"] >= 0.5   # MAX over ranks, identical everywhere


def test_bench_self_spawn_world8_gloo_record_order(tmp_path, monkeypatch):
    """The shape of BASELINE config C4 (8 independent contexts, one per rank) on CPU: `bench.launch` spawns EIGHT ranks over gloo,
    every rank contributes the record of its own context, and every rank must see all eight records in context order (shard order =
    rank order at N = 8) and the same max-over-ranks time.  The 8-GPU run itself needs a node this pool does not hand out."""
    import json
    import bench
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(var, raising=False)
    bench.launch([str(tmp_path), "4", "2", "8"], _bench_entry, 8)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(8)]
    for r, d in enumerate(res):
        assert d["rank"] == r == d["local_rank"] and d["world"] == 8
        assert d["thres"] == [0.25 + i for i in range(8)]
        assert d["len00"] == [1000 * (i + 1) for i in range(8)] and d["n_kept"] == [8 * 1000 * (i + 1) for i in range(8)]
    assert len({d["elapsed"] for d in res}) == 1 and res[0]["elapsed"] >= res[1]["mine"] >= 0.5
    assert [shard_contexts(8, r, 8) for r in range(8)] == [[i] for i in range(8)]


def test_bench_ranks_rejects_wrong_world(monkeypatch):
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit):
        bench.Ranks(2, "gloo")
    args = bench.parse(["--level", "head"])
    assert args.ratio == 0.6 and bench.parse([]).ratio == 0.3


def _forced_world1_entry(out_path):
    """`bench.py --force-dist` on one rank: the process group exists, the gather and the max reduction go through it."""
    import json
    import bench
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        os.environ.pop(var, None)
    ranks = bench.Ranks(1, "gloo", force=True)
    assert ranks.forced and ranks.dist is not None and ranks.dist.get_world_size() == 1
    ranks.barrier()
    len_k = torch.full((3, 2), 777, dtype=torch.int32)
    recs = ranks.gather_records(0.125, 0.3, len_k, 3, 2)
    t = ranks.max_over_ranks(1.5)
    ranks.close()
    with open(out_path, "w") as f:
        json.dump({"thres": recs[0]["thres"], "n_kept": recs[0]["n_kept"], "t": t, "n": len(recs)}, f)


def test_bench_force_dist_world1_gloo(tmp_path):
    """The forced single-rank process group (what tests/test_gpu_rccl.py runs over RCCL on the GPU box), here over gloo."""
    import json
    out = str(tmp_path / "forced.json")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_forced_world1_entry, args=(out,))
    p.start()
    p.join(timeout=120)
    assert p.exitcode == 0
    d = json.load(open(out))
    assert d == {"thres": 0.125, "n_kept": 3 * 2 * 777, "t": 1.5, "n": 1}
