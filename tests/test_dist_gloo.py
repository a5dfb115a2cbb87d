"""The N>1 path of the throughput run on CPU: two gloo processes shard 2x4 streams, exchange poses/counters exactly
like bench.py does over RCCL, and agree on the result."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, per_gpu, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from flvis_amd import dist as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = fd.shard_streams(rank, world, per_gpu)
    poses = torch.tensor([[float(s), 0, 0, 0, 0, 0, 1.0] for s in ids], dtype=torch.float64)
    all_poses, counters = fd.exchange_results(poses, [per_gpu * 10, rank + 1, 3])
    tmax = fd.max_over_ranks(1.0 + rank)
    q.put((rank, ids, all_poses[:, 0].tolist(), counters, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_exchange():
    world, per_gpu = 2, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6, 7]          # disjoint, contiguous shards
    for r in res:
        assert r[2] == [float(i) for i in range(8)]                          # all-gather ordered by global stream id
        assert r[3] == [80, 3, 6]                                            # all-reduce(sum) of counters
        assert r[4] == 2.0                                                   # max over ranks


def test_shard_validation_and_single_process_noop():
    sys.path.insert(0, ROOT)
    from flvis_amd import dist as fd
    with pytest.raises(ValueError):
        fd.shard_streams(2, 2, 4)
    p = torch.zeros((3, 7), dtype=torch.float64)
    out, c = fd.exchange_results(p, [1, 2])
    assert out is p and c == [1, 2] and fd.max_over_ranks(3.5) == 3.5
