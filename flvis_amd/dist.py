"""Multi-GPU plumbing of the throughput run: streams are independent, so the batch is partitioned over
ranks with NO data-path collective (weak scaling: every rank tracks `per_gpu` streams).  The only exchange is one
all-gather of the per-stream final poses and one all-reduce(sum) of the job counters after the timed region
(RCCL over xGMI with backend "nccl"; the same code runs on gloo/CPU in the tests)."""
import torch
import torch.distributed as dist


def shard_streams(rank, world, per_gpu):
    """Global stream ids owned by `rank`: rank g owns [g*per_gpu, (g+1)*per_gpu)."""
    if not (0 <= rank < world) or per_gpu <= 0:
        raise ValueError("bad shard request")
    return list(range(rank * per_gpu, (rank + 1) * per_gpu))


def exchange_results(poses, counters, device=None):
    """poses: [per_gpu, 7] float64 tensor of this rank; counters: sequence of ints.
    Returns (all_poses [world*per_gpu, 7] ordered by global stream id, summed counters list).  No-op without a group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return poses, [int(c) for c in counters]
    dev = device if device is not None else poses.device
    poses = poses.to(dev).contiguous()
    gathered = [torch.empty_like(poses) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, poses)
    c = torch.tensor([int(x) for x in counters], dtype=torch.int64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return torch.cat(gathered, 0), [int(x) for x in c.tolist()]


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (the bench's elapsed time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
