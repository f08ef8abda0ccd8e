"""Multi-GPU bookkeeping: robots are independent, so a node run shards the batch —
one process per GPU, contiguous shards generated per rank, NO collective on the data path. The only
communication is the timing protocol of bench.py: a barrier and a MAX-reduction of two scalars.
Both go over "gloo" by default, on GPUs too: with no exchange step on the path there is nothing for an RCCL
communicator to carry, and the timing protocol should not depend on one coming up ("nccl" = RCCL can be asked for)."""
import os

import torch
import torch.distributed as dist


def env_rank():
    """(rank, local_rank, world) from the torch.distributed.run environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, local_rank=0):
    rank, _, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(values, device="cpu"):
    """element-wise MAX of a list of floats over all ranks (identity when not distributed)"""
    if not dist.is_initialized():
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def shard_bounds(global_batch, world, rank):
    """contiguous slice [lo, hi) of a global batch owned by `rank` (remainder spread over low ranks)"""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def node_throughput(robots_per_rank, world, steps, elapsed_max):
    """whole-job control-ticks/sec: every rank's robots x steps over the slowest rank's time"""
    return robots_per_rank * world * steps / elapsed_max


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
