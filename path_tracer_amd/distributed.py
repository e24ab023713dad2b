"""spp-sharding over the GPUs of one node: one process per GPU, each renders a
contiguous range of the samples*samples grid into a full-frame fp32 radiance buffer,
one reduce (RCCL over xGMI: torch.distributed backend "nccl") merges the buffers on
rank 0, which resolves and owns the image.  The reference has no counterpart (single
process, shared-memory threads: src/parallel_renderer/mod.rs:88-145); the unit that
shards is the sample index s = sx * S + sy of src/raytracer.rs:77-81.

The same functions run on CPU tensors with the "gloo" backend (tests/test_distributed_cpu.py).
"""
import os


def shard_range(samples, rank, world):
    """Sample-index range [begin, end) of `rank`: contiguous, sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    total = samples * samples
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, device=None):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend=backend, **kwargs)
    return dist


def reduce_radiance(accum, dist, root=0):
    """Sum the per-rank radiance buffers into `root`'s (W*H*3 fp32: 24.9 MB at 1080p).
    Only the root needs the sum, so this is a reduce, not an all-reduce."""
    if dist is not None:
        dist.reduce(accum, dst=root, op=dist.ReduceOp.SUM)
    return accum
