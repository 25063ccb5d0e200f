"""Samples-per-pixel sharding across the GPUs of a node (one process per GPU).

Every sample is an independent path (reference src/camera/ray_casting.rs:82-105)
and the RNG is keyed by (seed, pixel, sample), so rank r of N renders sample
indices [r*spp/N, (r+1)*spp/N) of EVERY pixel and the union is the 1-GPU sample
set.  Each rank holds a full-frame RGB *sum*; one reduce (RCCL over xGMI on GPUs,
gloo in CPU tests) adds them on rank 0, which divides by spp (average_samples,
ray_casting.rs:154-173).  The result differs from the 1-GPU image only by the
order of the floating-point adds.
"""


def shard_range(rank, world, spp):
    """(sample_begin, sample_count) of `rank`; the ranges partition [0, spp)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside [0, world)")
    begin = rank * spp // world
    end = (rank + 1) * spp // world
    return begin, end - begin


def reduce_to_mean(sum_tensor, spp, dst=0):
    """In-place: reduce the per-rank sums to `dst` and turn them into the per-pixel mean there.
    Without an initialised process group this is the 1-GPU case (divide only)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(sum_tensor, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() != dst:
            return sum_tensor
    sum_tensor.div_(float(spp))
    return sum_tensor
