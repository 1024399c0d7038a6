"""torch.distributed glue for the two multi-GPU modes (one process per GPU, RCCL over xGMI).

PyTorch is plumbing here: it owns the process group and the collective; the tensors it
reduces are views of libmmgibbs' own device buffers (no copy), and the kernels are launched on
torch's current stream so the collective is ordered after them.

  chains mode  every rank runs its own chain(s) over the full hit matrix; ONE all-reduce of the
               posterior moments (sum log mu, sum log^2 mu) at the end.
  shard mode   rows are sharded; every iteration all-reduces the int32 count vector between the
               sample and update kernels; every rank then redraws the identical mu (same Philox
               key), so no broadcast is needed and the chain is bit-identical to the 1-GPU chain.
"""
import numpy as np


class _DevBuf:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def devptr_tensor(ptr, count, dtype):
    """Zero-copy torch view of `count` elements of `dtype` at device pointer `ptr`."""
    import torch
    typestr = {torch.int32: "<i4", torch.float64: "<f8"}[dtype]
    return torch.as_tensor(_DevBuf(ptr, count, typestr), device="cuda")


def use_current_stream(sampler):
    """Launch the sampler's kernels on torch's current stream, so that collectives issued through torch are ordered after them.
    torch's default stream is the legacy NULL stream (handle 0), which the C ABI reserves for "the sampler's own stream":
    it is passed as hipStreamLegacy (1)."""
    import torch
    h = torch.cuda.current_stream().cuda_stream
    sampler.set_stream(h if h else 1)


def counts_tensor(sampler):
    import torch
    ptr, cnt = sampler.counts_devptr()
    return devptr_tensor(ptr, cnt, torch.int32)


def moments_tensor(sampler):
    import torch
    ptr, cnt = sampler.moments_devptr()
    return devptr_tensor(ptr, cnt, torch.float64)


def shard_step(sampler, counts, group=None):
    """One read-sharded Gibbs iteration: local sample, all-reduce counts, identical update."""
    import torch
    import torch.distributed as dist
    sampler.sample()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        staged = dist.get_backend(group) != "nccl"     # gloo moves CUDA tensors through the host on streams of its own
        if staged:
            torch.cuda.current_stream().synchronize()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        if staged:
            torch.cuda.synchronize()
    sampler.update()


def pool_moments(moments, group=None):
    """chains mode epilogue: sum the per-chain moments over all ranks (in place)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


def row_shard(total_rows, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` (balanced to within one row)."""
    lo = (total_rows * rank) // world
    hi = (total_rows * (rank + 1)) // world
    return lo, hi


def pooled_summary(sum_log, sum_log2, n_samples):
    """Posterior mean / sd of log mu from pooled moments (host side, numpy)."""
    mean = sum_log / n_samples
    var = (sum_log2 - n_samples * mean * mean) / max(n_samples - 1, 1)
    return mean, np.sqrt(np.maximum(var, 0.0))
