"""Multi-GPU glue: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  torch is plumbing here: device selection, the process
group and the collective; the maps it reduces live in the uvrt context.

Two ways to shard.  By RAY RANGE (BASELINE configs[3], the default of bench.py at N > 1): every
rank traces its global-id range of EVERY launch into private int32 count planes, one SUM all-reduce of the
planes per batch (native: uvrt_reduce_batch over RCCL), then every rank replays accumulate + Shade -- see
ray_range / reduce_planes below and include/uvrt.h "batched tracing".  By LAUNCH (configs[4]):  A computation is a fixed global sequence of lamp launches
(iteration-major, lamp-minor: raytracer.cpp:66-72 inside myapp.cpp:156-163); launch k is traced
by rank k % world, every rank advances generate.cl's SEED chain over all launches
(RayTracer::shardRank/shardWorld), and the only exchange is one reduction at the end:

    photonMap     f64[T]  SUM   (counts * duration: integers, exact in f64 -> order independent)
    maxPhotonMap  f64[T]  MAX   (exact)

after which computeDosage gives every rank the single-GPU dose bit for bit.

Load order: initialise torch's HIP runtime (torch.cuda.set_device / init) BEFORE the first uvrt
context is created; a process that initialises /opt/rocm's libamdhip64 first leaves torch's
bundled runtime without devices.  Payload: 16 B per
triangle (0.7 MB for 45 k triangles) -- latency-bound, so the reduction happens once per
computation, never once per launch.
"""
from __future__ import annotations

import numpy as np

from . import capi


def owner(launch_index, world):
    return launch_index % world


def launches(lamp_world_positions, iterations):
    """The global launch sequence: (launch_index, lamp_index) iteration-major."""
    k = 0
    for _ in range(iterations):
        for li in range(len(lamp_world_positions)):
            yield k, li
            k += 1


def seed_chain(lamp_world_positions, light_length, iterations, seed0=0):
    """SEED_{k-1} for every launch k of the computation (host-side RNG walk, no GPU needed)."""
    seeds = []
    s = seed0
    for _, li in launches(lamp_world_positions, iterations):
        seeds.append(s)
        s = capi.seed_next(lamp_world_positions[li], light_length, s)
    return seeds, s


def ray_range(rank, world, n):
    """(first, count): the contiguous share of the global ids [0, n) of every launch that `rank` traces in a
    ray-range-sharded job (RayTracer::SetRayRange; the union over the ranks is the whole launch)."""
    share = (n + world - 1) // world
    first = min(rank * share, n)
    return first, min(share, n - first)


def reduce_planes(planes, group=None):
    """In-place int32 SUM all-reduce of the count planes [launches][T] of a batch: the ONE collective of a
    ray-range-sharded computation (the native path is uvrt_reduce_batch; this is the torch.distributed form
    for rehearsals on gloo)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(planes, op=dist.ReduceOp.SUM, group=group)


def reduce_maps(sum_map, max_map, group=None):
    """In-place all-reduce of the two per-triangle maps (torch tensors on any device): SUM for
    photonMap, MAX for maxPhotonMap.  Two collectives per computation in total."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(sum_map, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_map, op=dist.ReduceOp.MAX, group=group)


class _DevArray:
    """Zero-copy view of a device array for torch (`__cuda_array_interface__`, version 2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape),
                                         "typestr": typestr, "version": 2, "strides": None}


def wrap_map(ctx, which, device):
    """The context's per-triangle f64 map `which` (0 sum, 1 max) as a torch tensor aliasing the
    device memory."""
    import torch
    ptr, nbytes = ctx.device_ptr(which)
    t = torch.as_tensor(_DevArray(ptr, (nbytes // 8,), "<f8"), device=device)
    if t.data_ptr() != ptr:
        raise capi.UvrtError("torch copied the device array instead of aliasing it")
    return t


def wrap_array(ctx, which, device, typestr):
    """Any of the context's per-triangle arrays (uvrt_device_ptr numbering) as an aliasing torch
    tensor; which = 2 is tempPhotonMap (int32, folded to its single-array form by the call)."""
    import torch
    ptr, nbytes = ctx.device_ptr(which)
    itemsize = int(typestr[-1])
    t = torch.as_tensor(_DevArray(ptr, (nbytes // itemsize,), typestr), device=device)
    if t.data_ptr() != ptr:
        raise capi.UvrtError("torch copied the device array instead of aliasing it")
    return t


class MapReducer:
    """Reduces a RayTracer's photonMap / maxPhotonMap across ranks.  Ordering contract: the
    context must run on torch's CURRENT stream (ctx.set_stream(torch.cuda.Stream().cuda_stream)
    + torch.cuda.set_stream), which is the stream the collective is enqueued behind; with the
    context on its own private stream call ctx.sync() before and torch.cuda.synchronize() after.
    Aliases the device arrays when torch accepts the array interface, otherwise stages through
    two torch buffers with uvrt_copy_device."""

    def __init__(self, ctx, device, group=None):
        import torch
        self.ctx, self.group = ctx, group
        try:
            self.sum_t = wrap_map(ctx, 0, device)
            self.max_t = wrap_map(ctx, 1, device)
            self.staged = False
        except Exception:
            T = ctx.device_ptr(0)[1] // 8
            self.sum_t = torch.empty(T, dtype=torch.float64, device=device)
            self.max_t = torch.empty(T, dtype=torch.float64, device=device)
            self.staged = True

    def __call__(self):
        # uvrt_device_ptr orders the context's stream after all outstanding launches (they may sit
        # on the library's second stream, include/uvrt.h uvrt_set_pipeline) and makes the next
        # accumulate / Shade wait for what is enqueued here: call it before EVERY external use of the
        # maps, not only once when the tensors are made
        self.ctx.device_ptr(0)
        self.ctx.device_ptr(1)
        if self.staged:
            self.ctx.copy_device(0, self.sum_t.data_ptr(), False)
            self.ctx.copy_device(1, self.max_t.data_ptr(), False)
        reduce_maps(self.sum_t, self.max_t, self.group)
        if self.staged:
            self.ctx.copy_device(0, self.sum_t.data_ptr(), True)
            self.ctx.copy_device(1, self.max_t.data_ptr(), True)
