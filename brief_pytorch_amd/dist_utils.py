"""Multi-GPU glue for DivideTask: blocks are independent fits (main.py:547-575), so ranks only
meet for (i) the static block assignment, which every rank computes identically, and (ii) one
all-reduce of [SSE per checkpoint..., voxel count] for PSNR — RCCL over xGMI on GPUs, gloo in
the CPU tests."""
import numpy as np
import torch


def dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def assign_blocks(costs, world):
    """longest-processing-time-first: blocks sorted by cost, each to the least loaded rank
    (SURVEY.md section 8e).  Deterministic, so no communication is needed to agree on it."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def allreduce_sse(sse, count, device):
    """sum [sse_0..sse_k, count] over ranks (float64, exact for integer-valued SSE < 2^53)"""
    dist, _, _ = dist_info()
    if dist is not None and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor(list(sse) + [float(count)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t)
    out = t.cpu().numpy()
    return out[:-1], float(out[-1])


def gather_objects(obj):
    dist, _, world = dist_info()
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
