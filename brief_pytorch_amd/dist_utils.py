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


def allreduce_sum(values, device):
    """sum a small float64 vector over the ranks (SSE per checkpoint, SSIM sums, slice and voxel counts: exact for the
    integer-valued entries below 2^53).  RCCL over xGMI on GPUs, gloo (host tensors) in the CPU tests."""
    dist, _, _ = dist_info()
    if dist is not None and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor(np.asarray(values, np.float64), dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t)
    return t.cpu().numpy()


def allreduce_max(arr, device):
    """elementwise maximum of a small non-negative array over the ranks (max-intensity projections of z-slabs: a rank that
    does not hold a row leaves it zero).  Integers travel as int64, floats as float64: exact for every image dtype."""
    dist, _, _ = dist_info()
    a = np.asarray(arr)
    if dist is None:
        return a
    if dist.get_backend() == "gloo":
        device = "cpu"
    wide = np.int64 if np.issubdtype(a.dtype, np.integer) else np.float64
    t = torch.from_numpy(np.ascontiguousarray(a.astype(wide))).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy().astype(a.dtype)


def allreduce_sse(sse, count, device):
    """sum [sse_0..sse_k, count] over ranks"""
    out = allreduce_sum(list(sse) + [float(count)], device)
    return out[:-1], float(out[-1])


def broadcast_object(obj, src=0):
    """a small picklable object (block names and byte budgets, a directory name) from rank `src` to every rank"""
    dist, _, _ = dist_info()
    if dist is None:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]
