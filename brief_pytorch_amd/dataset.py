"""Coordinate grids of the reference (utils/dataset.py:11-62) as host tensors.

The fused kernels never need them — coordinates are synthesised in-kernel from the voxel index with the same
arithmetic — but code written against the reference (`reconstruct_flattened`, custom samplers) does.
torch.linspace is used as is, so the values are the reference's bit for bit.
"""
import math

import torch

__all__ = ["create_coords", "create_flattened_coords", "reconstruct_flattened"]


def _range(mode):
    if mode == "n11":
        return -1.0, 1.0
    if mode == "0p1":
        return 0.0, 1.0
    lo, hi = mode.split(",")
    return float(lo), float(hi)


def create_coords(coords_shape, mode="n11"):
    """(h,w,2) or (d,h,w,3) grid of linspace(lo,hi,n) per axis, 'ij' indexing (utils/dataset.py:11-35)"""
    lo, hi = _range(mode)
    if len(coords_shape) not in (2, 3):
        raise NotImplementedError
    axes = [torch.linspace(lo, hi, int(n)) for n in coords_shape]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)


def create_flattened_coords(coords_shape, mode="n11"):
    """create_coords flattened to ((d) h w, c) (utils/dataset.py:36-62)"""
    c = create_coords(coords_shape, mode)
    return c.reshape(-1, c.shape[-1])


def reconstruct_flattened(data_shape, sample_size, sample_nf, device="cpu", half=False, coords_mode="-1,1"):
    """utils/misc.py:59-92: evaluate `sample_nf` over the whole grid in chunks of `sample_size` coordinates and
    reshape to data_shape.  Kept for code written against the reference; NFGR.decompress uses SIREN.decode_grid
    (one launch, coordinates synthesised in-kernel) instead.  half=True follows the reference's fp16 decode (main.py:287-288):
    fp16 coordinates and an fp16 result; the module is then expected to be in its low-precision mode (SIREN.half())."""
    *coords_shape, data_channel = data_shape
    if len(coords_shape) not in (2, 3):
        raise NotImplementedError
    with torch.no_grad():
        coords = create_flattened_coords(tuple(coords_shape), coords_mode).to(device)
        pop = coords.shape[0]
        out = torch.zeros((pop, data_channel), device=device)
        if half:
            out = out.half()            # utils/misc.py:70-71, 78-79: an fp16 result, fp16 coordinates into the (half) module
        for i in range(math.ceil(pop / sample_size)):
            a, b = i * sample_size, min((i + 1) * sample_size, pop)
            sampled = coords[a:b, :]
            if half:
                sampled = sampled.half()
            out[a:b, :] = sample_nf(sampled)
    return out.reshape(*coords_shape, data_channel)
