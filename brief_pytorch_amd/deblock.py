"""Post-decode block-boundary filter for DivideTask outputs — the reference's deblock.py /
deblock.cpp (its only native component) on the GPU.

The filter is in place and sequential over boundary lines (neighbouring lines overlap by up
to three pixels), so bit-exact parity needs the reference's order: blocks in the given order,
per block left / right / bottom / top, every slice.  Slices are independent, so one launch
handles one line for all slices of a block; within a launch every thread owns disjoint pixels.
The order of `block_names` is part of the input (the reference takes os.listdir order).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib


def block_edges(block_names):
    """[(z1, z2, x1, y1, x2, y2)] in processing order with the reference's duplicate rule (deblock.py:109-130:
    a side is skipped when the same line of the block's first slice was already queued)."""
    seen = set()
    edges = []
    for name in block_names:
        dd, hh, ww = name.split("-")
        z1, z2 = (int(v) for v in dd.split("_")[1:])
        x1, x2 = (int(v) for v in ww.split("_")[1:])
        y1, y2 = (int(v) for v in hh.split("_")[1:])
        sides = [(x1, y1, x1, y2), (x2, y1, x2, y2), (x1, y1, x2, y1), (x1, y2, x2, y2)]
        todo = [s for s in sides if (z1,) + s not in seen]
        # the reference appends slice by slice (l, r, d, u per slice); lines of different slices never touch,
        # so issuing each side once for all slices gives identical pixels
        for s in todo:
            edges.append((z1, z2) + s)
            for z in range(z1, z2 + 1):
                seen.add((z,) + s)
    return edges


def deblock_volume(img, block_names, index_a=51, index_b=2000, thres=65535, mode=1):
    """img: uint16 tensor on the GPU, shape (d,h,w) or (d,h,w,1); filtered in place and returned"""
    if img.dtype != torch.uint16 or img.device.type != "cuda" or not img.is_contiguous():
        raise _lib.BriefError("deblock_volume needs a contiguous uint16 tensor on a ROCm GPU")
    d, h, w = img.shape[:3]
    L = _lib.lib()
    for (z1, z2, x1, y1, x2, y2) in block_edges(block_names):
        if x1 == x2:
            _lib.check(L.brief_deblock_edge(_lib.ptr(img), d, h, w, z1, z2, x1, y1, y2, 1, index_a, index_b, thres, mode, _lib.stream_ptr()))
        elif y1 == y2:
            _lib.check(L.brief_deblock_edge(_lib.ptr(img), d, h, w, z1, z2, y1, x1, x2, 0, index_a, index_b, thres, mode, _lib.stream_ptr()))
    return img


def main(step_dir, index_a=51, index_b=2000, thres=65535, mode=1):
    """deblock.py main(): <step_dir>/decompressed/<name>.tif + compressed/module/<blocks> -> deblock/<name>_deblocked.tif"""
    from .tool import read_img, save_img
    dec_dir = os.path.join(step_dir, "decompressed")
    name = sorted(os.listdir(dec_dir))[0]
    img = read_img(os.path.join(dec_dir, name))
    blocks = sorted(os.listdir(os.path.join(step_dir, "compressed", "module")))
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    out = deblock_volume(t, blocks, index_a, index_b, thres, mode).cpu().numpy()
    os.makedirs(os.path.join(step_dir, "deblock"), exist_ok=True)
    path = os.path.join(step_dir, "deblock", name[:-4] + "_deblocked.tif")
    save_img(path, out)
    return path
