"""Normalise / de-normalise and small file helpers (reference utils/io.py:65-214, 216-313).

normalize_data / invnormalize_data reproduce the reference's float32 operation order exactly
(tests pin them bit for bit against golden vectors).  On the fused path the de-normalise step is
normally done inside the decode kernel (SIREN.decode_grid(out_kind='u16')); the numpy versions
here serve host-side callers and the checker of that epilogue.
"""
import os

import numpy as np
import torch
import yaml

_TYPE_MAX = {"uint8": 255, "uint12": 4098, "uint16": 65535, "float32": 65535, "float64": 65535, "int16": 65535}


def get_type_max(data):
    """utils/tool.py:8-24"""
    name = data.dtype.name if hasattr(data.dtype, "name") else str(data.dtype).replace("torch.", "")
    if name not in _TYPE_MAX:
        raise NotImplementedError(name)
    return _TYPE_MAX[name]


def range_limit(data, rng):
    """utils/tool.py:26-30"""
    l, h = rng
    assert l >= 0 and l <= h and h <= get_type_max(data), "Improper range setting!"
    return [l, h]


def normalize_data(data, name, min=None, max=None):
    """utils/io.py:65-112 -> (torch.float32 tensor, sideinfos)"""
    if "minmaxany" in name:
        scale_min, scale_max = (float(v) for v in name.split("_")[1:])
        dtype = data.dtype.name
        data = data.astype(np.float32)
        if min is None:
            min = float(data.min())
        if max is None:
            max = float(data.max())
        data = (data - min) / (max - min)
        data *= (scale_max - scale_min)
        data += scale_min
        data = torch.tensor(data, dtype=torch.float)
        return data, {"dtype": dtype, "min": min, "max": max, "normalized_min": data.min().item(), "normalized_max": data.max().item()}
    if name == "minmax01_0mean":
        dtype = data.dtype.name
        data = data.astype(np.float32)
        min, max = float(data.min()), float(data.max())
        data = (data - min) / (max - min)
        mean = data.mean()
        data = torch.tensor(data - mean, dtype=torch.float)
        return data, {"dtype": dtype, "min": min, "max": max, "mean": mean, "normalized_min": -mean, "normalized_max": 1 - mean}
    if name == "minmax01_0mean1std":
        dtype = data.dtype.name
        data = data.astype(np.float32)
        min, max = float(data.min()), float(data.max())
        data = (data - min) / (max - min)
        mean, std = data.mean(), data.std()
        data = torch.tensor((data - mean) / std, dtype=torch.float)
        return data, {"dtype": dtype, "min": min, "max": max, "mean": mean, "std": std,
                      "normalized_min": (-mean) / std, "normalized_max": (1 - mean) / std}
    if name == "none":
        dtype = data.dtype.name
        data = data.astype(np.float32)
        min, max = float(data.min()), float(data.max())
        return torch.tensor(data, dtype=torch.float), {"dtype": dtype, "min": min, "max": max, "normalized_min": min, "normalized_max": max}
    raise NotImplementedError(name)


def normalize_data_device(data, name, device, min=None, max=None):
    """normalize_data for the 'minmaxany_a_b' rule with the arithmetic on `device`: the same float32 operations in the same
    order as the host version (utils/io.py:65-80: cast, subtract, divide, scale, shift — each its own rounding, IEEE on both
    sides), so the result is bit-identical (tests/test_gpu_framework.py) while a 512^3 volume stops costing eight host passes
    over half a gigabyte.  Returns (device float32 tensor, sideinfos); other rules and dtypes go through normalize_data."""
    if "minmaxany" not in name or data.dtype not in (np.uint8, np.uint16, np.int16, np.float32):
        t, side = normalize_data(data, name, min=min, max=max)
        return t.to(device), side
    scale_min, scale_max = (float(v) for v in name.split("_")[1:])
    dtype = data.dtype.name
    a = np.ascontiguousarray(data)
    if a.dtype == np.uint16:      # torch has no uint16 arithmetic: move the bits, widen on the device
        x = (torch.from_numpy(a.view(np.int16)).to(device).to(torch.int32) & 0xFFFF).to(torch.float32)
    else:
        x = torch.from_numpy(a).to(device).to(torch.float32)
    if min is None:
        min = float(x.min().item())
    if max is None:
        max = float(x.max().item())
    # (operands as 0-d DEVICE tensors: torch turns `tensor / python_scalar` into a multiplication by the reciprocal on the GPU)
    c = lambda v: torch.tensor(np.float32(v), dtype=torch.float32, device=device)
    x -= c(min)
    x /= c(np.float32(np.float32(max) - np.float32(min)))
    x *= c(scale_max - scale_min)
    x += c(scale_min)
    return x, {"dtype": dtype, "min": min, "max": max, "normalized_min": x.min().item(), "normalized_max": x.max().item()}


def invnormalize_data(data, sideinfos, name):
    """utils/io.py:114-214 (data: torch tensor, modified in place like the reference)"""
    dtype = sideinfos["dtype"]
    if dtype not in ("uint8", "uint16", "float32", "float64", "int16"):
        raise NotImplementedError(dtype)
    min, max = sideinfos["min"], sideinfos["max"]
    if "minmaxany" in name:
        scale_min, scale_max = (float(v) for v in name.split("_")[1:])
        data -= scale_min
        data /= (scale_max - scale_min)
        data = torch.clip(data, 0, 1)
        data = data * (max - min) + min
    elif name == "minmax01":
        data = torch.clip(data, 0, 1) * (max - min) + min
    elif name == "minmaxn11":
        data = (torch.clip(data, -1, 1) / 2 + 0.5) * (max - min) + min
    elif name == "minmax01_0mean":
        data = torch.clip(data + sideinfos["mean"], 0, 1) * (max - min) + min
    elif name == "minmax01_0mean1std":
        data = torch.clip(data * sideinfos["std"] + sideinfos["mean"], 0, 1) * (max - min) + min
    elif name == "none":
        data = torch.clip(data, min, max)
    else:
        raise NotImplementedError(name)
    return data.numpy().astype(dtype)      # truncating cast, as np.array(tensor, dtype=...) does


def minmaxany_range(name):
    """(a, b) of 'minmaxany_a_b', the only normalisation the fused decode epilogue implements"""
    if "minmaxany" not in name:
        return None
    a, b = (float(v) for v in name.split("_")[1:])
    return a, b


def get_folder_size(folder_path):
    """utils/io.py get_folder_size: bytes of every file below the folder"""
    total = 0
    for root, _, files in os.walk(folder_path):
        for f in files:
            total += os.path.getsize(os.path.join(root, f))
    return total


def _plain(o):
    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    return o


def save_yaml(obj, path):
    with open(path, "w") as f:
        yaml.safe_dump(_plain(obj), f, sort_keys=False)


def load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f)
