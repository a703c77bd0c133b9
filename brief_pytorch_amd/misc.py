"""Host-side helpers around the fit loop: checkpoint schedule, loss-weight maps, pre-processing,
block partition, parameter budget and merge (reference utils/misc.py:233-445,
utils/adaptive_blocking.py:16-24, 425-459).  Pure numpy; pinned to golden vectors.
"""
import os

import numpy as np

from .io import get_type_max, range_limit


# ---------------------------------------------------------------- schedules / weights / preprocess
def parse_checkpoints(checkpoints, max_steps):
    """utils/misc.py:255-271.  'none' | 'every_n' | 'a,b,c' (the reference raises on ints)."""
    if checkpoints == "none":
        return [max_steps]
    if "every" in checkpoints:
        interval = int(checkpoints.split("_")[1])
        return list(range(interval, max_steps, interval)) + [max_steps]
    return [int(s) for s in checkpoints.split(",") if int(s) < max_steps] + [max_steps]


def weight_is_unit(weight_type_list):
    """True when parse_weight() must return all ones whatever the data holds: only 'none' and 'value_l_h_s' entries with
    s == 1 (the shipped YAMLs: 'value_65535_65535_1').  Lets the caller skip building (and scanning) a volume-sized map."""
    for spec in weight_type_list:
        if spec == "none":
            continue
        if "quantile" in spec or "exp" in spec or "value" not in spec:
            return False
        if float(spec.split("_")[3]) != 1.0:
            return False
    return True


def parse_weight(data, weight_type_list):
    """utils/misc.py:272-307: per-voxel loss weights from 'value_l_h_s' / 'quantile_t_ql_qh_s' / 'exp_x_v' / 'none'"""
    data = np.asarray(data)
    weight = np.ones_like(data).astype(np.float32)
    for spec in weight_type_list:
        if "quantile" in spec:
            _, ge, ql, qh, scale = spec.split("_")
            sel = data[data >= float(ge)]
            l, h = range_limit(data, [np.quantile(sel, float(ql)), np.quantile(sel, float(qh))])
            weight[(data >= l) * (data <= h)] = float(scale)
        elif "value" in spec:
            _, l, h, scale = spec.split("_")
            l, h = range_limit(data, [float(l), float(h)])
            weight[(data >= l) * (data <= h)] = float(scale)
        elif "exp" in spec:
            _, mid_x, mid_value = spec.split("_")
            weight = np.exp(-(-np.log(float(mid_value)) / float(mid_x)) * data)
        elif spec != "none":
            raise NotImplementedError(spec)
    return weight




def preprocess_is_identity(data, denoise_level, denoise_close, clip_range):
    """True when preprocess() would return the data unchanged (the shipped defaults on unsigned data: level 0 only
    rewrites zeros with zeros, the clip covers the dtype's range)"""
    lo, hi = range_limit(data, clip_range)
    unsigned = np.issubdtype(data.dtype, np.unsignedinteger)
    return bool(unsigned and denoise_level <= 0 and lo <= 0 and hi >= np.iinfo(data.dtype).max)


def preprocess(data, denoise_level, denoise_close, clip_range):
    """utils/misc.py:244-254: threshold-denoise (optionally through a binary opening) and clip.  Unlike the reference
    this does NOT write into its argument: callers keep evaluating against the untouched original (the reference
    re-reads the file for that, main.py:433, 624); when nothing would change the input itself is returned."""
    if preprocess_is_identity(data, denoise_level, denoise_close, clip_range):
        return data
    data = np.array(data, copy=True)
    if denoise_close is False:
        data[data <= denoise_level] = 0
    else:
        from scipy import ndimage
        close = tuple(list(denoise_close if data.ndim == 4 else denoise_close[:2]) + [1])
        data[ndimage.binary_opening(data <= denoise_level, structure=np.ones(close), iterations=1)] = 0
    return data.clip(*range_limit(data, clip_range))


def mip_ops(data, save_dir=None, data_name="", suffix=""):
    """utils/misc.py:233-242: max-intensity projections along d, h, w, optionally written as
    <save_dir>/<data_name>_mip_{d,h,w}<suffix>"""
    assert data.ndim == 4
    mips = data.max(0), data.max(1), data.max(2)
    if save_dir is not None:
        save_mips(mips, save_dir, data_name, suffix)
    return mips


def save_mips(mips, save_dir, data_name, suffix):
    from .tool import save_img
    for ax, img in zip("dhw", mips):
        if suffix == ".png" and img.dtype not in (np.uint8, np.uint16):
            continue                                  # PNG holds 8 / 16-bit integers only
        save_img(os.path.join(save_dir, "%s_mip_%s%s" % (data_name, ax, suffix)), img)


# ---------------------------------------------------------------- partition
def chunk_name(c):
    if "d" in c:
        return "d_{}_{}-h_{}_{}-w_{}_{}".format(*c["d"], *c["h"], *c["w"])
    return "h_{}_{}-w_{}_{}".format(*c["h"], *c["w"])


def parse_chunk_name(name):
    """inverse of chunk_name: inclusive index ranges (main.py:306-311)"""
    out = {}
    for part in name.split("-"):
        k, a, b = part.split("_")
        out[k] = [int(a), int(b)]
    return out


def _sections(n, step):
    return [i for i in range(n) if i % step == 0] + [n]


def divide_data(data, divide_type):
    """utils/misc.py:329-394: 'total_nd_nh_nw' / 'every_d_h_w' fixed grid (remainder blocks allowed).
    Returns (chunks, outline image with block faces burnt in at 2000 for 3-D data)."""
    spec = [int(v) for v in divide_type.split("_")[1:]]
    ndim = data.ndim - 1
    sizes = data.shape[:ndim]
    spec = spec[-ndim:] if len(spec) >= ndim else spec
    if "total" in divide_type:
        steps = [int(sizes[a] / spec[a]) for a in range(ndim)]
    elif "every" in divide_type:
        steps = spec
    else:
        raise NotImplementedError(divide_type)
    secs = [_sections(sizes[a], steps[a]) for a in range(ndim)]
    outline = data.copy()
    chunks = []
    if ndim == 3:
        for di in range(len(secs[0]) - 1):
            for hi in range(len(secs[1]) - 1):
                for wi in range(len(secs[2]) - 1):
                    z0, z1 = secs[0][di], secs[0][di + 1]
                    y0, y1 = secs[1][hi], secs[1][hi + 1]
                    x0, x1 = secs[2][wi], secs[2][wi + 1]
                    chunks.append({"data": data[z0:z1, y0:y1, x0:x1], "d": [z0, z1 - 1], "h": [y0, y1 - 1], "w": [x0, x1 - 1]})
                    for sl in ((z0, slice(y0, y1), slice(x0, x1)), (z1 - 1, slice(y0, y1), slice(x0, x1)),
                               (slice(z0, z1), y0, slice(x0, x1)), (slice(z0, z1), y1 - 1, slice(x0, x1)),
                               (slice(z0, z1), slice(y0, y1), x0), (slice(z0, z1), slice(y0, y1), x1 - 1)):
                        outline[sl] = 2000
    else:
        for hi in range(len(secs[0]) - 1):
            for wi in range(len(secs[1]) - 1):
                y0, y1 = secs[0][hi], secs[0][hi + 1]
                x0, x1 = secs[1][wi], secs[1][wi + 1]
                chunks.append({"data": data[y0:y1, x0:x1], "h": [y0, y1 - 1], "w": [x0, x1 - 1]})
    for c in chunks:
        c["total_size"] = data.size
        c["size"] = c["data"].size
        c["name"] = chunk_name(c)
    return chunks, outline


def rgb2gray(img, order="rgb"):
    """cv2.cvtColor(img, COLOR_RGB2GRAY / COLOR_BGR2GRAY) for integer images: OpenCV's 14-bit fixed-point weights
    (R 4899, G 9617, B 1868) with rounding.  cv2 is not in this image, so this restatement is unpinned."""
    r, g, b = (img[..., 0], img[..., 1], img[..., 2]) if order == "rgb" else (img[..., 2], img[..., 1], img[..., 0])
    if np.issubdtype(img.dtype, np.integer):
        y = (r.astype(np.int64) * 4899 + g.astype(np.int64) * 9617 + b.astype(np.int64) * 1868 + 8192) >> 14
        return y.astype(img.dtype)
    return (0.299 * r + 0.587 * g + 0.114 * b).astype(img.dtype)


def cal_feature(image):
    """utils/adaptive_blocking.py:16-24: max|FFT| / sum|FFT| (both truncated to int); (d,h,w,c) data -> 3-D FFT,
    (h,w,3) images -> BGR2GRAY then 2-D FFT"""
    if image.ndim == 3:
        gray = rgb2gray(image, "bgr") if image.shape[-1] == 3 else image[..., 0]
        f = np.abs(np.fft.fft(np.fft.fft(gray, axis=0), axis=1))
    elif image.ndim == 4:
        f = np.abs(np.fft.fft(np.fft.fft(np.fft.fft(image, axis=0), axis=1), axis=2))
    else:
        raise NotImplementedError("cal_feature needs (h,w,c) or (d,h,w,c) data")
    return int(f.max()) / int(f.sum())


def alloc_param(data_chunk_list, param_size, param_alloc, param_size_thres):
    """utils/misc.py:395-428: split the byte budget over blocks; blocks whose share falls below
    param_size_thres are dropped and the budget is re-split (recursively)."""
    n = len(data_chunk_list)
    if param_alloc == "equal":
        for c in data_chunk_list:
            c["param_size"] = param_size / n
    elif param_alloc == "by_size":
        for c in data_chunk_list:
            c["param_size"] = param_size * c["size"] / c["total_size"]
    elif param_alloc in ("by_var", "by_d", "by_dv"):
        if param_alloc == "by_var":
            v = [((c["data"] - c["data"].mean()) ** 2).mean() for c in data_chunk_list]
        elif param_alloc == "by_d":
            v = [1 / cal_feature(c["data"]) for c in data_chunk_list]
        else:
            v = [c["size"] / cal_feature(c["data"]) for c in data_chunk_list]
        tot = 0
        for x in v:
            tot += x
        for c, x in zip(data_chunk_list, v):
            c["param_size"] = float(param_size * x / tot)
    else:
        raise NotImplementedError(param_alloc)
    kept = [c for c in data_chunk_list if c["param_size"] >= param_size_thres]
    if len(kept) < n:
        return alloc_param(kept, param_size, param_alloc, param_size_thres)
    return kept


def merge_divided_data(chunk_list, data_shape):
    """utils/misc.py:430-445: paste by inclusive ranges into float32 zeros (+=), clip to the dtype max, cast"""
    top = get_type_max(chunk_list[0]["data"])
    out = np.zeros(data_shape, dtype=np.float32)
    for c in chunk_list:
        h0, h1 = c["h"]
        w0, w1 = c["w"]
        if len(data_shape) == 4:
            d0, d1 = c["d"]
            out[d0:d1 + 1, h0:h1 + 1, w0:w1 + 1] += c["data"]
        else:
            out[h0:h1 + 1, w0:w1 + 1] += c["data"]
    return out.clip(None, top).astype(chunk_list[0]["data"].dtype)


def _proper_divisors(n):
    return [1] + [i for i in range(2, n) if n % i == 0]


def cal_divide_num(d, h, w, Nb, param_size):
    """utils/adaptive_blocking.py:425-459: (nd,nh,nw) among proper divisors maximising nd*nh*nw <= Nb,
    ties -> smallest variance of the block edge lengths (first found wins on equal variance)."""
    if Nb <= 0:
        Nb = max(int(param_size / (4 * 1361)), 1)
    best, best_num, best_var = None, 0, None
    for nd in _proper_divisors(d):
        for nh in _proper_divisors(h):
            for nw in _proper_divisors(w):
                num = nd * nh * nw
                if num > Nb:
                    continue
                size = np.array([d / nd, h / nh, w / nw])
                var = ((size - size.mean()) ** 2).mean()
                if num > best_num or (num == best_num and var < best_var):
                    best, best_num, best_var = (nd, nh, nw), num, var
    return best


def configure_optimizer(parameters, optimizer, lr):
    """utils/misc.py:174-183: a torch optimizer over `parameters` (for code that keeps the reference's loop and uses
    the module through autograd, see networks._SirenFn; the fused Fitter applies the same rules in-kernel)."""
    import torch
    if optimizer == "Adam":
        return torch.optim.Adam(parameters, lr=lr)
    if optimizer == "Adamax":
        return torch.optim.Adamax(parameters, lr=lr)
    if optimizer == "SGD":
        return torch.optim.SGD(parameters, lr=lr)
    raise NotImplementedError(optimizer)


def configure_lr_scheduler(optimizer, lr_scheduler_opt):
    """utils/misc.py:184-197"""
    import copy
    import torch
    o = copy.deepcopy(dict(lr_scheduler_opt))
    name = o.pop("name")
    if name == "MultiStepLR":
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, **o)
    if name == "CyclicLR":
        return torch.optim.lr_scheduler.CyclicLR(optimizer, **o)
    if name == "StepLR":
        return torch.optim.lr_scheduler.StepLR(optimizer, **o)
    if name == "none":
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[100000000000])
    raise NotImplementedError(name)
