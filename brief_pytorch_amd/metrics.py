"""MSE / PSNR / SSIM as the reference evaluates them (utils/misc.py:447-499, utils/ssim.py:9-150).

Host (numpy/torch-CPU free) implementations used by the CLI; the GPU path gets the SSE for PSNR
from brief_sse_u16 so that multi-GPU DivideTask only needs to all-reduce [SSE, n].
"""
import numpy as np

from .io import get_type_max


def cal_mse(data1, data2):
    return ((data1 - data2) ** 2).mean()


def cal_psnr(origin_data, decompressed_data, data_range):
    mse = np.mean(np.power(origin_data / data_range - decompressed_data / data_range, 2))
    return -10 * np.log10(mse)


def psnr_from_sse(sse, count, data_range):
    return float(-10.0 * np.log10(sse / count / float(data_range) ** 2))


def _gauss_win(size=11, sigma=1.5):
    c = np.arange(size, dtype=np.float32) - size // 2
    g = np.exp(-(c ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (g / g.sum()).astype(np.float32)


def _filt(x, win):
    k = win.size
    out = x
    if out.shape[0] >= k:
        out = sum(win[i] * out[i:out.shape[0] - k + 1 + i, :] for i in range(k))
    if out.shape[1] >= k:
        out = sum(win[i] * out[:, i:out.shape[1] - k + 1 + i] for i in range(k))
    return out


def ssim2d(x, y, data_range, K=(0.01, 0.03)):
    """utils/ssim.py:55-92 for one (h, w) channel: 11-tap sigma-1.5 separable Gaussian, valid padding"""
    win = _gauss_win()
    x, y = x.astype(np.float32), y.astype(np.float32)
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filt(x, win), _filt(y, win)
    s1 = _filt(x * x, win) - mu1 * mu1
    s2 = _filt(y * y, win) - mu2 * mu2
    s12 = _filt(x * y, win) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    return float((((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs).mean(dtype=np.float64))


def cal_ssim(origin_data, decompressed_data, data_range):
    """utils/misc.py:458-475: (h,w,c) -> channel mean; (d,h,w,c) -> mean over d of the per-slice value"""
    a, b = np.asarray(origin_data), np.asarray(decompressed_data)
    if a.ndim == 3:
        return float(np.mean([ssim2d(a[..., c], b[..., c], data_range) for c in range(a.shape[-1])]))
    tot = 0.0
    for i in range(a.shape[0]):
        tot += float(np.mean([ssim2d(a[i, ..., c], b[i, ..., c], data_range) for c in range(a.shape[-1])]))
    return float(tot / a.shape[0])


def eval_performance(steps, data1, data2, Log, mse, psnr, ssim):
    """utils/misc.py:477-499"""
    perf = {"steps": steps}
    max_range = get_type_max(data1)
    data1 = data1.astype(np.float32)
    data2 = data2.astype(np.float32)
    if mse:
        perf["mse"] = cal_mse(data1, data2)
        if Log is not None:
            Log.log_metrics({"mse": perf["mse"]}, steps)
    if psnr:
        perf["psnr"] = cal_psnr(data1, data2, max_range)
        if Log is not None:
            Log.log_metrics({"psnr": perf["psnr"]}, steps)
    if ssim:
        perf["ssim"] = cal_ssim(data1, data2, max_range)
        if Log is not None:
            Log.log_metrics({"ssim": perf["ssim"]}, steps)
    return perf


def gpu_ssim_u16(a, b, data_range=65535):
    """cal_ssim for single-channel uint16 volumes resident on the GPU: tensors (d,h,w) or (d,h,w,1).
    Returns (sum of per-slice means, slices) so that ranks can all-reduce both; SSIM = sum / slices."""
    import ctypes as C
    import torch
    from . import _lib
    d, h, w = a.shape[:3]
    L = _lib.lib()
    n = L.brief_ssim_partial_count(d, h, w)
    if n < 0:
        raise _lib.BriefError("gpu_ssim_u16 needs h, w >= 11")
    part = torch.empty(n, dtype=torch.float64, device=a.device)
    win = torch.from_numpy(_gauss_win()).to(a.device)
    _lib.check(L.brief_ssim_u16(_lib.ptr(a), _lib.ptr(b), d, h, w, _lib.ptr(win), float(data_range), _lib.ptr(part), n, _lib.stream_ptr()))
    per_slice = part.view(d, -1).sum(1) / float((h - 10) * (w - 10))
    return float(per_slice.sum().item()), d


def gpu_eval_u16(orig, dec, mse=True, psnr=True, ssim=True):
    """eval_performance (utils/misc.py:477-499) for single-channel uint16 volumes on the GPU: SSE by
    brief_sse_u16, SSIM by brief_ssim_u16.  orig/dec: numpy (d,h,w,1)."""
    import torch
    from . import _lib
    a = torch.from_numpy(np.ascontiguousarray(orig)).cuda()
    b = torch.from_numpy(np.ascontiguousarray(dec)).cuda()
    out = {}
    if mse or psnr:
        sse = torch.zeros(1, dtype=torch.float64, device=a.device)
        _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(sse), _lib.stream_ptr()))
        v = sse.item()
        if mse:
            out["mse"] = v / a.numel()
        if psnr:
            out["psnr"] = psnr_from_sse(v, a.numel(), 65535)
    if ssim:
        s, n = gpu_ssim_u16(a, b)
        out["ssim"] = s / n
    return out
