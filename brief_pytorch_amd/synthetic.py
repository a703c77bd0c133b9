"""Deterministic synthetic biomedical-like volumes.

The reference ships one 64^3 uint16 brain crop and lists its larger volumes as
missing blobs, so every benchmark / parity volume is generated here: a few
band-limited 3-D sinusoids, Gaussian blobs, tubular (vessel-like) curves and
additive noise on an offset that mimics the shipped sample's range
(16 633 ... 24 070 counts).  Generation is slab-wise along the first axis so a
1024^3 volume never needs a float copy of the whole array.
"""
import numpy as np

__all__ = ["make_volume", "make_vessel_volume", "make_volume_torch", "ensure_dataset"]


def make_volume(shape, seed=42, dtype=np.uint16, n_waves=8, n_blobs=32, n_tubes=16,
                noise_sigma=200.0, base=16000.0, span=8000.0, slab=16,
                wave_gain=0.35, blob_gain=0.25, tube_gain=0.5, floor=0.5):
    """Return a (d, h, w, 1) volume of `dtype` (the reference's 3-D layout,
    utils/tool.py:73-92: tif stacks are read as (d,h,w) and get a channel axis)."""
    d, h, w = (int(s) for s in shape)
    rng = np.random.default_rng(seed)
    # sinusoid bank: integer cycle counts <= 6 per axis
    freqs = rng.integers(0, 7, size=(n_waves, 3)).astype(np.float64)
    phases = rng.uniform(0, 2 * np.pi, size=n_waves)
    amps = rng.uniform(0.3, 1.0, size=n_waves)
    amps /= amps.sum()
    # blobs: centre (unit cube), radius, amplitude
    bc = rng.uniform(0, 1, size=(n_blobs, 3))
    br = rng.uniform(0.03, 0.12, size=n_blobs)
    ba = rng.uniform(0.2, 0.9, size=n_blobs)
    # tubes: straight segments p0->p1 with a radius
    t0 = rng.uniform(0, 1, size=(n_tubes, 3))
    t1 = rng.uniform(0, 1, size=(n_tubes, 3))
    tr = rng.uniform(0.008, 0.02, size=n_tubes)
    ta = rng.uniform(0.5, 1.0, size=n_tubes)
    info = np.iinfo(dtype) if np.issubdtype(dtype, np.integer) else None
    out = np.empty((d, h, w, 1), dtype=dtype)
    y = (np.arange(h, dtype=np.float64) / max(h - 1, 1))[None, :, None]
    x = (np.arange(w, dtype=np.float64) / max(w - 1, 1))[None, None, :]
    for z0 in range(0, d, slab):
        z1 = min(z0 + slab, d)
        z = (np.arange(z0, z1, dtype=np.float64) / max(d - 1, 1))[:, None, None]
        f = np.zeros((z1 - z0, h, w), dtype=np.float64)
        for k in range(n_waves):
            f += amps[k] * np.sin(2 * np.pi * (freqs[k, 0] * z + freqs[k, 1] * y + freqs[k, 2] * x) + phases[k])
        f = floor + wave_gain * f
        for k in range(n_blobs):
            r2 = (z - bc[k, 0]) ** 2 + (y - bc[k, 1]) ** 2 + (x - bc[k, 2]) ** 2
            f += ba[k] * blob_gain * np.exp(-r2 / (2 * br[k] ** 2))
        for k in range(n_tubes):
            a, b = t0[k], t1[k]
            ab = b - a
            den = float(ab @ ab) + 1e-12
            tt = ((z - a[0]) * ab[0] + (y - a[1]) * ab[1] + (x - a[2]) * ab[2]) / den
            tt = np.clip(tt, 0.0, 1.0)
            r2 = (z - (a[0] + tt * ab[0])) ** 2 + (y - (a[1] + tt * ab[1])) ** 2 + (x - (a[2] + tt * ab[2])) ** 2
            f += ta[k] * tube_gain * np.exp(-r2 / (2 * tr[k] ** 2))
        # per-slab noise stream keyed by (seed, z0) so slab size does not change the volume
        nrng = np.random.default_rng([seed, z0 // slab, 7])
        v = base + span * f + nrng.normal(0.0, noise_sigma, size=f.shape)
        if info is not None:
            v = np.clip(np.rint(v), info.min, info.max)
        out[z0:z1, :, :, 0] = v.astype(dtype)
    return out


def make_vessel_volume(shape, seed=42, dtype=np.uint16):
    """Sparse, vessel-like variant (BASELINE config 5: the shape of the reference's
    dataset/example/vessel-0_64-0_512-0_512.tif): dark, nearly flat background, ~5 % bright tubular
    foreground, little noise — the statistics `adaptotal`/`by_var` budgeting is meant for."""
    return make_volume(shape, seed=seed, dtype=dtype, n_waves=3, n_blobs=6, n_tubes=32, noise_sigma=60.0,
                       base=1200.0, span=9000.0, wave_gain=0.02, blob_gain=0.05, tube_gain=3.0, floor=0.05)


def make_volume_torch(shape, seed=42, device="cuda", n_waves=8, n_blobs=32, noise_sigma=200.0, base=16000.0, span=8000.0, slab=32,
                      detail=0, detail_fmax=64.0, detail_gain=0.6):
    """Same kind of field as make_volume, generated on the device slab by slab (512^3 in about a
    second) for the benchmark volumes.  Returns a uint16 torch tensor (d, h, w, 1).  The field
    parameters come from numpy's generator so they do not depend on the torch build.
    `detail` > 0 adds that many oriented sinusoids with |f| log-uniform in [6, detail_fmax] cycles and
    amplitude ~ 1/|f| (a 1/f texture): the plain field is fitted down to its noise floor by any net, the
    textured one makes PSNR depend on the bitrate (tools/rate_distortion.py)."""
    import torch
    d, h, w = (int(s) for s in shape)
    rng = np.random.default_rng(seed)
    freqs = rng.integers(0, 7, size=(n_waves, 3)).astype(np.float32)
    phases = rng.uniform(0, 2 * np.pi, size=n_waves).astype(np.float32)
    amps = rng.uniform(0.3, 1.0, size=n_waves)
    amps = (amps / amps.sum()).astype(np.float32)
    bc = rng.uniform(0, 1, size=(n_blobs, 3)).astype(np.float32)
    br = rng.uniform(0.03, 0.12, size=n_blobs).astype(np.float32)
    ba = rng.uniform(0.2, 0.9, size=n_blobs).astype(np.float32)
    if detail:
        drng = np.random.default_rng([seed, 99])
        dmag = np.exp(drng.uniform(np.log(6.0), np.log(detail_fmax), size=detail))
        ddir = drng.normal(size=(detail, 3))
        ddir /= np.linalg.norm(ddir, axis=1, keepdims=True)
        dfreq = (ddir * dmag[:, None]).astype(np.float32)
        dph = drng.uniform(0, 2 * np.pi, size=detail).astype(np.float32)
        damp = (1.0 / dmag)
        damp = (detail_gain * damp / np.sqrt((damp ** 2).sum())).astype(np.float32)
    out = torch.empty((d, h, w, 1), dtype=torch.uint16, device=device)
    y = (torch.arange(h, device=device, dtype=torch.float32) / max(h - 1, 1))[None, :, None]
    x = (torch.arange(w, device=device, dtype=torch.float32) / max(w - 1, 1))[None, None, :]
    gen = torch.Generator(device=device)
    for z0 in range(0, d, slab):
        z1 = min(z0 + slab, d)
        z = (torch.arange(z0, z1, device=device, dtype=torch.float32) / max(d - 1, 1))[:, None, None]
        f = torch.zeros((z1 - z0, h, w), device=device)
        for k in range(n_waves):
            f += float(amps[k]) * torch.sin(6.283185307179586 * (float(freqs[k, 0]) * z + float(freqs[k, 1]) * y + float(freqs[k, 2]) * x) + float(phases[k]))
        f = 0.5 + 0.35 * f
        for k in range(detail):
            f += float(damp[k]) * torch.sin(6.283185307179586 * (float(dfreq[k, 0]) * z + float(dfreq[k, 1]) * y + float(dfreq[k, 2]) * x) + float(dph[k]))
        for k in range(n_blobs):
            r2 = (z - float(bc[k, 0])) ** 2 + (y - float(bc[k, 1])) ** 2 + (x - float(bc[k, 2])) ** 2
            f += float(ba[k]) * 0.25 * torch.exp(-r2 / (2 * float(br[k]) ** 2))
        gen.manual_seed(seed * 1000003 + z0)
        v = base + span * f + noise_sigma * torch.randn(f.shape, device=device, generator=gen)
        out[z0:z1, :, :, 0] = v.round().clamp_(0, 65535).to(torch.int32).to(torch.uint16)
    return out


def ensure_dataset(path):
    """`dataset/synthetic_<n>.tif`, `synthetic_<d>x<h>x<w>.tif` and `synthetic_vessel_<d>x<h>x<w>.tif` are
    generated on first use; any other missing path is an error.  The reference's sample volumes are not
    redistributed here."""
    import os
    import re
    if os.path.exists(path):
        return path
    m = re.fullmatch(r"synthetic_(vessel_)?(\d+)(?:x(\d+)x(\d+))?\.tiff?", os.path.basename(path))
    if not m:
        raise FileNotFoundError(path)
    d = int(m.group(2))
    shape = (d, int(m.group(3)), int(m.group(4))) if m.group(3) else (d, d, d)
    from .tool import save_img
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    save_img(path, (make_vessel_volume if m.group(1) else make_volume)(shape, seed=42))
    return path


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a synthetic uint16 test volume")
    ap.add_argument("--shape", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", type=str, required=True)
    a = ap.parse_args()
    from .tool import save_img
    save_img(a.out, make_volume(a.shape, seed=a.seed))
    print("wrote", a.out)
