"""PSNR / SSIM against bitrate on a synthetic uint16 volume (BASELINE metric: "PSNR@bitrate").

    python tools/rate_distortion.py --size 256 --steps 2000 20000 --ratios 1024 256 64 --out gpurun_out/rd.md
    python tools/rate_distortion.py --size 512 --steps 2000 20000 --features 64 128 256 384 512 --detail 0 --ssim --out gpurun_out/rd.md

For every compression ratio the network width is solved from the byte budget exactly as NFGR does
(utils/Networks.py:299-314 -> SIREN.calc_features), the fit runs through the fused path (Fitter ==
main.py:385-400 with the randompoint sampler of main.py:126-163) and the decoded uint16 volume is scored on
the device (brief_sse_u16 / brief_ssim_u16).  bits/voxel = 32 P / voxels (fp32 weights, side info excluded).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brief_pytorch_amd import _lib                           # noqa: E402
from brief_pytorch_amd import metrics                        # noqa: E402
from brief_pytorch_amd.fit import Fitter                     # noqa: E402
from brief_pytorch_amd.networks import SIREN                 # noqa: E402
from brief_pytorch_amd.synthetic import make_volume_torch    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--w0", type=float, default=20.0)
    ap.add_argument("--steps", type=int, nargs="+", default=[2000, 20000])
    ap.add_argument("--ratios", type=float, nargs="+", default=[1024, 512, 256, 128, 64])
    ap.add_argument("--features", type=int, nargs="+", default=[], help="explicit net widths instead of --ratios (the ratio column is then derived)")
    ap.add_argument("--sample-size", type=int, default=100000)
    ap.add_argument("--ssim", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--detail", type=int, default=64, help="1/f texture components added to the synthetic field (0 = the bench volume)")
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    n = a.size
    dims = (n, n, n)
    vox = n ** 3
    vol = make_volume_torch(dims, seed=42, detail=a.detail)                  # (d,h,w,1) uint16 on the device
    vf = vol.to(torch.int32).to(torch.float32)
    vmin, vmax = float(vf.min()), float(vf.max())
    tv = ((vf - vmin) / (vmax - vmin) * 100.0).reshape(vox, 1).contiguous()      # minmaxany_0_100 (utils/io.py:65-80)
    del vf
    # the same field without its N(0, 200) noise: PSNR against it separates a good fit from a bad one (against the noisy source every codec is capped at 50.3 dB)
    clean = make_volume_torch(dims, seed=42, detail=a.detail, noise_sigma=0.0)
    lines = ["| ratio | features | params | bits/voxel | steps | fit s | Mvoxel-samples/s | PSNR dB | PSNR vs noise-free field dB |%s" % (" SSIM |" if a.ssim else ""),
             "|---|---|---|---|---|---|---|---|---|%s" % ("---|" if a.ssim else "")]
    jobs = [(vox * 2.0 / (4.0 * SIREN.calc_param_count(3, 1, F, a.layers)), F) for F in a.features] if a.features else \
           [(ratio, SIREN.calc_features(vox * 2 / ratio / 4, 3, 1, a.layers)) for ratio in a.ratios]
    for ratio, F in jobs:
        if F > 4096:
            print("ratio %g needs %d features (> 4096): skipped" % (ratio, F), flush=True)
            continue
        torch.manual_seed(42)
        m = SIREN(coords_channel=3, data_channel=1, features=F, layers=a.layers, w0=a.w0, precision=a.precision).to("cuda")
        fit = Fitter(m, tv, dims, sampler="randompoint", sample_size=a.sample_size, seed=42)
        done, t_fit = 0, 0.0
        for target in sorted(a.steps):
            torch.cuda.synchronize()
            t0 = time.time()
            fit.run(target - done)
            torch.cuda.synchronize()
            t_fit += time.time() - t0
            done = target
            dec = m.decode_grid(dims, out_kind="u16", scale=(0.0, 100.0), vrange=(vmin, vmax))     # fused invnormalize
            sse = torch.zeros(1, dtype=torch.float64, device="cuda")
            _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(vol), _lib.ptr(dec), vox, _lib.ptr(sse), _lib.stream_ptr()))
            psnr = -10.0 * np.log10(sse.item() / vox / 65535.0 ** 2)
            _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(clean), _lib.ptr(dec), vox, _lib.ptr(sse), _lib.stream_ptr()))
            psnr_c = -10.0 * np.log10(sse.item() / vox / 65535.0 ** 2)
            if a.ssim:
                ss, ns = metrics.gpu_ssim_u16(vol.reshape(dims), dec.reshape(dims))
            row = "| %.4g | %d | %d | %.4f | %d | %.1f | %.1f | %.2f | %.2f |" % (ratio, F, m.param_count, 32.0 * m.param_count / vox, done, t_fit,
                                                                              done * fit.n / t_fit / 1e6, psnr, psnr_c)
            if a.ssim:
                row += " %.4f |" % (ss / ns)
            print(row, flush=True)
            lines.append(row)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
