"""BASELINE config 5 on one GPU: the vessel-shaped DivideTask (opt/DivideTask/vessel.yaml: adaptotal <= 4 blocks,
by_size budget, 7-layer nets, w0 = 10) on the synthetic 64x512x512 vessel stack, swept over compression ratios.

    python tools/vessel_sweep.py --steps 4000 --ratios 512 256 128 64 --out gpurun_out/vessel_sweep.md

Each ratio is one NFGR.compress_divide run (partition -> budget -> co-trained per-block fits -> per-block decode ->
merge -> PSNR from the summed SSE, SSIM on the merged volume), exactly what `python main.py -p opt/DivideTask/vessel.yaml`
does, with max_steps reduced so that the sweep fits a few GPU-minutes."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brief_pytorch_amd import config                      # noqa: E402
from brief_pytorch_amd.framework import NFGR, MyLogger    # noqa: E402
from brief_pytorch_amd.synthetic import ensure_dataset    # noqa: E402
from brief_pytorch_amd.tool import read_img               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--ratios", type=float, nargs="+", default=[512, 256, 128, 64])
    ap.add_argument("--shape", type=str, default="64x512x512")
    ap.add_argument("--precision", type=str, default="fp32")
    ap.add_argument("--divide", type=str, default="", help="override Compress.divide.divide_type (e.g. adaptive_-1_-1_0_0_20: octree + tree knapsack, mixed block sizes)")
    ap.add_argument("--alloc", type=str, default="", help="override Compress.divide.param_alloc (equal | by_size | by_var | by_d | by_dv)")
    ap.add_argument("--out", type=str, default="")
    ap.add_argument("--workdir", type=str, default="/tmp/vessel_sweep")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    t0 = time.time()
    path = ensure_dataset(os.path.join(a.workdir, "dataset", "synthetic_vessel_%s.tif" % a.shape))
    vol = read_img(path)
    print("volume %s %s, %.1f %% above 4000 counts, generated/read in %.0f s" % (vol.shape, vol.dtype, 100 * (vol > 4000).mean(), time.time() - t0), flush=True)
    lines = ["| ratio | blocks | features per block | bits/voxel | steps | fit s (all blocks, co-trained) | PSNR dB | SSIM |", "|---|---|---|---|---|---|---|---|"]
    for ratio in a.ratios:
        opt = config.load(os.path.join(ROOT, "opt", "DivideTask", "vessel.yaml"))
        opt.Dataset.data_path = path
        cf = opt.CompressFramework
        cf.Compress.max_steps = a.steps
        cf.Compress.checkpoints = "none"
        cf.Compress.param.filesize_ratio = ratio
        cf.Compress.precision = a.precision
        if a.divide:
            cf.Compress.divide.divide_type = a.divide
        if a.alloc:
            cf.Compress.divide.param_alloc = a.alloc
        cf.Decompress.mip = False
        cf.Decompress.keep_decompressed = False
        opt.Log.outputs_dir = os.path.join(a.workdir, "outputs")
        opt.Log.project_name = "vessel_r%g" % ratio
        opt.Log.time = False
        Log = MyLogger(**opt.Log)
        torch.manual_seed(42)
        fw = NFGR(cf, Log=Log)
        res = fw.compress_divide(path, opt)[a.steps]
        cdir = os.path.join(Log.logdir, "steps%d" % a.steps, "compressed")
        names = sorted(os.listdir(os.path.join(cdir, "sideinfos")))
        feats = [config.load(os.path.join(cdir, "sideinfos", n, "sideinfos.yaml"))["phi_features"] for n in names]
        nbytes = sum(os.path.getsize(os.path.join(dp, f)) for n in names for dp, _, fs in os.walk(os.path.join(cdir, "module", n)) for f in fs)
        from collections import Counter
        from brief_pytorch_amd.misc import parse_chunk_name
        shapes = Counter("x".join(str(r[k][1] - r[k][0] + 1) for k in "dhw") for r in map(parse_chunk_name, names))
        print("   block shapes:", dict(shapes), flush=True)
        feat_txt = ",".join(str(f) for f in feats) if len(feats) <= 8 else "%d..%d (%s)" % (min(feats), max(feats), ", ".join("%d x %s" % (c, s_) for s_, c in sorted(shapes.items())))
        row = "| %g | %d | %s | %.4f | %d | %.2f | %.2f | %.4f |" % (ratio, len(names), feat_txt, 8.0 * nbytes / vol.size,
                                                                   a.steps, fw.fit_seconds, res["psnr"], res.get("ssim", float("nan")))
        print(row, flush=True)
        lines.append(row)
        Log.close()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
