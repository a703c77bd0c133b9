"""Whole-job wall clock of a SingleTask on the bench volume THROUGH the product's file interface: a 512^3 uint16 .tif on disk ->
read_img -> preprocess / normalise on the host -> H2D -> 2000 fused steps -> artefact written -> decode -> PSNR / SSIM on the GPU
-> decoded .tif written.  (bench.py's `value` starts with the volume resident in HBM; this is the number with disk, host and
PCIe included.)      python tools/e2e_singletask.py [steps] [edge] [c2|default]   -> gpurun_out/e2e_singletask[_<edge>_<mode>].json
mode c2 (default): the bench's 4x256 net (given_size); mode default: the shipped default.yaml's own budget (filesize_ratio 80, layers / w0 as shipped) with
100 000-voxel randompoint steps — on a 1024^3 volume that is the 4x1494 net, 20 000 steps = the yaml's max_steps."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np, torch
from brief_pytorch_amd import config
from brief_pytorch_amd.framework import NFGR, MyLogger
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch
from brief_pytorch_amd.tool import save_img

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else "c2"
work = tempfile.mkdtemp(prefix="brief_e2e_")
path = os.path.join(work, "volume.tif" if E < 1024 else "volume.npy")      # (a 2 GB volume: .npy, memory-mapped by read_img)
tg = time.perf_counter()
save_img(path, make_volume_torch((E, E, E), seed=42, device="cuda").cpu().numpy())
print("volume written: %.1f s" % (time.perf_counter() - tg), flush=True)
opt = config.load("opt/SingleTask/default.yaml")
cf = opt.CompressFramework
if mode == "c2":
    cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 4.0 * SIREN.calc_param_count(3, 1, 256, 5)
cf.Compress.max_steps, cf.Compress.checkpoints, cf.Compress.loss_log_freq = steps, "none", 10 ** 9
cf.Compress.sampler.name, cf.Compress.sampler.sample_size = "randompoint", 100000
if mode == "c2":
    cf.Module.phi.layers, cf.Module.phi.w0 = 5, 20
cf.Decompress.mip = False
cf["_seed"] = 42
Log = MyLogger(outputs_dir=work, project_name="e2e", time=False)
torch.manual_seed(42)
fw = NFGR(cf, Log=Log)
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx = fw.prepare_fit(path)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("prepared: %.1f s" % (t1 - t0), flush=True)
done = 0
while done < steps:                      # (in runs of 2000 steps with a progress line: run(a) + run(b) == run(a + b) bit for bit)
    k = min(2000, steps - done)
    loss = ctx["fit"].run(k)
    torch.cuda.synchronize()
    done += k
    print("step %d / %d  loss %.6g  %.1f s" % (done, steps, float(loss), time.perf_counter() - t1), flush=True)
t2 = time.perf_counter()
fw.checkpoint(ctx, steps, loss, True)
torch.cuda.synchronize()
t3 = time.perf_counter()
vox = float(E) ** 3
out = {"mode": mode, "net": "%d x %d" % (ctx["fit"].m.layers - 1, ctx["fit"].m.features), "volume": [E, E, E], "file_MB": os.path.getsize(path) / 1e6, "steps": steps,
       "prepare_s (read .tif, preprocess, normalise, weights, H2D, net init)": t1 - t0,
       "fit_s": t2 - t1, "checkpoint_s (artefact, decode, PSNR / SSIM on the GPU, decoded .tif)": t3 - t2,
       "total_s": t3 - t0, "encode_voxels_per_s (prepare + fit)": vox / (t2 - t0), "whole_job_voxels_per_s": vox / (t3 - t0),
       "perf": {k: float(v) for k, v in ctx["results"][steps].items()}}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/e2e_singletask%s.json" % ("" if (E, mode) == (512, "c2") else "_%d_%s" % (E, mode)), "w"), indent=1)
shutil.rmtree(work, ignore_errors=True)
