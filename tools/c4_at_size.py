"""BASELINE config 4 at size on ONE GPU through the product path: a 1024^3 uint16 volume (2 GiB, written octant by octant),
`divide_type: adaptive_-1_-1_0_0_8` (octree + tree knapsack -> the eight 512^3 octants), eight 4x256 nets co-trained on this
GPU (brief_multi_fit), artefact tree, z-sharded evaluation (one slab here), PSNR / SSIM of the merged volume.
    python tools/c4_at_size.py [steps] [edge]       -> prints phase timings; gpurun_out/c4_at_size.json"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np, torch
from brief_pytorch_amd import config
from brief_pytorch_amd.framework import NFGR, MyLogger
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch
from brief_pytorch_amd.tool import create_stack, write_slab

def run(steps=200, E=1024, keep_json=True):
  H = E // 2
  work = tempfile.mkdtemp(prefix="brief_c4_")
  path = os.path.join(work, "volume.npy")
  t = {}
  t0 = time.perf_counter()
  mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint16, shape=(E, E, E, 1))
  for o in range(8):
      z, y, x = (o >> 2) & 1, (o >> 1) & 1, o & 1
      blk = make_volume_torch((H, H, H), seed=100 + o, device="cuda").cpu().numpy()
      mm[z * H:(z + 1) * H, y * H:(y + 1) * H, x * H:(x + 1) * H] = blk
  mm.flush(); del mm
  t["generate_s"] = time.perf_counter() - t0
  opt = config.load("opt/DivideTask/default.yaml")
  cf = opt.CompressFramework
  cf.Compress.divide.divide_type = "adaptive_-1_-1_0_0_8"
  cf.Compress.divide.param_alloc = "by_size"
  cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 8 * 4.0 * SIREN.calc_param_count(3, 1, 256, 5)
  cf.Compress.max_steps, cf.Compress.checkpoints, cf.Compress.loss_log_freq = steps, "none", 10 ** 9
  cf.Compress.sampler.name = "randompoint"
  cf.Decompress.keep_decompressed, cf.Decompress.mip = True, False
  cf["_seed"] = 42
  Log = MyLogger(outputs_dir=work, project_name="c4", time=False)
  torch.manual_seed(42)
  fw = NFGR(cf, Log=Log)
  t0 = time.perf_counter()
  res = fw.compress_divide(path, opt)
  torch.cuda.synchronize()
  t["compress_divide_s"] = time.perf_counter() - t0
  t["fit_s"] = fw.fit_seconds
  names = sorted(os.listdir(os.path.join(Log.logdir, "steps%d" % steps, "compressed", "module")))
  out = {"edge": E, "steps": steps, "blocks": names, "perf": {k: float(v) for k, v in res[steps].items()}, "timings": t,
         "fit_voxels_per_s": 8 * 100000 * steps / fw.fit_seconds, "peak_host_rss_gb": None}
  try:
      import resource
      out["peak_host_rss_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
  except Exception:
      pass
  dec = np.load(os.path.join(Log.logdir, "steps%d" % steps, "decompressed", "volume_decompressed.npy"), mmap_mode="r")
  out["decoded_shape"] = list(dec.shape)
  if keep_json:
      print(json.dumps(out, indent=1))
      os.makedirs("gpurun_out", exist_ok=True)
      json.dump(out, open("gpurun_out/c4_at_size.json", "w"), indent=1)
  shutil.rmtree(work, ignore_errors=True)
  return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
