"""BASELINE config 4 at size through the MULTI-RANK product path, rehearsed on one GPU: N processes (torch.distributed, gloo, all on
cuda:0) share a 1024^3 uint16 volume file; rank 0 partitions (adaptive octree -> eight 512^3 octants) and broadcasts, every rank fits
the octants it owns from a memory map, the evaluation is sharded by z, one all-reduce gives PSNR / SSIM, each rank writes its slab of
the decoded volume.  What an 8-GPU node runs with one octant per rank (RCCL instead of gloo).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29555 tools/c4_multirank.py [steps] [edge]"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np, torch
import torch.distributed as dist
from brief_pytorch_amd import config
from brief_pytorch_amd.dist_utils import broadcast_object
from brief_pytorch_amd.framework import NFGR, MyLogger
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = E // 2
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
work = broadcast_object(tempfile.mkdtemp(prefix="brief_c4mr_") if rank == 0 else None)
path = os.path.join(work, "volume.npy")
t0 = time.perf_counter()
if rank == 0:
    mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint16, shape=(E, E, E, 1))
    for o in range(8):
        z, y, x = (o >> 2) & 1, (o >> 1) & 1, o & 1
        mm[z * H:(z + 1) * H, y * H:(y + 1) * H, x * H:(x + 1) * H] = make_volume_torch((H, H, H), seed=100 + o, device="cuda").cpu().numpy()
    mm.flush(); del mm
dist.barrier()
t_gen = time.perf_counter() - t0
opt = config.load("opt/DivideTask/default.yaml")
cf = opt.CompressFramework
cf.Compress.divide.divide_type = "adaptive_-1_-1_0_0_8"
cf.Compress.divide.param_alloc = "by_size"
cf.Compress.param.filesize_ratio, cf.Compress.param.given_size = 0, 8 * 4.0 * SIREN.calc_param_count(3, 1, 256, 5)
cf.Compress.max_steps, cf.Compress.checkpoints, cf.Compress.loss_log_freq = steps, "none", 10 ** 9
cf.Compress.sampler.name = "randompoint"
cf.Decompress.keep_decompressed, cf.Decompress.mip = True, True
cf["_seed"] = 42
logdir = os.path.join(work, "c4")
Log = MyLogger(outputs_dir=work, project_name="c4", time=False, logdir=logdir)
torch.manual_seed(42)
fw = NFGR(cf, Log=Log)
t0 = time.perf_counter()
res = fw.compress_divide(path, opt)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
import resource
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
allrss = [None] * world
dist.all_gather_object(allrss, (rank, rss, fw.fit_seconds))
if rank == 0:
    names = sorted(os.listdir(os.path.join(logdir, "steps%d" % steps, "compressed", "module")))
    dec = np.load(os.path.join(logdir, "steps%d" % steps, "decompressed", "volume_decompressed.npy"), mmap_mode="r")
    vol = np.load(path, mmap_mode="r")
    # spot check of the merged output against the metrics: PSNR of one z-slab per rank's share, recomputed on the host
    zs = [E * r // world + 3 for r in range(world)]
    d = np.concatenate([(dec[z].astype(np.int64) - vol[z].astype(np.int64)).ravel() for z in zs])
    out = {"ranks": world, "edge": E, "steps": steps, "blocks": names, "perf": {k: float(v) for k, v in res[steps].items()},
           "generate_s": t_gen, "compress_divide_s": t_all, "per_rank (rank, peak host RSS GB, fit seconds)": allrss,
           "decoded_shape": list(dec.shape), "host_psnr_of_%d_sampled_slices" % len(zs): float(-10 * np.log10((d.astype(np.float64) ** 2).mean() / 65535.0 ** 2)),
           "mip_files": sorted(os.listdir(os.path.join(logdir, "steps%d" % steps, "mip")))}
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/c4_multirank.json", "w"), indent=1)
dist.barrier()
if rank == 0:
    shutil.rmtree(work, ignore_errors=True)
dist.destroy_process_group()
