#!/usr/bin/env python3
"""Command line of brief_pytorch_amd — a drop-in for the reference's main.py (main.py:664-706):

    python main.py -p opt/SingleTask/default.yaml -g 0
    python -m torch.distributed.run --nproc-per-node 8 main.py -p opt/DivideTask/default.yaml

SingleTask fits one SIREN to the volume; DivideTask partitions it and fits the blocks
independently, one process per GPU when launched under torch.distributed (RCCL only reduces the
final SSE).  Flags -gc/-cc/-t/-m/-dropslice/-debug of the reference's NVIDIA process farm are
accepted and ignored.
"""
import argparse
import os
import random
import shutil
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def reproduc(opt):
    """main.py:653-661"""
    import numpy as np
    import torch
    random.seed(opt.seed)
    np.random.seed(opt.seed)
    torch.manual_seed(opt.seed)
    torch.cuda.manual_seed_all(opt.seed)


def main(argv=None):
    ap = argparse.ArgumentParser(description="single task for datacompress (MI355X fused path)")
    ap.add_argument("-p", type=str, default=os.path.join(ROOT, "opt", "SingleTask", "default.yaml"), help="yaml file path")
    ap.add_argument("-g", help="available gpu list", default=None, type=lambda s: [int(i) for i in s.split(",")])
    for flag, kw in (("-gc", dict(type=int, default=8000)), ("-cc", dict(type=int, default=3000)), ("-t", dict(type=float, default=2)),
                     ("-m", dict(type=int, default=33))):
        ap.add_argument(flag, **kw)
    ap.add_argument("-dropslice", action="store_true")
    ap.add_argument("-debug", action="store_true")
    ap.add_argument("-substore", action="store_true", help="keep the per-block sub-experiment directories")
    # the reference's flag is inverted (main.py:695: store_false, tested at main.py:452): WITHOUT -stepstore a SingleTask run
    # deletes every steps{k} directory except the last one after evaluating it; WITH it they are all kept
    ap.add_argument("-stepstore", action="store_false", help="keep the intermediate steps{k} directories of a SingleTask run")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.g:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", ",".join(str(i) for i in args.g[:1]))
    import torch
    from brief_pytorch_amd import config
    from brief_pytorch_amd.framework import NFGR, MyLogger
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        # "nccl" is RCCL on ROCm; BRIEF_DIST_BACKEND=gloo rehearses the multi-rank path when ranks share a GPU
        dist.init_process_group(os.environ.get("BRIEF_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    opt = config.load(args.p)
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # rank 0 picks the (timestamped) run directory, the others attach to it: a second run never lands in the
        # directory of an earlier one, and directories are created by one process
        from brief_pytorch_amd.dist_utils import broadcast_object
        Log = MyLogger(**dict(opt.Log)) if rank == 0 else None
        logdir = broadcast_object(Log.logdir if rank == 0 else None)
        if rank != 0:
            Log = MyLogger(**{**dict(opt.Log), "logdir": logdir})
    else:
        Log = MyLogger(**dict(opt.Log))
    if rank == 0:
        shutil.copy(args.p, Log.script_dir)
    reproduc(opt.Reproduc)
    cf = opt.CompressFramework
    cf["_seed"] = opt.Reproduc.seed
    from brief_pytorch_amd.synthetic import ensure_dataset
    if int(os.environ.get("RANK", "0")) == 0:
        ensure_dataset(opt.Dataset.data_path)      # dataset/synthetic_<n>.tif is generated on first use
    if world > 1:
        torch.distributed.barrier()
    fw = NFGR(cf, Log=Log, args=args)
    if cf.Compress.divide.divide_type == "none":
        res = fw.compress(opt.Dataset.data_path)
    else:
        res = fw.compress_divide(opt.Dataset.data_path, opt)
    if int(os.environ.get("RANK", "0")) == 0:
        for k, v in (res or {}).items():
            print("steps %d: %s" % (k, {m: float(x) for m, x in v.items() if m != "steps"}))
        print("outputs in", Log.logdir)
    Log.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
