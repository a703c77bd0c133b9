"""NFGR — the compress / decompress task framework of the reference (main.py:164-651) driving
the fused HIP fit/decode path.  Same YAML schema, same on-disk artefact tree (SURVEY.md
Appendix C), same metric names; the per-block process farm of the reference
(utils/TasksManager.py) is replaced by a static longest-first assignment of blocks to the ranks
of a torch.distributed job (one process per GPU) — blocks are independent fits, so the only
collective is the final [SSE, n] all-reduce for PSNR.
"""
import copy
import csv
import logging
import os
import shutil
import time
from os.path import basename as opb
from os.path import join as opj
from os.path import splitext as ops

import numpy as np
import torch

from . import _lib, config
from .fit import Fitter
from .io import get_folder_size, get_type_max, invnormalize_data, minmaxany_range, normalize_data, normalize_data_device, save_yaml, load_yaml
from .metrics import cal_ssim, eval_performance, gpu_eval_u16, gpu_ssim_u16, psnr_from_sse
from .misc import (alloc_param, cal_divide_num, divide_data, merge_divided_data, mip_ops, save_mips, parse_checkpoints,
                   parse_chunk_name, parse_weight, preprocess, preprocess_is_identity, weight_is_unit)
from .modelsave import CopyDir, load_model, save_model
from .networks import (ALL_CALC_PHI_FEATURES, ALL_CALC_PHI_PARAM_COUNT, get_nnmodule_param_count, init_phi)
from .tool import create_stack, read_img, save_img, write_slab


class MyLogger:
    """utils/Logger.py:10-66 without tensorboard: timestamped run directory, script/ copy dir,
    scalar log (metrics.csv with the reference's scalar names)."""

    def __init__(self, outputs_dir="outputs", project_name="single", stdlog=False, tensorboard=False, time=True, logdir=None, **_):
        """logdir: attach to a directory another rank created (a multi-rank job: rank 0 picks the timestamped directory
        and broadcasts it, see main.py) instead of choosing one"""
        import time as _t
        if logdir is None:
            name = project_name + (_t.strftime("_%Y_%m%d_%H%M%S") if time else "")
            logdir = opj(outputs_dir, name)
            if os.path.exists(logdir) and time:
                for i in range(10):
                    if not os.path.exists("%s-%d" % (logdir, i)):
                        logdir = "%s-%d" % (logdir, i)
                        break
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self.script_dir = opj(logdir, "script")
        os.makedirs(self.script_dir, exist_ok=True)
        self._scalars = opj(logdir, "metrics.csv")

    def log_metrics(self, metrics, step):
        new = not os.path.exists(self._scalars)
        with open(self._scalars, "a", newline="") as f:
            w = csv.writer(f)
            if new:
                w.writerow(["step", "name", "value"])
            for k, v in metrics.items():
                w.writerow([step, k, float(v)])

    def close(self):
        pass


from .dist_utils import allreduce_max, allreduce_sum, assign_blocks, broadcast_object, dist_info as _dist  # noqa: E402


class NFGR:
    def __init__(self, opt, Log=None, args=None):
        self.opt = opt
        self.module = {}
        self.Log = Log
        self.args = args
        self.half = bool(opt.Compress.half)
        # optional key of this build (a reference YAML does not have it): fp32 (default, the parity path) | bf16 | bf16x3
        self.precision = str(opt.Compress.get("precision", "fp32"))
        if self.half:
            # Compress.half of the reference (main.py:388-399: fp16 forward/backward, the fp16-rounded weights are what
            # the optimizer updates) maps to the bf16 matrix pipe with fp32 master weights; the budget rule keeps the
            # reference's 2 bytes per parameter (main.py:217, 242).  Pinned band: tests/golden/half.npz — the reference
            # loses 3.7 dB by its half mode on that case, this path 1.3 dB.
            logging.warning("Compress.half: running the hidden GEMMs on the bf16 matrix pipe with fp32 master weights (BRIEF_PREC_BF16)")
            self.precision = "bf16"
        if opt.Compress.loss.name not in ("datal2", "datasmoothl1"):
            raise NotImplementedError(opt.Compress.loss.name)
        if not opt.Compress.gpu or not torch.cuda.is_available():
            raise _lib.BriefError("NFGR on this build runs on a ROCm GPU only (Compress.gpu must be true); there is no CPU fallback")
        self.device = "cuda"

    # ---- budget -> module (main.py:199-264)
    def parse_param_size(self, data_path=None):
        p = self.opt.Compress.param
        if (p.given_size > 0 and p.filesize_ratio > 0) or (p.given_size == 0 and p.filesize_ratio == 0):
            raise ValueError("There can only be one arg to be used")
        return p.given_size if p.given_size > 0 else os.path.getsize(data_path) / p.filesize_ratio

    #: widest net each optional precision has kernels for (include/brief_hip.h); wider nets run exact fp32
    PRECISION_MAX_FEATURES = {"bf16": 512, "bf16x3": 256}

    def init_module(self):
        precision = self.precision
        feats = int(self.opt.Module.phi.get("features", 0) or 0)
        limit = self.PRECISION_MAX_FEATURES.get(precision)
        if limit is not None and feats > limit:
            # decided BEFORE any work starts (in a DivideTask one large block would otherwise abort the job after partitioning):
            # this net runs on the exact fp32 kernels; the artefact records the precision it was fitted in (sideinfos phi_precision)
            logging.warning("Compress.precision=%s supports at most %d features; this net has %d and runs in fp32" % (precision, limit, feats))
            precision = "fp32"
        elif precision == "bf16x3" and 0 < feats < 64:
            logging.warning("Compress.precision=bf16x3 pads every net to 256 features: a %d-wide net runs ~%dx the work of the fp32 path" % (feats, (256 // max(feats, 1)) ** 2 // 4 or 1))
        self.module_precision = precision
        self.module["phi"] = init_phi({**dict(self.opt.Module.phi), "precision": precision})

    @staticmethod
    def estimate_module_size(ideal_module_size, opt):
        name = opt.Module.phi.name
        if name not in ALL_CALC_PHI_FEATURES:
            raise NotImplementedError("Module.phi.name=%r" % name)
        ideal = ideal_module_size / (2.0 if opt.Compress.half else 4.0)
        feats = ALL_CALC_PHI_FEATURES[name](param_count=ideal, **{k: v for k, v in opt.Module.phi.items() if k != "name"})
        kw = {k: v for k, v in opt.Module.phi.items() if k not in ("name", "features")}
        actual = ALL_CALC_PHI_PARAM_COUNT[name](features=feats, **kw)
        return feats, actual, actual * (2.0 if opt.Compress.half else 4.0)

    def prepare_module(self, ideal_module_size):
        feats, actual, theory = NFGR.estimate_module_size(ideal_module_size, self.opt)
        err = (theory - ideal_module_size) / ideal_module_size
        if abs(err) > 0.05:
            logging.warning("Error_rate={:.3f}>0.05! ideal_module_size={} theory_module_size={} ".format(err, ideal_module_size, theory))
        self.opt.Module.phi.features = feats
        self.init_module()
        assert get_nnmodule_param_count(self.module["phi"]) == actual, "calc_phi_param_count mismatch get_nnmodule_param_count !"
        self.module["phi"].to(self.device)
        return feats, theory

    # ---- decode (main.py:266-297)
    @staticmethod
    def decompress(opt, module_path, sideinfos, device="cuda"):
        """opt: loaded options (or a path); sideinfos: dict (or a path).  Returns the numpy volume in
        the source dtype.  The whole grid is evaluated by the forward kernel with the de-normalise +
        truncating cast fused in when Normalize is 'minmaxany_a_b'."""
        if isinstance(opt, str):
            opt = config.load(opt)
        if isinstance(sideinfos, str):
            sideinfos = load_yaml(sideinfos)
        cf = copy.deepcopy(opt.CompressFramework)
        cf.Module.phi.features = sideinfos["phi_features"]
        cf.Module.phi.name = sideinfos["phi_name"]
        # decode in the arithmetic the net was fitted in (side info records it when it is not fp32)
        phi = init_phi({**dict(cf.Module.phi), "precision": str(sideinfos.get("phi_precision", cf.Compress.get("precision", "fp32")))})
        load_model(phi, module_path, "cpu")
        phi.to(device)
        shape = list(sideinfos["data_shape"])
        dims = shape[:-1]
        lo, hi = _coords_range(cf.Compress.coords_mode)
        rng = minmaxany_range(cf.Normalize.name)
        if rng is not None and sideinfos["dtype"] in ("uint8", "uint16"):
            kind = "u8" if sideinfos["dtype"] == "uint8" else "u16"
            out = phi.decode_grid(dims, lo, hi, out_kind=kind, scale=rng, vrange=(sideinfos["min"], sideinfos["max"]))
            data = out.cpu().numpy().reshape(shape)
        else:
            yhat = phi.decode_grid(dims, lo, hi).cpu().reshape(shape)
            data = invnormalize_data(yhat, sideinfos, cf.Normalize.name)
        pp = cf.Decompress.postprocess
        return preprocess(data, pp.denoise.level, pp.denoise.close, pp.clip)

    def sample_nf(self, coords):
        """main.py:266-268: the fitted network evaluated at `coords` without autograd"""
        with torch.no_grad():
            return self.module["phi"].forward(coords)

    def decompress_divide(self, orig_sideinfos_path, module_save_dir, sideinfos_save_dir, opt=None):
        """main.py:299-320: decode every block of a stored DivideTask artefact (steps{k}/compressed/{module,sideinfos}/
        <block>/...) and paste the blocks back by the inclusive index ranges in their names.  `opt` defaults to this
        object's options (the reference re-reads the YAML given on the command line)."""
        orig = load_yaml(orig_sideinfos_path)
        data_shape = list(orig["data_shape"])
        opt = opt if opt is not None else _wrap(self.opt)
        parts = []
        for chunk_name in sorted(os.listdir(module_save_dir)):
            dec = NFGR.decompress(opt, opj(module_save_dir, chunk_name, "module"), opj(sideinfos_save_dir, chunk_name, "sideinfos.yaml"), self.device)
            parts.append({"data": dec, "name": chunk_name, **parse_chunk_name(chunk_name)})
        return merge_divided_data(parts, data_shape)

    # ---- SingleTask encode (main.py:322-454)
    def prepare_fit(self, data_path, data=None, logdir=None):
        """everything main.py:322-384 sets up before the loop: preprocess, loss weights, normalise, size the
        net, put targets on the device, build the Fitter.  Returns the context the loop and the checkpoints use."""
        opt, C_ = self.opt, self.opt.Compress
        logdir = logdir or self.Log.logdir
        if data is None:
            data = read_img(data_path)
        data = np.asarray(data)
        cube = C_.sampler.cube_len
        cube_vox = cube[0] * cube[1] * cube[2] if data.ndim == 4 else cube[1] * cube[2]
        if C_.sampler.name == "randomcube" and min(data.size, cube_vox) > 80 * 80 * 80:
            logging.warning("Chunk size should not be larger than 80*80*80*1!")
            C_.sampler.name = "randompoint"
        pre = preprocess(data, C_.preprocess.denoise.level, C_.preprocess.denoise.close, C_.preprocess.clip)
        name, ext = ops(opb(data_path))
        save_img(opj(logdir, name + "_preprocessed" + ext), pre)
        # loss weights (utils/misc.py:272-307): a map is only built when the spec can produce something else than ones
        weight = None if weight_is_unit(C_.loss.weight) else parse_weight(pre, C_.loss.weight).astype(np.float32, copy=False)     # ('exp_x_v' on integer data is float64)
        # normalise on the device (bit-identical to utils/io.py:65-80, io.normalize_data_device): the targets live there anyway
        tgt_dev, sideinfos = normalize_data_device(pre, opt.Normalize.name, self.device)
        ideal = self.parse_param_size(data_path)
        feats, theory_size = self.prepare_module(ideal)
        phi = self.module["phi"]
        if C_.param.init_net_path != "none":
            load_model(phi, C_.param.init_net_path, "cpu")
        sideinfos = {**sideinfos, "data_shape": list(pre.shape), "phi_features": feats, "phi_name": opt.Module.phi.name}
        if self.precision != "fp32":
            sideinfos["phi_precision"] = getattr(self, "module_precision", self.precision)      # extra key only off the reference's fp32 path: what THIS net was fitted in
        dims = list(pre.shape[:-1])
        cout = pre.shape[-1]
        tgt = tgt_dev.reshape(-1, cout)
        wts = None
        if weight is not None and not bool(np.all(weight == 1.0)):
            wts = torch.from_numpy(np.ascontiguousarray(weight.reshape(-1, cout))).to(self.device)
        assert C_.loss.weight_thres <= get_type_max(pre), "The weight threshold should be less than the data maximum!"
        thr_t, _ = normalize_data(np.array(C_.loss.weight_thres), **opt.Normalize, max=sideinfos["max"], min=sideinfos["min"])
        thr = float(thr_t)
        sampler, index_stream = "randompoint", None
        if C_.sampler.name == "randomcube":
            cl = [min(cube[i], data.shape[i]) for i in range(len(dims))]
            if all(cl[i] == dims[i] for i in range(len(dims))):
                sampler = "full"            # one window == the whole volume every step
            else:
                # the window draws continue the CPU generator's stream behind the net's init draws, as in the reference
                # (main.py:112 after reproduc + init_phi) — through a PRIVATE generator forked from it, so that the draws of
                # one block do not depend on which other blocks are trained beside it, nor on checkpoint spacing / chunking
                gen = torch.Generator()
                gen.set_state(torch.get_rng_state())
                index_stream = _CubeIndexStream(dims, cl, C_.sampler.cube_count, self.device, generator=gen)
        elif C_.sampler.name != "randompoint":
            raise NotImplementedError(C_.sampler.name)
        # Compress.sampler.rng (an extra key, default "philox"): which generator draws the randompoint indices.  "philox" — in the fused
        # kernel, counter-based, keyed by (seed, step): statistically the reference's sampler, not its stream.  "torch" — the reference's
        # own draws (main.py:154-163: torch.randint(0, pop, (sample_size,)) on the CPU generator, once per step, behind reproduc(seed)
        # and the net's init draws), fed to the fit loop as a device-resident index stream: `python main.py -p ...` then touches the
        # voxels the reference touches, step for step.
        rng_mode = str(C_.sampler.get("rng", "philox")).lower()
        if rng_mode not in ("philox", "torch"):
            raise ValueError("Compress.sampler.rng must be 'philox' or 'torch' (got %r)" % rng_mode)
        if rng_mode == "torch" and sampler == "randompoint" and index_stream is None:
            gen = torch.Generator()
            gen.set_state(torch.get_rng_state())      # a private fork, as for the windowed cube sampler above
            index_stream = _PointIndexStream(int(np.prod(dims)), int(C_.sampler.sample_size), self.device, generator=gen)
        n_step = C_.sampler.sample_size if index_stream is None else index_stream.n
        fit = Fitter(phi, tgt, dims, _coords_range(C_.coords_mode), weights=wts, sampler=sampler, sample_size=n_step,
                     optimizer=C_.optimizer_name_phi, lr=C_.lr_phi, scheduler=config.to_plain(C_.lr_scheduler_phi),
                     loss=C_.loss.name, thr=thr, beta=C_.loss.beta, seed=getattr(opt, "_seed", 42), index_stream=index_stream)
        self.sideinfos = sideinfos
        return {"fit": fit, "phi": phi, "data": data, "data_path": data_path, "logdir": logdir, "name": name, "ext": ext,
                "sideinfos": sideinfos, "theory_size": theory_size, "results": {}}

    def checkpoint(self, ctx, steps, loss, evaluate=True):
        """main.py:405-450 at one checkpoint: weight files + sideinfos, optional decode + metrics."""
        opt, C_, Log = self.opt, self.opt.Compress, self.Log
        data, name, ext, sideinfos = ctx["data"], ctx["name"], ctx["ext"], ctx["sideinfos"]
        sdir = opj(ctx["logdir"], "steps{}".format(steps))
        cdir = opj(sdir, "compressed")
        os.makedirs(cdir, exist_ok=True)
        module_path, side_path = opj(cdir, "module"), opj(cdir, "sideinfos.yaml")
        save_yaml(sideinfos, side_path)
        save_model(ctx["phi"], module_path, self.device)
        orig_bytes = os.path.getsize(ctx["data_path"]) if os.path.exists(ctx["data_path"]) else data.nbytes
        side_bytes = os.path.getsize(side_path)
        if Log is not None:
            Log.log_metrics({"compress_ratio/theory": orig_bytes / (side_bytes + ctx["theory_size"]),
                             "compress_ratio/actual": orig_bytes / (side_bytes + get_folder_size(module_path))}, steps)
        if C_.decompress and evaluate:
            dec = NFGR.decompress(_wrap(opt), module_path, sideinfos, self.device)
            if opt.Decompress.keep_decompressed:
                ddir = opj(sdir, "decompressed")
                os.makedirs(ddir, exist_ok=True)
                save_img(opj(ddir, name + "_decompressed" + ext), dec)
            if opt.Decompress.mip and data.ndim == 4:
                mdir = opj(sdir, "mip")
                os.makedirs(mdir, exist_ok=True)
                for tag, vol in ((name, data), (name + "_decompressed", dec)):      # main.py:433-438: in the data's format and as .png
                    mips = mip_ops(vol, mdir, tag, ext)
                    if ext != ".png":
                        save_mips(mips, mdir, tag, ".png")
            if data.dtype == np.uint16 and data.ndim == 4 and data.shape[-1] == 1 and min(data.shape[1:3]) >= 11:
                perf = {"steps": steps, **gpu_eval_u16(data, dec, opt.Decompress.mse, opt.Decompress.psnr, opt.Decompress.ssim)}
                if Log is not None:
                    Log.log_metrics({k: v for k, v in perf.items() if k != "steps"}, steps)
            else:
                perf = eval_performance(steps, data, dec, Log, opt.Decompress.mse, opt.Decompress.psnr, opt.Decompress.ssim)
            perf["loss"] = float(loss.item())
            _append_csv(opj(ctx["logdir"], "performance.csv"), perf)
            ctx["results"][steps] = perf
        # (which step directories survive is compress()'s business: main.py:452-453 / -stepstore)

    def compress(self, data_path, data=None, logdir=None, evaluate=True):
        """main.py:322-454.  The loop body runs inside libbrief_hip.so: one brief_siren_fit call covers every step up to
        the next loss-log or checkpoint boundary (the reference's per-step loss.item() sync, main.py:401, is dropped)."""
        C_, Log = self.opt.Compress, self.Log
        ctx = self.prepare_fit(data_path, data, logdir)
        fit = ctx["fit"]
        max_steps = C_.max_steps
        checkpoints = parse_checkpoints(C_.checkpoints, max_steps)
        freq = int(C_.loss_log_freq) if Log is not None else 0
        stops = sorted(set(checkpoints) | (set(range(freq, max_steps + 1, freq)) if freq > 0 else set()) | {max_steps})
        t_fit, done = 0.0, 0
        for stop in stops:
            t0 = time.perf_counter()
            loss = fit.run(stop - done)
            done = stop
            if freq > 0 and stop % freq == 0:
                Log.log_metrics({"loss": loss.item()}, stop)      # the only host sync, at log frequency
            if stop in checkpoints or stop == max_steps:
                torch.cuda.synchronize()                          # (the weight files are read back next anyway: fit_seconds is device time)
            t_fit += time.perf_counter() - t0
            if stop in checkpoints:
                self.checkpoint(ctx, stop, loss, evaluate)
                # main.py:452-453: the CLI keeps only the last step directory unless -stepstore was given (args.stepstore is
                # True when the flag is ABSENT); library callers (args is None) keep everything
                if self.args is not None and getattr(self.args, "stepstore", False) and stop < max_steps:
                    shutil.rmtree(opj(ctx["logdir"], "steps{}".format(stop)), ignore_errors=True)
        self.fit_seconds = t_fit
        return ctx["results"]

    # ---- DivideTask (main.py:484-651)
    def divide(self, data, data_path, param_size):
        dt = self.opt.Compress.divide.divide_type
        shape = data.shape
        if "adaptive" in dt:
            Nb = int(dt.split("_")[-1])
            if Nb < 8:
                logging.warning("The number of blocks is less than 8!")
                dt = "adaptotal_-1_-1_-1_{}".format(Nb)
            else:
                from .adaptive_blocking import adaptive_chunk
                return adaptive_chunk(data, param_size, dt)
        if "adaptotal" in dt:
            _, nd, nh, nw, Nb = dt.split("_")
            nd, nh, nw, Nb = int(nd), int(nh), int(nw), int(Nb)
            if len(shape) == 3:
                if nh == -1 or nw == -1:
                    nd, nh, nw = cal_divide_num(1, shape[0], shape[1], Nb, param_size)
            elif nd == -1 or nh == -1 or nw == -1:
                nd, nh, nw = cal_divide_num(shape[0], shape[1], shape[2], Nb, param_size)
            return divide_data(data, "total_{}_{}_{}".format(nd, nh, nw))
        if "every" in dt or "total" in dt:
            return divide_data(data, dt)
        raise NotImplementedError(dt)

    def compress_divide(self, data_path, opt_full, data=None, marks=(), on_mark=None):
        """marks / on_mark: optional measurement hook (bench.py): the fit pauses at each optimizer-step count in `marks` and
        calls on_mark(step) — between two marks every block of this rank advances by exactly that many steps.

        main.py:509-651 as one torch.distributed job (one process per GPU), in three phases:
          _partition_blocks  rank 0 partitions the volume and sizes the budgets, the (small) block list is broadcast and
                             mapped to the ranks (longest-first, computed identically everywhere);
          _fit_blocks        every rank fits the blocks it owns (co-trained on HIP streams) from a memory-mapped view of the
                             volume and writes their artefacts into the shared steps{k}/compressed tree;
          _evaluate_divide   the decode of the merged volume is sharded by z: a rank decodes the slices of its slab block by
                             block from the stored artefacts (pruned / dropped regions stay zero, utils/misc.py:432), takes
                             SSE and the per-slice SSIM sums on the GPU against the ORIGINAL data, and writes its slab of
                             the output file; one all-reduce of [SSE_k, SSIM-sum_k, slices, voxels] (RCCL over xGMI) gives
                             PSNR / SSIM.  No volume-sized object ever travels between ranks."""
        dist, rank, world = _dist()
        Log, C_ = self.Log, self.opt.Compress
        logdir = Log.logdir
        if data is None:
            data = read_img(data_path, mmap=True)
        assert data.ndim == self.opt.Module.phi.coords_channel + 1, "The data dimension {} is inconsistent with the neural network input {}!".format(data.ndim - 1, self.opt.Module.phi.coords_channel)
        assert data.shape[-1] == self.opt.Module.phi.data_channel, "The number of data channels {} is inconsistent with the output of neural network {}!".format(data.shape[-1], self.opt.Module.phi.data_channel)
        name, ext = ops(opb(data_path))
        checkpoints = parse_checkpoints(C_.checkpoints, C_.max_steps)
        chunks, owner, src, orig_sideinfos = self._partition_blocks(data, data_path, name, ext)
        self.block_order = [c["name"] for i, c in enumerate(chunks) if owner[i] == rank]      # the order this rank's nets were initialised in
        self._fit_blocks(chunks, owner, src, ext, checkpoints, marks, on_mark)
        if rank == 0:
            for k in checkpoints:
                save_yaml(orig_sideinfos, opj(logdir, "steps{}".format(k), "compressed", "sideinfos.yaml"))
        _barrier(dist)
        results = {}
        if C_.decompress:
            results = self._evaluate_divide(data, data_path, chunks, checkpoints, name, ext)
        _barrier(dist)
        if rank == 0 and not (self.args is not None and getattr(self.args, "substore", False)):
            shutil.rmtree(opj(logdir, "subexps"), ignore_errors=True)
        return results

    def _partition_blocks(self, data, data_path, name, ext):
        """partition + budget on rank 0 only (octree FFT features, variances: main.py:520-546), then names and byte budgets are
        broadcast; returns (blocks, owner rank of every block, the array blocks are cut from, side info of the whole job)"""
        dist, rank, world = _dist()
        C_, logdir = self.opt.Compress, self.Log.logdir
        orig_sideinfos = {"data_shape": list(data.shape)}
        pp = C_.preprocess
        identity = preprocess_is_identity(data, pp.denoise.level, pp.denoise.close, pp.clip)
        param_size = self.parse_param_size(data_path)
        pre = None
        if rank == 0 or not identity:
            # (a non-trivial denoise / clip is applied to the whole volume, as the reference does before dividing it)
            pre = data if identity else preprocess(np.asarray(data), pp.denoise.level, pp.denoise.close, pp.clip)
        desc = None
        if rank == 0:
            save_img(opj(logdir, name + "_preprocessed" + ext), pre)
            chunks, outline = self.divide(pre, data_path, param_size)
            save_img(opj(logdir, "divide" + ext), outline)
            del outline
            n_all = len(chunks)
            chunks = alloc_param(chunks, param_size, C_.divide.param_alloc, C_.divide.param_size_thres)
            desc = {"n_all": n_all, "chunks": [{k: c[k] for k in ("name", "param_size", "size", "total_size")} for c in chunks]}
        desc = broadcast_object(desc)
        orig_sideinfos["chunks_numbers"] = desc["n_all"]
        chunks = desc["chunks"]
        # cost model for the assignment: steps x samples/step x parameters of the block's net
        costs = []
        for c in chunks:
            task = copy.deepcopy(self.opt)
            task.Compress.param.filesize_ratio, task.Compress.param.given_size = 0, c["param_size"]
            f, pcount, c["theory_module_size"] = NFGR.estimate_module_size(c["param_size"], task)
            ns = min(c["size"], C_.sampler.sample_size) if c["size"] > 80 ** 3 else c["size"]
            costs.append(float(C_.max_steps) * ns * pcount)
        return chunks, assign_blocks(costs, world), (data if identity else pre), orig_sideinfos

    def _fit_blocks(self, chunks, owner, src, ext, checkpoints, marks=(), on_mark=None):
        """main.py:547-607 for the blocks this rank owns.  Every block is prepared first (each from the job's seed, as the
        reference's per-block processes do), then all of them are trained TOGETHER: brief_multi_fit spreads them over HIP streams
        so the launches of narrow nets overlap; the results per block are those of a fit on its own.  Blocks with option
        overrides (Compress.divide.exception) keep their own schedule and sampler: they are fitted one by one afterwards."""
        dist, rank, world = _dist()
        C_, logdir = self.opt.Compress, self.Log.logdir
        exceptions = _exceptions(self.opt)
        mine = []
        for i, c in enumerate(chunks):
            if owner[i] != rank:
                continue
            over = exceptions.get(c["name"])
            sub = NFGR(_block_opt(self.opt, c["param_size"], over), Log=None, args=self.args)
            if over and not set(checkpoints) <= set(parse_checkpoints(sub.opt.Compress.checkpoints, sub.opt.Compress.max_steps)):
                raise ValueError("Compress.divide.exception[%s] changes max_steps / checkpoints: the block would not produce the checkpoints %s the job stores" % (c["name"], checkpoints))
            sub_dir = opj(logdir, "subexps", c["name"])
            os.makedirs(sub_dir, exist_ok=True)
            block = np.ascontiguousarray(_orig_block(src, c))
            # the reference fits every block in a process of its own that starts with reproduc(seed) (main.py:573, 653-661, 670):
            # reseeding per block gives each block that generator state for its init and sampler draws, whichever rank owns it
            # and whatever is trained beside it
            seed = getattr(self.opt, "_seed", None)
            if seed is not None:
                torch.manual_seed(int(seed))
            mine.append((c, sub, sub_dir, sub.prepare_fit(opj(sub_dir, c["name"] + ext), data=block, logdir=sub_dir)))
            del block
        special = [m for m in mine if m[0]["name"] in exceptions]
        together = [m for m in mine if m[0]["name"] not in exceptions]
        cotrain = len(together) > 1 and os.environ.get("BRIEF_COTRAIN", "1") != "0"     # every schedule and sampler runs inside brief_multi_fit
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = [int(m) for m in marks if 0 < int(m) <= C_.max_steps]
        if cotrain or on_mark is not None:
            self._fit_step_synchronous(together, checkpoints, marks, on_mark, cotrain)
        else:
            special, together = together + special, []
        for c, sub, sub_dir, ctx in special:        # one block after the other, each through its own checkpoints
            done = 0
            for k in checkpoints:
                loss = ctx["fit"].run(k - done)
                done = k
                sub.checkpoint(ctx, k, loss, evaluate=False)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.fit_seconds = time.perf_counter() - t0
        # ---- artefact tree (main.py:585-607)
        for c, sub, sub_dir, ctx in together + special:
            for k in checkpoints:
                srcd = opj(sub_dir, "steps{}".format(k), "compressed")
                mdst = opj(logdir, "steps{}".format(k), "compressed", "module", c["name"])
                sdst = opj(logdir, "steps{}".format(k), "compressed", "sideinfos", c["name"])
                os.makedirs(mdst, exist_ok=True)
                os.makedirs(sdst, exist_ok=True)
                CopyDir(opj(srcd, "module"), opj(mdst, "module"))
                shutil.copy(opj(srcd, "sideinfos.yaml"), opj(sdst, "sideinfos.yaml"))
            ctx["fit"] = None
            ctx["phi"] = None

    @staticmethod
    def _fit_step_synchronous(blocks, checkpoints, marks, on_mark, cotrain):
        """all blocks of this rank advance together from stop to stop (checkpoints and measurement marks); co-trained on HIP
        streams by brief_multi_fit when there are several.  A mark is reported BEFORE the checkpoint that may fall on the same
        step, so that a timed window (bench.py) never contains artefact writes."""
        from .fit import MultiFitter
        fits = [m[3]["fit"] for m in blocks]
        group = MultiFitter(fits) if cotrain else None
        done = 0
        for k in sorted(set(checkpoints) | set(marks)):
            losses = group.run(k - done) if group is not None else [f.run(k - done) for f in fits]
            done = k
            if on_mark is not None and k in marks:
                on_mark(k)
            if k in checkpoints:
                for (c, sub, sub_dir, ctx), loss in zip(blocks, losses):
                    sub.checkpoint(ctx, k, loss, evaluate=False)

    def _decode_slab(self, step_dir, chunks, z0, z1, shape, dtype):
        """slices [z0, z1) of the merged volume decoded from the stored artefacts of `step_dir`: every block that meets the
        slab is evaluated on its own grid over exactly the z-range of the intersection (a contiguous range of its
        flattened voxels) and pasted by its inclusive index ranges (main.py:299-320, utils/misc.py:430-445); returns a
        device tensor in the source dtype.  2-D data: the whole image (z0 = 0, z1 = 1)."""
        three_d = len(shape) == 4
        tdt = {"uint8": torch.uint8, "uint16": torch.uint16}.get(np.dtype(dtype).name)
        pp = self.opt.Decompress.postprocess
        # the fused epilogue writes the source dtype directly; a Decompress.postprocess that changes values (denoise / a
        # narrowing clip, main.py:294-295) is applied per block by NFGR.decompress, so such jobs take the host branch
        post_identity = preprocess_is_identity(np.zeros(1, dtype), pp.denoise.level, pp.denoise.close, pp.clip)
        fused = tdt is not None and minmaxany_range(self.opt.Normalize.name) is not None and post_identity
        out_shape = ([z1 - z0] + list(shape[1:])) if three_d else list(shape)
        slab = torch.zeros(out_shape, dtype=tdt, device=self.device) if fused else np.zeros(out_shape, dtype=np.float32)
        lo, hi = _coords_range(self.opt.Compress.coords_mode)
        for c in chunks:
            r = parse_chunk_name(c["name"])
            bz0, bz1 = (r["d"][0], r["d"][1] + 1) if three_d else (0, 1)
            za, zb = max(bz0, z0), min(bz1, z1)
            if za >= zb:
                continue
            mod = opj(step_dir, "compressed", "module", c["name"], "module")
            side = load_yaml(opj(step_dir, "compressed", "sideinfos", c["name"], "sideinfos.yaml"))
            y0, y1, x0, x1 = r["h"][0], r["h"][1] + 1, r["w"][0], r["w"][1] + 1
            if fused:
                cf = copy.deepcopy(self.opt)
                cf.Module.phi.features = side["phi_features"]
                cf.Module.phi.name = side.get("phi_name", cf.Module.phi.name)
                phi = init_phi({**dict(cf.Module.phi), "precision": str(side.get("phi_precision", self.precision))})
                load_model(phi, mod, "cpu")
                phi.to(self.device)
                dims = list(side["data_shape"])[:-1]
                plane = int(np.prod(dims[1:])) if three_d else int(np.prod(dims))
                off, cnt = ((za - bz0) * plane, (zb - za) * plane) if three_d else (0, plane)
                part = phi.decode_grid(dims, lo, hi, offset=off, count=cnt, out_kind="u8" if tdt == torch.uint8 else "u16",
                                       scale=minmaxany_range(self.opt.Normalize.name), vrange=(side["min"], side["max"]))
                if three_d:
                    slab[za - z0:zb - z0, y0:y1, x0:x1] = part.view(zb - za, y1 - y0, x1 - x0, -1)
                else:
                    slab[y0:y1, x0:x1] = part.view(y1 - y0, x1 - x0, -1)
            else:       # other normalisations / dtypes: the whole block through NFGR.decompress on the host
                dec = NFGR.decompress(_wrap(self.opt), mod, side, self.device)
                if three_d:
                    slab[za - z0:zb - z0, y0:y1, x0:x1] += dec[za - bz0:zb - bz0]
                else:
                    slab[y0:y1, x0:x1] += dec
        if fused:
            return slab
        return torch.from_numpy(slab.clip(None, get_type_max(np.zeros(1, dtype))).astype(dtype))

    def _evaluate_divide(self, data, data_path, chunks, checkpoints, name, ext):
        """decode + metrics of the merged volume, sharded by z over the ranks (the reference decodes every block serially in
        one process, main.py:613-642)"""
        dist, rank, world = _dist()
        Log, logdir = self.Log, self.Log.logdir
        shape = list(data.shape)
        three_d = len(shape) == 4
        nz = shape[0] if three_d else 1
        z0, z1 = (nz * rank // world, nz * (rank + 1) // world) if three_d else ((0, 1) if rank == 0 else (0, 0))
        drange = get_type_max(data)
        gpu_metrics = data.dtype == np.uint16 and three_d and shape[-1] == 1 and min(shape[1:3]) >= 11
        K = len(checkpoints)
        acc = np.zeros(2 * K + 2, np.float64)        # [SSE_k..., SSIM-sum_k..., slices, elements]
        keep = bool(self.opt.Decompress.keep_decompressed)
        out_paths = {}
        for k in checkpoints:
            sdir = opj(logdir, "steps{}".format(k))
            out_paths[k] = opj(sdir, "decompressed", name + "_decompressed" + ext)
            if keep and rank == 0:
                os.makedirs(opj(sdir, "decompressed"), exist_ok=True)
                if three_d:
                    create_stack(out_paths[k], shape, data.dtype)
        _barrier(dist)
        want_mip = bool(self.opt.Decompress.mip) and three_d
        mips = {}                                       # tag -> this rank's share of the three projections (main.py:622-631)

        def slab_mips(vol):
            """max over z of the slab; the h- and w-projections hold this rank's z rows, zeros elsewhere"""
            md = vol.max(0)
            mh, mw = np.zeros((nz,) + vol.shape[2:], vol.dtype), np.zeros((nz, vol.shape[1]) + vol.shape[3:], vol.dtype)
            mh[z0:z1], mw[z0:z1] = vol.max(1), vol.max(2)
            return [md, mh, mw]
        if z1 > z0:
            orig = np.array(data[z0:z1] if three_d else data, copy=True)      # this rank's slab of the ORIGINAL data
            orig_t = torch.from_numpy(orig).to(self.device) if gpu_metrics else None
            if want_mip:
                mips[name] = slab_mips(orig)
            for ki, k in enumerate(checkpoints):
                sdir = opj(logdir, "steps{}".format(k))
                dec_t = self._decode_slab(sdir, chunks, z0, z1, shape, data.dtype)
                if gpu_metrics and not dec_t.is_cuda:
                    dec_t = dec_t.to(self.device)      # host branch of _decode_slab (other normalisations / postprocess): the metric kernels read device memory
                if want_mip:
                    mips[(k, name + "_decompressed")] = slab_mips(dec_t.cpu().numpy())
                if gpu_metrics:
                    sse = torch.zeros(1, dtype=torch.float64, device=self.device)
                    _lib.check(_lib.lib().brief_sse_u16(_lib.ptr(orig_t), _lib.ptr(dec_t), orig_t.numel(), _lib.ptr(sse), _lib.stream_ptr()))
                    acc[ki] = sse.item()
                    if self.opt.Decompress.ssim:
                        acc[K + ki], _ = gpu_ssim_u16(orig_t, dec_t)
                    dec = dec_t.cpu().numpy() if keep else None
                else:
                    dec = dec_t.cpu().numpy()
                    d64 = dec.astype(np.int64) - orig.astype(np.int64) if np.issubdtype(data.dtype, np.integer) else dec.astype(np.float64) - orig
                    acc[ki] = float((d64 * d64).sum())
                    if self.opt.Decompress.ssim:
                        if three_d:
                            acc[K + ki] = sum(cal_ssim(orig[i].astype(np.float32), dec[i].astype(np.float32), drange) for i in range(z1 - z0))
                        else:
                            acc[K + ki] = cal_ssim(orig.astype(np.float32), dec.astype(np.float32), drange)
                if keep:
                    if three_d:
                        write_slab(out_paths[k], z0, dec)
                    else:
                        save_img(out_paths[k], dec)
            acc[2 * K] = float(z1 - z0)
            acc[2 * K + 1] = float(orig.size)
        tot = allreduce_sum(acc, self.device)         # RCCL over xGMI: the one data collective of this path
        if want_mip:
            # projections of the z-slabs: elementwise MAX over the ranks (the data is non-negative; a rank without a row holds zeros)
            zero = [np.zeros(tuple(shape[1:]), data.dtype), np.zeros((nz,) + tuple(shape[2:]), data.dtype), np.zeros((nz, shape[1]) + tuple(shape[3:]), data.dtype)]
            for tag in [name] + [(k, name + "_decompressed") for k in checkpoints]:
                full = [allreduce_max(m, self.device) for m in mips.get(tag, zero)]
                if rank == 0:
                    for k in (checkpoints if tag == name else [tag[0]]):
                        mdir = opj(logdir, "steps{}".format(k), "mip")
                        os.makedirs(mdir, exist_ok=True)
                        label = tag if tag == name else tag[1]
                        save_mips(full, mdir, label, ext)
                        if ext != ".png":
                            save_mips(full, mdir, label, ".png")
        results = {}
        if rank == 0:
            for ki, k in enumerate(checkpoints):
                sdir = opj(logdir, "steps{}".format(k))
                perf = {"steps": k, "psnr": psnr_from_sse(tot[ki], tot[2 * K + 1], drange)}
                if self.opt.Decompress.mse:
                    perf["mse"] = tot[ki] / tot[2 * K + 1]
                if self.opt.Decompress.ssim:
                    perf["ssim"] = tot[K + ki] / tot[2 * K]
                orig_bytes = os.path.getsize(data_path) if os.path.exists(data_path) else data.nbytes
                cdir = opj(sdir, "compressed")
                theory = get_folder_size(opj(cdir, "sideinfos")) + sum(c["theory_module_size"] for c in chunks)
                Log.log_metrics({"compress_ratio/theory": orig_bytes / theory,
                                 "compress_ratio/actual": orig_bytes / get_folder_size(cdir)}, k)
                Log.log_metrics({m: v for m, v in perf.items() if m != "steps"}, k)
                _append_csv(opj(logdir, "performance.csv"), perf)
                results[k] = perf
        return results


# ------------------------------------------------------------------------------------------ helpers
def _coords_range(mode):
    if mode == "n11":
        return -1.0, 1.0
    if mode == "0p1":
        return 0.0, 1.0
    lo, hi = mode.split(",")
    return float(lo), float(hi)


class _Wrapped(config.Opt):
    pass


def _wrap(cf):
    """NFGR.decompress takes the full option tree; NFGR itself holds only CompressFramework"""
    return _Wrapped({"CompressFramework": cf})


def _block_opt(cf, param_size, override=None):
    """the options of one block's fit (main.py:549-569): the job's options with the block's byte budget, no second
    preprocessing, no evaluation; `override` = Compress.divide.exception[<block name>], a partial tree of the WHOLE option
    file (only its CompressFramework part can matter here), merged last as the reference does"""
    o = copy.deepcopy(cf)
    o.Compress.divide.divide_type = "none"
    o.Compress.param.filesize_ratio = 0
    o.Compress.param.given_size = param_size
    o.Compress.preprocess.denoise.level = 0
    o.Compress.preprocess.denoise.close = False
    o.Compress.decompress = False
    if override:
        part = config.to_plain(override).get("CompressFramework", {})
        if part:
            seed = o.get("_seed")
            o = config.merge(o, part)
            if seed is not None:
                o["_seed"] = seed
    return o


def _exceptions(cf):
    """Compress.divide.exception: 'none' or {block name: partial option tree} (main.py:535-537)"""
    e = cf.Compress.divide.get("exception", "none")
    return {} if e in (None, "none") else dict(e)


def _barrier(dist):
    if dist is not None:
        dist.barrier()


def _orig_block(data, c):
    r = parse_chunk_name(c["name"])
    if "d" in r:
        return data[r["d"][0]:r["d"][1] + 1, r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1]
    return data[r["h"][0]:r["h"][1] + 1, r["w"][0]:r["w"][1] + 1]


def _append_csv(path, row):
    new = not os.path.exists(path)
    with open(path, "a", newline="") as f:
        w = csv.writer(f, dialect="excel")
        if new:
            w.writerow(row.keys())
        w.writerow([row[k] for k in row.keys()])


class _PointIndexStream:
    """RandompointSampler's own index draws (main.py:154-163): `torch.randint(0, pop_size, (sample_size,))` on the CPU generator, one call
    per optimizer step (NFGR.prepare_fit hands over a private fork of the global generator taken behind the net's init draws, which is where
    the reference's loop finds it).  Selected by Compress.sampler.rng: torch; the default sampler draws Philox indices inside the fused kernel.
    The draws of a run of steps travel to the device as ONE [steps, n] tensor (brief_fit_job.idx_stride): brief_siren_fit only enqueues, so
    the host draws the next run while the device works through the current one (`steps_per_call` bounds the run: 64 steps of 100 000
    indices are 30 ms of mt19937 draws and 51 MB).  Pinned: tests/golden/trace.npz (pt_idx)."""

    steps_per_call = 64

    def __init__(self, pop, n, device, generator=None):
        self.pop, self.n = int(pop), int(n)
        self.gen = generator                 # None: the global generator, as the reference
        self.device = device

    def __call__(self, t):
        return torch.randint(0, self.pop, (self.n,), generator=self.gen).to(self.device)

    def batch(self, t_first, steps):
        host = torch.empty((int(steps), self.n), dtype=torch.int64)
        for k in range(int(steps)):          # one randint call per step: the reference's consumption of the generator, draw for draw
            host[k] = torch.randint(0, self.pop, (self.n,), generator=self.gen)
        return host.to(self.device)


class _CubeIndexStream:
    """RandomCubeSampler with windows smaller than the volume (main.py:38-125): `unfold` enumerates the
    prod(dims - cube_len + 1) window origins in row-major order (d slowest); every step draws cube_count of them with
    torch.randint on the CPU generator (main.py:112, seeded by reproduc; NFGR.prepare_fit hands over a private fork of it) and yields the flat voxel indices of
    those windows, window after window, (ds, hs, ws) row-major inside a window.  Pinned: tests/golden/cube.npz."""

    def __init__(self, dims, cube_len, cube_count, device, generator=None):
        self.dims, self.cl, self.count = list(dims), list(cube_len), int(cube_count)
        self.n = self.count * int(np.prod(self.cl))
        self.gen = generator                 # None: the global generator, as the reference
        self.device = device
        grids = torch.meshgrid(*[torch.arange(c) for c in self.cl], indexing="ij")
        strides = [int(np.prod(self.dims[a + 1:])) for a in range(len(self.dims))]
        self.local = sum(g.reshape(-1) * s for g, s in zip(grids, strides))
        self.strides = strides
        self.pops = [self.dims[a] - self.cl[a] + 1 for a in range(len(self.dims))]
        self.pop_size = int(np.prod(self.pops))

    def _origins(self, win):
        """flat voxel index of the first voxel of window number `win` (int64 tensor of any shape)"""
        org, rem = torch.zeros_like(win), win.clone()
        for a in reversed(range(len(self.pops))):
            org += (rem % self.pops[a]) * self.strides[a]
            rem //= self.pops[a]
        return org

    def __call__(self, t):
        win = torch.randint(0, self.pop_size, (self.count,), generator=self.gen)
        return (self._origins(win)[:, None] + self.local[None, :]).reshape(-1).to(self.device)

    def batch(self, t_first, steps):
        """the index sets of `steps` consecutive optimizer steps as one device tensor [steps, n]: the window numbers are
        drawn step by step on the host generator (the reference's draw order), the expansion into voxel indices happens once
        on the device — what lets brief_siren_fit run the windowed sampler without a host round trip per step"""
        win = torch.stack([torch.randint(0, self.pop_size, (self.count,), generator=self.gen) for _ in range(int(steps))])
        org = self._origins(win).to(self.device)
        if not hasattr(self, "_local_dev"):
            self._local_dev = self.local.to(self.device)
        return (org[:, :, None] + self._local_dev[None, None, :]).reshape(int(steps), -1).contiguous()
