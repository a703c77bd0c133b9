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
from .io import get_folder_size, get_type_max, invnormalize_data, minmaxany_range, normalize_data, save_yaml, load_yaml
from .metrics import cal_ssim, eval_performance, gpu_eval_u16, gpu_ssim_u16, psnr_from_sse
from .misc import (alloc_param, cal_divide_num, divide_data, merge_divided_data, mip_ops, parse_checkpoints,
                   parse_chunk_name, parse_weight, preprocess)
from .modelsave import CopyDir, load_model, save_model
from .networks import (ALL_CALC_PHI_FEATURES, ALL_CALC_PHI_PARAM_COUNT, get_nnmodule_param_count, init_phi)
from .tool import read_img, save_img


class MyLogger:
    """utils/Logger.py:10-66 without tensorboard: timestamped run directory, script/ copy dir,
    scalar log (metrics.csv with the reference's scalar names)."""

    def __init__(self, outputs_dir="outputs", project_name="single", stdlog=False, tensorboard=False, time=True, **_):
        import time as _t
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


from .dist_utils import allreduce_sse, assign_blocks, dist_info as _dist, gather_objects  # noqa: E402


class NFGR:
    def __init__(self, opt, Log=None, args=None):
        self.opt = opt
        self.module = {}
        self.Log = Log
        self.args = args
        self.half = opt.Compress.half
        if self.half:
            raise NotImplementedError("Compress.half (fp16 autocast) is not available on the fused path; "
                                      "Compress.precision: bf16 is its MI355X counterpart (bf16 MFMA, fp32 master weights)")
        # optional key of this build (a reference YAML does not have it): fp32 (default, the parity path) | bf16
        self.precision = str(opt.Compress.get("precision", "fp32"))
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

    def init_module(self):
        self.module["phi"] = init_phi({**dict(self.opt.Module.phi), "precision": self.precision})

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
        cube = C_.sampler.cube_len
        cube_vox = cube[0] * cube[1] * cube[2] if data.ndim == 4 else cube[1] * cube[2]
        if C_.sampler.name == "randomcube" and min(data.size, cube_vox) > 80 * 80 * 80:
            logging.warning("Chunk size should not be larger than 80*80*80*1!")
            C_.sampler.name = "randompoint"
        pre = preprocess(data, C_.preprocess.denoise.level, C_.preprocess.denoise.close, C_.preprocess.clip)
        name, ext = ops(opb(data_path))
        save_img(opj(logdir, name + "_preprocessed" + ext), pre)
        weight = parse_weight(pre, C_.loss.weight)
        norm, sideinfos = normalize_data(pre, **opt.Normalize)
        ideal = self.parse_param_size(data_path)
        feats, theory_size = self.prepare_module(ideal)
        phi = self.module["phi"]
        if C_.param.init_net_path != "none":
            load_model(phi, C_.param.init_net_path, "cpu")
        sideinfos = {**sideinfos, "data_shape": list(norm.shape), "phi_features": feats, "phi_name": opt.Module.phi.name}
        if self.precision != "fp32":
            sideinfos["phi_precision"] = self.precision      # extra key only off the reference's fp32 path
        dims = list(norm.shape[:-1])
        cout = norm.shape[-1]
        tgt = norm.reshape(-1, cout).to(self.device)
        wts = None if bool(np.all(weight == 1.0)) else torch.from_numpy(np.ascontiguousarray(weight.reshape(-1, cout))).to(self.device)
        assert C_.loss.weight_thres <= get_type_max(pre), "The weight threshold should be less than the data maximum!"
        thr_t, _ = normalize_data(np.array(C_.loss.weight_thres), **opt.Normalize, max=sideinfos["max"], min=sideinfos["min"])
        thr = float(thr_t)
        sampler, index_stream = "randompoint", None
        if C_.sampler.name == "randomcube":
            cl = [min(cube[i], data.shape[i]) for i in range(len(dims))]
            if all(cl[i] == dims[i] for i in range(len(dims))):
                sampler = "full"            # one window == the whole volume every step
            else:
                index_stream = _CubeIndexStream(dims, cl, C_.sampler.cube_count, self.device)
        elif C_.sampler.name != "randompoint":
            raise NotImplementedError(C_.sampler.name)
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
                for tag, vol in ((name, data), (name + "_decompressed", dec)):
                    for ax, img in zip("dhw", mip_ops(vol)):
                        save_img(opj(mdir, "%s_mip_%s%s" % (tag, ax, ext)), img)
            if data.dtype == np.uint16 and data.ndim == 4 and data.shape[-1] == 1 and min(data.shape[1:3]) >= 11:
                perf = {"steps": steps, **gpu_eval_u16(data, dec, opt.Decompress.mse, opt.Decompress.psnr, opt.Decompress.ssim)}
                if Log is not None:
                    Log.log_metrics({k: v for k, v in perf.items() if k != "steps"}, steps)
            else:
                perf = eval_performance(steps, data, dec, Log, opt.Decompress.mse, opt.Decompress.psnr, opt.Decompress.ssim)
            perf["loss"] = float(loss.item())
            _append_csv(opj(ctx["logdir"], "performance.csv"), perf)
            ctx["results"][steps] = perf
        # step directories are always kept (the reference's inverted -stepstore flag is not reproduced)

    def compress(self, data_path, data=None, logdir=None, evaluate=True):
        C_, Log = self.opt.Compress, self.Log
        ctx = self.prepare_fit(data_path, data, logdir)
        fit = ctx["fit"]
        max_steps = C_.max_steps
        checkpoints = parse_checkpoints(C_.checkpoints, max_steps)
        t_fit = 0.0
        for steps in range(1, max_steps + 1):
            t0 = time.perf_counter()
            loss = fit.step()
            if steps % C_.loss_log_freq == 0 and Log is not None:
                Log.log_metrics({"loss": loss.item()}, steps)      # the only host sync, at log frequency
            t_fit += time.perf_counter() - t0
            if steps in checkpoints:
                self.checkpoint(ctx, steps, loss, evaluate)
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

    def compress_divide(self, data_path, opt_full, data=None):
        """partition -> budget -> independent per-block fits spread over the ranks -> per-block decode
        on the owning rank -> SSE all-reduce (PSNR) -> rank 0 merges, saves and evaluates."""
        dist, rank, world = _dist()
        Log, C_ = self.Log, self.opt.Compress
        logdir = Log.logdir
        if data is None:
            data = read_img(data_path)
        assert data.ndim == self.opt.Module.phi.coords_channel + 1, "The data dimension {} is inconsistent with the neural network input {}!".format(data.ndim - 1, self.opt.Module.phi.coords_channel)
        assert data.shape[-1] == self.opt.Module.phi.data_channel, "The number of data channels {} is inconsistent with the output of neural network {}!".format(data.shape[-1], self.opt.Module.phi.data_channel)
        orig_sideinfos = {"data_shape": list(data.shape)}
        pre = preprocess(data, C_.preprocess.denoise.level, C_.preprocess.denoise.close, C_.preprocess.clip)
        name, ext = ops(opb(data_path))
        if rank == 0:
            save_img(opj(logdir, name + "_preprocessed" + ext), pre)
        param_size = self.parse_param_size(data_path)
        chunks, outline = self.divide(pre, data_path, param_size)
        if rank == 0:
            save_img(opj(logdir, "divide" + ext), outline)
        orig_sideinfos["chunks_numbers"] = len(chunks)
        chunks = alloc_param(chunks, param_size, C_.divide.param_alloc, C_.divide.param_size_thres)
        checkpoints = parse_checkpoints(C_.checkpoints, C_.max_steps)
        # cost model for the assignment: steps x samples/step x train FLOPs of the block's net
        costs = []
        for c in chunks:
            task = copy.deepcopy(self.opt)
            task.Compress.param.filesize_ratio, task.Compress.param.given_size = 0, c["param_size"]
            f, pcount, c["theory_module_size"] = NFGR.estimate_module_size(c["param_size"], task)
            ns = min(c["size"], C_.sampler.sample_size) if c["size"] > 80 ** 3 else c["size"]
            costs.append(float(C_.max_steps) * ns * pcount)
        owner = assign_blocks(costs, world)
        sse = np.zeros(len(checkpoints), np.float64)
        cnt = 0.0
        decoded = {k: [] for k in checkpoints}
        # every block this rank owns is prepared first (nets are initialised in partition order, as a serial
        # run would), then all of them are trained TOGETHER: brief_multi_fit spreads them over HIP streams so the
        # launches of narrow nets overlap; the results per block are those of a fit on its own.
        mine = []
        for i, c in enumerate(chunks):
            if owner[i] != rank:
                continue
            sub = NFGR(_block_opt(self.opt, c["param_size"]), Log=None, args=self.args)
            sub_dir = opj(logdir, "subexps", c["name"])
            os.makedirs(sub_dir, exist_ok=True)
            block = np.ascontiguousarray(c["data"])
            mine.append((c, sub, sub_dir, block, sub.prepare_fit(opj(sub_dir, c["name"] + ext), data=block, logdir=sub_dir)))
        cotrain = len(mine) > 1 and all(m[4]["fit"].index_stream is None for m in mine) and os.environ.get("BRIEF_COTRAIN", "1") != "0"
        t0 = time.perf_counter()
        if cotrain:
            from .fit import MultiFitter
            group = MultiFitter([m[4]["fit"] for m in mine])
            done = 0
            for k in checkpoints:
                losses = group.run(k - done)
                done = k
                for (c, sub, sub_dir, block, ctx), loss in zip(mine, losses):
                    sub.checkpoint(ctx, k, loss, evaluate=False)
        else:
            for c, sub, sub_dir, block, ctx in mine:
                for steps in range(1, C_.max_steps + 1):
                    loss = ctx["fit"].step()
                    if steps in checkpoints:
                        sub.checkpoint(ctx, steps, loss, evaluate=False)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        self.fit_seconds = time.perf_counter() - t0
        for c, sub, sub_dir, block, ctx in mine:
            for ki, k in enumerate(checkpoints):
                src = opj(sub_dir, "steps{}".format(k), "compressed")
                mdst = opj(logdir, "steps{}".format(k), "compressed", "module", c["name"])
                sdst = opj(logdir, "steps{}".format(k), "compressed", "sideinfos", c["name"])
                os.makedirs(mdst, exist_ok=True)
                os.makedirs(sdst, exist_ok=True)
                CopyDir(opj(src, "module"), opj(mdst, "module"))
                shutil.copy(opj(src, "sideinfos.yaml"), opj(sdst, "sideinfos.yaml"))
                if C_.decompress:
                    dec = NFGR.decompress(_wrap(sub.opt), opj(mdst, "module"), opj(sdst, "sideinfos.yaml"), self.device)
                    d64 = dec.astype(np.int64) - np.ascontiguousarray(_orig_block(data, c)).astype(np.int64)
                    sse[ki] += float((d64 * d64).sum())
                    decoded[k].append({"data": dec, "name": c["name"], **parse_chunk_name(c["name"])})
            cnt += float(block.size)
        results = {}
        if C_.decompress:
            tot_sse, tot_cnt = allreduce_sse(sse, cnt, self.device)   # RCCL over xGMI: the one collective of this path
            tot = np.concatenate([tot_sse, [tot_cnt]])
            gathered = gather_objects(decoded)                        # decoded blocks to every rank (rank 0 merges)
            parts = {k: [x for g in gathered for x in g[k]] for k in checkpoints}
            if rank == 0:
                drange = get_type_max(data)
                for ki, k in enumerate(checkpoints):
                    sdir = opj(logdir, "steps{}".format(k))
                    save_yaml(orig_sideinfos, opj(sdir, "compressed", "sideinfos.yaml"))
                    merged = merge_divided_data(parts[k], list(data.shape))
                    if self.opt.Decompress.keep_decompressed:
                        os.makedirs(opj(sdir, "decompressed"), exist_ok=True)
                        save_img(opj(sdir, "decompressed", name + "_decompressed" + ext), merged)
                    # PSNR from the all-reduced SSE covers the fitted blocks; dropped/pruned blocks decode as zeros
                    missing = float(data.size) - tot[-1]
                    sse_all = tot[ki]
                    if missing > 0:
                        d64 = merged.astype(np.int64) - data.astype(np.int64)
                        sse_all = float((d64 * d64).sum())
                    perf = {"steps": k, "psnr": psnr_from_sse(sse_all, float(data.size), drange)}
                    if self.opt.Decompress.mse:
                        perf["mse"] = sse_all / float(data.size)
                    if self.opt.Decompress.ssim:
                        if data.dtype == np.uint16 and data.shape[-1] == 1 and min(data.shape[1:3]) >= 11:
                            ss, ns = gpu_ssim_u16(torch.from_numpy(np.ascontiguousarray(data)).cuda(), torch.from_numpy(np.ascontiguousarray(merged)).cuda())
                            perf["ssim"] = ss / ns
                        else:
                            perf["ssim"] = cal_ssim(data.astype(np.float32), merged.astype(np.float32), drange)
                    orig_bytes = os.path.getsize(data_path) if os.path.exists(data_path) else data.nbytes
                    cdir = opj(sdir, "compressed")
                    theory = get_folder_size(opj(cdir, "sideinfos")) + sum(c["theory_module_size"] for c in chunks)
                    Log.log_metrics({"compress_ratio/theory": orig_bytes / theory,
                                     "compress_ratio/actual": orig_bytes / get_folder_size(cdir)}, k)
                    Log.log_metrics({m: v for m, v in perf.items() if m != "steps"}, k)
                    _append_csv(opj(logdir, "performance.csv"), perf)
                    results[k] = perf
        if dist is not None:
            dist.barrier()
        if rank == 0 and not (self.args is not None and getattr(self.args, "substore", False)):
            shutil.rmtree(opj(logdir, "subexps"), ignore_errors=True)
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


def _block_opt(cf, param_size):
    o = copy.deepcopy(cf)
    o.Compress.divide.divide_type = "none"
    o.Compress.param.filesize_ratio = 0
    o.Compress.param.given_size = param_size
    o.Compress.preprocess.denoise.level = 0
    o.Compress.preprocess.denoise.close = False
    o.Compress.decompress = False
    return o


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


class _CubeIndexStream:
    """RandomCubeSampler with windows smaller than the volume (main.py:38-125): `unfold` enumerates the
    prod(dims - cube_len + 1) window origins in row-major order (d slowest); every step draws cube_count of them with
    torch.randint on the GLOBAL CPU generator (main.py:112, seeded by reproduc) and yields the flat voxel indices of
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

    def __call__(self, t):
        win = torch.randint(0, self.pop_size, (self.count,), generator=self.gen)
        idx = []
        for wv in win.tolist():
            org, rem = 0, wv
            for a in reversed(range(len(self.pops))):
                org += (rem % self.pops[a]) * self.strides[a]
                rem //= self.pops[a]
            idx.append(self.local + org)
        return torch.cat(idx).to(self.device)
