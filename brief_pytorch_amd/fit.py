"""The fit loop body of the reference (main.py:385-400) on the fused HIP path.

One Fitter owns: the SIREN module, the device-resident normalised targets (what
RandompointSampler keeps in self.data, main.py:126-163), the optimizer state and the lr
schedule.  step() = sample -> train_step (forward+loss+backward) -> optimizer -> repack,
all enqueued on torch's current stream without any host synchronisation (the reference's
per-step loss.item() sync, main.py:401, is dropped: fetch `loss` when you log).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def multistep_lr(base_lr, milestones, gamma):
    """lr at optimizer step t (1-based): torch MultiStepLR stepped after every optimizer step
    (utils/misc.py:184-197, main.py:400) multiplies the running lr when last_epoch hits a milestone."""
    counts = {}
    for m in milestones:
        counts[int(m)] = counts.get(int(m), 0) + 1
    state = {"lr": float(base_lr), "epoch": 0}

    def lr_at(t):
        while state["epoch"] < t - 1:
            state["epoch"] += 1
            if state["epoch"] in counts:
                state["lr"] = state["lr"] * gamma ** counts[state["epoch"]]
        return state["lr"]
    return lr_at


class Fitter:
    def __init__(self, module, targets, dims, coords_range=(-1.0, 1.0), weights=None, sampler="randompoint",
                 sample_size=100000, optimizer="Adamax", lr=1e-3, scheduler=None, loss="datal2", thr=0.0, beta=0.01,
                 seed=42, index_stream=None):
        self.m = module
        for what, t in (("targets", targets), ("weights", weights)):
            # the kernels read these through raw pointers as contiguous f32 (a float64 weight map, e.g. parse_weight's
            # 'exp_x_v' on integer data, would be silently reinterpreted)
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != module.params.device):
                raise _lib.BriefError("%s must be a contiguous float32 tensor on %s (got %s on %s)" % (what, module.params.device, t.dtype, t.device))
        self.targets = targets              # [pop, cout] f32 on device (normalised)
        self.weights = weights              # [pop, cout] f32 or None (all ones elided)
        self.dims = tuple(int(v) for v in dims)
        self.pop = int(np.prod(self.dims))
        self.range = coords_range
        self.sampler = sampler
        self.n = self.pop if sampler == "full" else int(sample_size)
        self.loss_name, self.thr, self.beta = loss, float(thr), float(beta)
        self.opt = _lib.OPT_KIND[optimizer]
        self.seed = int(seed)
        self.index_stream = index_stream    # optional callable t -> int64 device tensor (parity replays)
        dev = module.params.device
        self.s1 = torch.zeros_like(module.params)
        self.s2 = torch.zeros_like(module.params)
        sch = scheduler or {"name": "none"}
        if sch.get("name") == "MultiStepLR":
            self.lr_at = multistep_lr(lr, sch.get("milestones", []), sch.get("gamma", 0.1))
        elif sch.get("name") in ("none", None):
            self.lr_at = lambda t: float(lr)
        elif sch.get("name") == "StepLR":
            ss, gm = int(sch["step_size"]), float(sch.get("gamma", 0.1))
            self.lr_at = lambda t: float(lr) * gm ** ((t - 1) // ss)
        else:
            raise NotImplementedError("lr scheduler %r" % sch.get("name"))
        self.t = 0
        self._sched_name = sch.get("name") or "none"
        self._milestones = sorted(int(v) for v in sch.get("milestones", [])) if sch.get("name") == "MultiStepLR" else []
        self._gamma = float(sch.get("gamma", 0.1)) if self._milestones else 1.0
        self._ms_arr = (C.c_int64 * max(len(self._milestones), 1))(*self._milestones)
        self._keep = None

    def job(self, steps, log=False):
        """brief_fit_job for the next `steps` optimizer steps (include/brief_hip.h).  Tensors referenced by the
        job are kept alive by this object until the next call."""
        if self.index_stream is not None:
            raise _lib.BriefError("a replayed index stream needs step(): brief_siren_fit draws its indices in-kernel")
        if self._sched_name == "StepLR":
            raise _lib.BriefError("StepLR is applied by step()/run(); brief_fit_job carries MultiStepLR only")
        m = self.m
        m._require_gpu()
        m.sync_packed()
        m.ensure_train_buffers(self.n)
        loss_log = torch.zeros(max(int(steps), 1), dtype=torch.float32, device=m.params.device) if log else None
        g = m._grid(self.dims, self.range[0], self.range[1])
        rnd = self.sampler == "randompoint"
        b = _lib.BatchDesc(None, self.targets.data_ptr(), self.weights.data_ptr() if self.weights is not None else None, None,
                           0, int(self.n), int(self.pop) if rnd else 0, int(self.seed) if rnd else 0, 0)
        j = _lib.FitJob()
        j.desc, j.grid, j.batch = m.desc, g, b
        j.params, j.packed = m.params.data_ptr(), m.packed.data_ptr()
        j.state1, j.state2 = self.s1.data_ptr(), self.s2.data_ptr()
        j.grads, j.loss_out = m.grads.data_ptr(), m._loss.data_ptr()
        j.loss_log = loss_log.data_ptr() if log else None
        j.workspace, j.workspace_bytes = m._ws.data_ptr(), m._ws.numel() * 4
        j.loss_kind, j.optim_kind = _lib.LOSS_KIND[self.loss_name], self.opt
        j.thr, j.beta = self.thr, self.beta
        j.lr, j.beta1, j.beta2, j.eps = self.lr_at(self.t + 1), 0.9, 0.999, 1e-8
        j.milestones = C.cast(self._ms_arr, C.POINTER(C.c_int64))
        j.n_milestones, j.gamma, j.t0 = len(self._milestones), self._gamma, self.t
        self._keep = (loss_log, g, b)
        return j, loss_log

    def run(self, steps, log=False):
        """`steps` optimizer steps in ONE C-ABI call (brief_siren_fit): same results, bit for bit, as calling
        step() that many times.  Returns the device loss of the last step, or the per-step loss tensor if log."""
        if self._sched_name == "StepLR":
            # closed-form schedule (lr * gamma^(epoch // step_size)), not the running product brief_siren_fit applies:
            # keep it exact by stepping from here
            losses = [self.step().clone() for _ in range(int(steps))]
            return torch.cat(losses) if log else self.m._loss
        j, loss_log = self.job(steps, log)
        _lib.check(_lib.lib().brief_siren_fit(C.byref(j), int(steps), _lib.stream_ptr()))
        self.t += int(steps)
        return loss_log if log else self.m._loss

    def step(self):
        """one optimisation step; returns the device loss tensor (no sync)."""
        self.t += 1
        t = self.t
        idx, rng = None, None
        if self.sampler == "randompoint":
            if self.index_stream is not None:
                idx = self.index_stream(t)
            else:
                rng = (self.pop, self.seed, t)      # drawn inside the fused kernel (== brief_sample_indices(pop, seed, t))
        return self.m.fit_step(self.n, self.targets, self.opt, self.s1, self.s2, self.lr_at(t), t, idx=idx, weights=self.weights,
                               grid=(self.dims, self.range[0], self.range[1]), loss=self.loss_name, thr=self.thr, beta=self.beta, rng=rng)


class MultiFitter:
    """Independent fits trained together on one GPU (the blocks of a DivideTask partition that one rank owns,
    main.py:547-575): brief_multi_fit runs fitter j on internal HIP stream j mod 8, so the launches of narrow
    nets overlap instead of leaving most CUs idle.  Every fit's results are identical to running it alone."""

    def __init__(self, fitters):
        self.fitters = list(fitters)

    def run(self, steps, log=False):
        if not self.fitters:
            return []
        jobs, logs = zip(*(f.job(steps, log) for f in self.fitters))
        arr = (_lib.FitJob * len(jobs))(*jobs)
        _lib.check(_lib.lib().brief_multi_fit(arr, len(jobs), int(steps), _lib.stream_ptr()))
        for f in self.fitters:
            f.t += int(steps)
        return list(logs) if log else [f.m._loss for f in self.fitters]
