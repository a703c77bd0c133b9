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


def cyclic_lr(base_lr, max_lr, step_size_up=2000, step_size_down=None, mode="triangular", gamma=1.0, scale_mode=None,
              cycle_momentum=True, base_momentum=0.8, max_momentum=0.9, **_):
    """torch.optim.lr_scheduler.CyclicLR as the reference configures it (utils/misc.py:188-189: the YAML keys are passed
    through): returns (lr_at, beta1_at) for optimizer step t (1-based; the scheduler is stepped after every optimizer
    step, main.py:400, so step t runs with last_epoch = t - 1).  With an Adam-family optimizer torch cycles beta1
    inversely to the learning rate when cycle_momentum is left at its default (True)."""
    import math
    up = float(step_size_up)
    down = float(step_size_down) if step_size_down is not None else up
    total = up + down
    ratio = up / total
    if mode not in ("triangular", "triangular2", "exp_range"):
        raise NotImplementedError("CyclicLR mode %r" % mode)
    if scale_mode is None:
        scale_mode = "iterations" if mode == "exp_range" else "cycle"

    def scale_fn(x):
        if mode == "triangular":
            return 1.0
        if mode == "triangular2":
            return 1.0 / (2.0 ** (x - 1))
        return gamma ** x

    def factors(t):
        e = t - 1
        cycle = math.floor(1 + e / total)
        x = 1.0 + e / total - cycle
        sf = x / ratio if x <= ratio else (x - 1) / (ratio - 1)
        return sf, scale_fn(cycle if scale_mode == "cycle" else e)

    def lr_at(t):
        sf, sc = factors(t)
        return base_lr + (max_lr - base_lr) * sf * sc

    def beta1_at(t):
        if not cycle_momentum:
            return None
        sf, sc = factors(t)
        return max_momentum - (max_momentum - base_momentum) * sf * sc
    return lr_at, beta1_at


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
        elif sch.get("name") == "CyclicLR":
            self.lr_at, self.beta1_at = cyclic_lr(**{k: v for k, v in sch.items() if k != "name"})
            if optimizer == "SGD":
                self.beta1_at = lambda t: None      # (torch cycles SGD's momentum; the reference builds SGD without momentum and torch then raises)
        else:
            raise NotImplementedError("lr scheduler %r" % sch.get("name"))
        if not hasattr(self, "beta1_at"):
            self.beta1_at = lambda t: None
        self.t = 0
        self._sched_name = sch.get("name") or "none"
        self._milestones = sorted(int(v) for v in sch.get("milestones", [])) if sch.get("name") == "MultiStepLR" else []
        self._gamma = float(sch.get("gamma", 0.1)) if self._milestones else 1.0
        self._ms_arr = (C.c_int64 * max(len(self._milestones), 1))(*self._milestones)
        self._keep = None

    # indices of a run of steps live on the device as one [steps, n] int64 tensor: at most this many bytes per C-ABI call
    INDEX_STREAM_BYTES = 1 << 28

    def max_steps_per_call(self, share=1):
        """steps one C-ABI call may cover; `share`: how many fitters with index streams split the budget (MultiFitter: every
        co-trained block keeps its run of index sets alive until the call returns, so the budget is divided, not multiplied)"""
        if self.index_stream is None:
            return 1 << 62
        cap = max(1, self.INDEX_STREAM_BYTES // max(int(share), 1) // (8 * self.n))
        return min(cap, int(getattr(self.index_stream, "steps_per_call", cap)))      # (a host-drawn stream asks for short runs: the next one is drawn while this one trains)

    def _index_batch(self, t_first, steps):
        """the index sets of steps t_first .. t_first + steps - 1 as one device tensor [steps, n] (draw order = step order,
        so the host RNG is consumed exactly as by per-step calls)"""
        if hasattr(self.index_stream, "batch"):
            return self.index_stream.batch(t_first, steps)
        return torch.stack([self.index_stream(t_first + k) for k in range(steps)]).contiguous()

    def job(self, steps, log=False):
        """brief_fit_job for the next `steps` optimizer steps (include/brief_hip.h).  Tensors and host arrays referenced by
        the job are kept alive by this object until the next call.  MultiStepLR travels as milestones; the closed-form
        schedules (StepLR, CyclicLR incl. its cycled beta1) as per-step host tables; a replayed / windowed-cube index stream
        as one device-resident [steps, n] tensor."""
        steps = int(steps)
        if steps > self.max_steps_per_call():
            raise _lib.BriefError("index stream: at most %d steps per call (use run())" % self.max_steps_per_call())
        m = self.m
        m._require_gpu()
        m.sync_packed()
        m.ensure_train_buffers(self.n)
        loss_log = torch.zeros(max(steps, 1), dtype=torch.float32, device=m.params.device) if log else None
        g = m._grid(self.dims, self.range[0], self.range[1])
        rnd = self.sampler == "randompoint" and self.index_stream is None
        idx = self._index_batch(self.t + 1, steps) if (self.index_stream is not None and steps > 0) else None
        if idx is not None and (idx.dtype != torch.int64 or tuple(idx.shape) != (steps, self.n) or idx.device != m.params.device):
            raise _lib.BriefError("index stream must yield int64 [steps, %d] on %s" % (self.n, m.params.device))
        b = _lib.BatchDesc(None, self.targets.data_ptr(), self.weights.data_ptr() if self.weights is not None else None,
                           idx.data_ptr() if idx is not None else None,
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
        lr_tab = b1_tab = None
        if self._sched_name in ("StepLR", "CyclicLR") and steps > 0:
            lr_tab = (C.c_double * steps)(*[self.lr_at(self.t + 1 + k) for k in range(steps)])
            j.lr_table = C.cast(lr_tab, C.POINTER(C.c_double))
            if self.beta1_at(self.t + 1) is not None:
                b1_tab = (C.c_double * steps)(*[self.beta1_at(self.t + 1 + k) for k in range(steps)])
                j.beta1_table = C.cast(b1_tab, C.POINTER(C.c_double))
        j.idx_stride = self.n if idx is not None else 0
        self._keep = (loss_log, g, b, idx, lr_tab, b1_tab)
        return j, loss_log

    def run(self, steps, log=False):
        """`steps` optimizer steps in as few C-ABI calls as the index-stream budget allows (one, unless a windowed cube
        sampler or a replayed stream is attached): same results, bit for bit, as calling step() that many times.  Returns
        the device loss of the last step, or the per-step loss tensor if log."""
        steps, logs = int(steps), []
        while True:
            k = min(steps, self.max_steps_per_call())
            j, loss_log = self.job(k, log)
            _lib.check(_lib.lib().brief_siren_fit(C.byref(j), k, _lib.stream_ptr()))
            self.t += k
            steps -= k
            if log:
                logs.append(loss_log[:k])
            if steps <= 0:
                break
        return (torch.cat(logs) if len(logs) > 1 else logs[0]) if log else self.m._loss

    def step(self):
        """one optimisation step; returns the device loss tensor (no sync)."""
        self.t += 1
        t = self.t
        idx, rng = None, None
        if self.index_stream is not None:
            idx = self.index_stream(t)
        elif self.sampler == "randompoint":
            rng = (self.pop, self.seed, t)      # drawn inside the fused kernel (== brief_sample_indices(pop, seed, t))
        b1 = self.beta1_at(t)
        return self.m.fit_step(self.n, self.targets, self.opt, self.s1, self.s2, self.lr_at(t), t, idx=idx, weights=self.weights,
                               grid=(self.dims, self.range[0], self.range[1]), loss=self.loss_name, thr=self.thr, beta=self.beta, rng=rng,
                               betas=(0.9 if b1 is None else b1, 0.999))


class MultiFitter:
    """Independent fits trained together on one GPU (the blocks of a DivideTask partition that one rank owns,
    main.py:547-575): brief_multi_fit runs fitter j on internal HIP stream j mod 8, so the launches of narrow
    nets overlap instead of leaving most CUs idle.  Every fit's results are identical to running it alone."""

    def __init__(self, fitters):
        self.fitters = list(fitters)

    def run(self, steps, log=False):
        if not self.fitters:
            return []
        steps, logs = int(steps), [[] for _ in self.fitters]
        while True:
            share = sum(1 for f in self.fitters if f.index_stream is not None)
            k = min([steps] + [f.max_steps_per_call(share) for f in self.fitters])
            jobs, lg = zip(*(f.job(k, log) for f in self.fitters))
            arr = (_lib.FitJob * len(jobs))(*jobs)
            _lib.check(_lib.lib().brief_multi_fit(arr, len(jobs), k, _lib.stream_ptr()))
            for f, acc, l in zip(self.fitters, logs, lg):
                f.t += k
                if log:
                    acc.append(l[:k])
            steps -= k
            if steps <= 0:
                break
        return [torch.cat(a) if len(a) > 1 else a[0] for a in logs] if log else [f.m._loss for f in self.fitters]
