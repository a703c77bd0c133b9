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
