"""SIREN on the fused HIP path — the drop-in for the object init_phi() returns in the reference.

Mirrors the surface NFGR / ModelSave use (reference utils/Networks.py:246-314, 795-802;
call sites SURVEY.md section 8b): constructor kwargs, forward(coords), parameters(),
state_dict(), to()/float()/half(), net[l][0].weight/.bias (readable AND assignable),
calc_param_count / calc_features.  All arithmetic runs in libbrief_hip.so; there is no
torch.autograd MLP and no CPU fallback.
"""
import copy
import ctypes as C
import logging
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

__all__ = ["SIREN", "init_phi", "ALLPHI", "ALL_CALC_PHI_FEATURES", "ALL_CALC_PHI_PARAM_COUNT",
           "ALL_CHECK_PARAM_COUNT", "get_nnmodule_param_count"]


class _ParamView:
    """`net[l][0].weight` / `.bias`: a window into the packed parameter buffer.
    `.data` reads a tensor view; assigning `.data = t` copies into the buffer (what
    utils/ModelSave.py:20,27 does) and marks the fragment-ordered copy stale."""

    def __init__(self, owner, off, shape):
        self._o, self._off, self._shape = owner, off, tuple(shape)

    def _view(self, buf):
        n = int(np.prod(self._shape))
        return buf[self._off:self._off + n].view(self._shape)

    @property
    def data(self):
        return self._view(self._o.params)

    @data.setter
    def data(self, value):
        v = torch.as_tensor(value, dtype=torch.float32).reshape(self._shape)
        self._view(self._o.params).copy_(v.to(self._o.params.device))
        self._o._stale = True

    @property
    def grad(self):
        return None if self._o.grads is None else self._view(self._o.grads)

    @property
    def shape(self):
        return torch.Size(self._shape)

    def size(self, dim=None):
        return self.shape if dim is None else self._shape[dim]

    def numel(self):
        return int(np.prod(self._shape))

    def detach(self):
        return self.data

    def __len__(self):
        return self._shape[0]


class _Linear:
    def __init__(self, owner, woff, wshape, boff):
        self.weight = _ParamView(owner, woff, wshape)
        self.bias = _ParamView(owner, boff, (wshape[0],))
        self.in_features, self.out_features = wshape[1], wshape[0]


class _Seq:
    """stands for nn.Sequential(Linear[, Sine]); index 0 is the Linear."""

    def __init__(self, lin):
        self._lin = lin

    def __getitem__(self, i):
        if i != 0:
            raise IndexError("only the Linear (index 0) carries parameters")
        return self._lin

    def __len__(self):
        return 1


class _SirenFn(torch.autograd.Function):
    """forward/backward of the module under torch autograd (main.py:391-396: `data_hat = phi.forward(x);
    loss = loss_func(...); loss.backward()`).  backward hands dL/dyhat to the fused train step
    (BRIEF_LOSS_EXTERNAL) and leaves the gradient in module.params.grad, where torch.optim looks for it."""

    @staticmethod
    def forward(ctx, anchor, coords, module):
        ctx.module = module
        ctx.save_for_backward(coords)
        return module._forward_plain(coords)

    @staticmethod
    def backward(ctx, gy):
        (coords,) = ctx.saved_tensors
        m = ctx.module
        cin, cout = m.coords_channel, m.data_channel
        n = coords.numel() // cin
        m.train_step(n, gy.reshape(n, cout).to(torch.float32).contiguous(), coords=coords.reshape(n, cin), loss="external")
        if m.params.grad is None:
            m.params.grad = m.grads.clone()
        else:
            m.params.grad += m.grads
        return gy.new_zeros(1), None, None


class SIREN:
    """reference: utils/Networks.py:236-314."""

    def __init__(self, coords_channel=3, data_channel=1, features=256, layers=5, w0=30, res=False,
                 output_act=False, device=None, precision="fp32", **kwargs):
        """precision: 'fp32' (exact f32 MFMA, the parity path, features <= 4096), 'bf16' (hidden GEMMs on the bf16 matrix pipe,
        f32 master weights: the MI355X counterpart of Compress.half; include/brief_hip.h: BRIEF_PREC_BF16, features <= 512) or 'bf16x3'
        (split precision inside the fp32 parity bands, BRIEF_PREC_BF16X3, features <= 256; training AND inference run the split chains)."""
        if res:
            # HalfResidual blocks cannot be saved by the reference's own ModelSave
            raise NotImplementedError("SIREN(res=True) is unsupported on the fused path")
        self.coords_channel, self.data_channel = int(coords_channel), int(data_channel)
        self.features, self.layers = int(features), int(layers)
        self.w0, self.output_act = float(w0), bool(output_act)
        self.precision = str(precision)
        self.desc = _lib.SirenDesc(self.coords_channel, self.data_channel, self.layers, self.features,
                                   self.w0, 30.0, int(self.output_act), _lib.PRECISION[self.precision])
        F = self.features
        self._shapes = [(F, self.coords_channel)] + [(F, F)] * (self.layers - 2) + [(self.data_channel, F)]
        self.param_count = sum(o * i + o for o, i in self._shapes)
        self.params = self._reference_init()          # CPU until .to(device)
        self.grads = None
        self.packed = None
        self._stale = True
        self._seen_version = -1
        self._autograd = False
        self._anchor = None
        self._ws = None
        self._fws = None
        self._loss = None
        net, off = [], 0
        for (o, i) in self._shapes:
            net.append(_Seq(_Linear(self, off, (o, i), off + o * i)))
            off += o * i + o
        self.net = net
        if device is not None:
            self.to(device)

    # ---- initialisation: replays the reference's torch-RNG draws so that equal seeds give equal nets
    def _reference_init(self):
        """nn.Linear default init for every layer in order (weight then bias), then sine_init
        over all weights in layer order, then first_layer_sine_init (utils/Networks.py:215-226,
        246-266).  Values AND generator consumption match torch, so after
        torch.manual_seed(s) this reproduces the reference's tensors bit for bit."""
        ws, bs = [], []
        for (o, i) in self._shapes:
            w = torch.empty(o, i)
            gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
            bound = math.sqrt(3.0) * gain / math.sqrt(i)
            w.uniform_(-bound, bound)
            b = torch.empty(o)
            bb = 1 / math.sqrt(i) if i > 0 else 0
            b.uniform_(-bb, bb)
            ws.append(w)
            bs.append(b)
        for w in ws:
            num_input = w.size(-1)
            w.uniform_(-np.sqrt(6 / num_input) / 30, np.sqrt(6 / num_input) / 30)
        num_input = ws[0].size(-1)
        ws[0].uniform_(-1 / num_input, 1 / num_input)
        return torch.cat([torch.cat([w.reshape(-1), b]) for w, b in zip(ws, bs)]).contiguous()

    # ---- nn.Module-like surface
    def parameters(self):
        return [self.params]

    def state_dict(self):
        sd = OrderedDict()
        for l, seq in enumerate(self.net):
            sd["net.%d.0.weight" % l] = seq[0].weight.data
            sd["net.%d.0.bias" % l] = seq[0].bias.data
        return sd

    def load_state_dict(self, sd):
        for l, seq in enumerate(self.net):
            seq[0].weight.data = sd["net.%d.0.weight" % l]
            seq[0].bias.data = sd["net.%d.0.bias" % l]

    def to(self, device):
        device = torch.device(device)
        if self.params.device != device:
            self.params = self.params.to(device)
            self.grads = None
            self.packed = None
            self._ws = self._fws = None
            self._stale = True
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def _set_precision(self, precision):
        if precision != self.precision:
            self.precision = precision
            self.desc.precision = _lib.PRECISION[precision]
            self.packed = None          # the fragment-ordered copy has another size and content
            self._ws = self._fws = None
            self._stale = True
        return self

    def float(self):
        """nn.Module.float() as the reference's low-precision loop uses it (main.py:398: back to fp32 for the update)"""
        # (the mode half() recorded is consumed here: after a _set_precision / to() round trip a later float() cannot restore a stale one)
        return self._set_precision(self.__dict__.pop("_float_precision", self.precision))

    def half(self):
        """nn.Module.half() as the reference uses it (main.py:212, 287-288, 389): its fp16 mode.  The MI355X counterpart is the
        bf16 matrix pipe with fp32 master weights (BRIEF_PREC_BF16, what Compress.half selects in NFGR): the parameters stay
        fp32, the hidden GEMMs of forward / backward / decode run on v_mfma_f32_32x32x16_bf16.  Widths above 512 have no bf16
        kernels and stay exact."""
        if self.precision != "bf16":
            self._float_precision = self.precision
        if self.features > 512 and not getattr(SIREN, "_warned_half_wide", False):
            SIREN._warned_half_wide = True
            logging.warning("SIREN.half(): no bf16 kernels above 512 features (this net has %d): it stays in exact fp32; "
                            "callers that round coordinates / outputs to fp16 around it get the I/O rounding only", self.features)
        return self._set_precision("bf16" if self.features <= 512 else self.precision)

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def requires_grad_(self, flag=True):
        """requires_grad_(True): forward() takes part in torch autograd (the reference's own loop body then runs on
        this module: zero_grad, forward, loss, backward, torch.optim step).  Default off: forward() returns plain
        tensors and the fused Fitter path is the fast one."""
        self._autograd = bool(flag)
        return self

    @property
    def device(self):
        return self.params.device

    # ---- fused path
    def _require_gpu(self):
        if self.params.device.type != "cuda":
            raise _lib.BriefError("the fused SIREN path needs the parameters on a ROCm GPU (module.to('cuda')); "
                                  "there is no CPU fallback")

    def sync_packed(self):
        """refresh the fragment-ordered weight copy after any change of self.params"""
        self._require_gpu()
        if self.packed is None:
            n = _lib.lib().brief_packed_count(C.byref(self.desc))
            if n < 0:      # (a shape the library refuses, e.g. precision = 'bf16' above 512 features: its message, not a torch allocation error)
                raise _lib.BriefError(_lib.lib().brief_last_error().decode())
            self.packed = torch.empty(n, dtype=torch.float32, device=self.params.device)
            self._stale = True
        if self.params._version != self._seen_version:      # torch changed the parameters in place (e.g. optimizer.step())
            self._stale = True
        if self._stale:
            _lib.check(_lib.lib().brief_siren_repack(C.byref(self.desc), _lib.ptr(self.params), _lib.ptr(self.packed), _lib.stream_ptr()))
            self._stale = False
            self._seen_version = self.params._version

    @staticmethod
    def _grid(dims, lo, hi):
        g = _lib.GridDesc()
        g.ndim = len(dims)
        for a, v in enumerate(dims):
            g.dims[a] = int(v)
        g.lo, g.hi = float(lo), float(hi)
        return g

    def forward(self, coords):
        """SIREN.forward (utils/Networks.py:269-271): coords [..., cin] -> [..., cout].  Differentiable w.r.t. the
        parameters after requires_grad_(True) (see _SirenFn); otherwise a plain no-grad evaluation."""
        if self._autograd and torch.is_grad_enabled():
            self._require_gpu()
            if self._anchor is None or self._anchor.device != self.params.device:
                self._anchor = torch.zeros(1, device=self.params.device, requires_grad=True)
            return _SirenFn.apply(self._anchor, coords.to(self.params.device, torch.float32).contiguous(), self)
        return self._forward_plain(coords)

    def _forward_plain(self, coords):
        self._require_gpu()
        self.sync_packed()
        c = coords.to(self.params.device, torch.float32).contiguous()
        lead = c.shape[:-1]
        n = int(np.prod(lead)) if len(lead) else 1
        out = torch.empty((n, self.data_channel), dtype=torch.float32, device=c.device)
        if n == 0:
            return out.view(*lead, self.data_channel)
        b = _lib.BatchDesc(c.data_ptr(), None, None, None, 0, n, 0, 0, 0)
        ws, ws_bytes = self._forward_scratch(n)
        _lib.check(_lib.lib().brief_siren_forward_ws(C.byref(self.desc), _lib.ptr(self.packed), None, C.byref(b), _lib.ptr(out),
                                                     _lib.OUT_F32, 0.0, 1.0, 0.0, 1.0, ws, ws_bytes, _lib.stream_ptr()))
        return out.view(*lead, self.data_channel)

    def _forward_scratch(self, n):
        """(pointer, bytes) of the inference scratch brief_siren_forward_ws wants: nothing up to 1024 features, two ping-pong
        activation planes per workgroup above (include/brief_hip.h); allocated once, kept"""
        need = _lib.lib().brief_forward_workspace_bytes(C.byref(self.desc), int(n))
        if need < 0:
            raise _lib.BriefError(_lib.lib().brief_last_error().decode())
        if need == 0:
            return None, 0
        if self._fws is None or self._fws.numel() * 4 < need:
            self._fws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.params.device)
        return _lib.ptr(self._fws), self._fws.numel() * 4

    def __call__(self, coords):
        return self.forward(coords)

    def decode_grid(self, dims, lo=-1.0, hi=1.0, offset=0, count=None, out=None, out_kind="f32",
                    scale=(0.0, 100.0), vrange=(0.0, 1.0)):
        """forward over `count` voxels of the flattened (d,h,w) grid starting at `offset`, coordinates
        synthesised in-kernel (replaces create_flattened_coords + the chunked loop of
        utils/misc.py:59-92).  out_kind 'u8'/'u16' fuses invnormalize_data (utils/io.py:136-147)."""
        self._require_gpu()
        self.sync_packed()
        total = int(np.prod(dims))
        count = total - offset if count is None else int(count)
        kind = {"f32": _lib.OUT_F32, "u8": _lib.OUT_U8, "u16": _lib.OUT_U16}[out_kind]
        dt = {"f32": torch.float32, "u8": torch.uint8, "u16": torch.uint16}[out_kind]
        if out is None:
            out = torch.empty((count, self.data_channel), dtype=dt, device=self.params.device)
        g = self._grid(dims, lo, hi)
        b = _lib.BatchDesc(None, None, None, None, int(offset), count, 0, 0, 0)
        ws, ws_bytes = self._forward_scratch(count)
        _lib.check(_lib.lib().brief_siren_forward_ws(C.byref(self.desc), _lib.ptr(self.packed), C.byref(g), C.byref(b), _lib.ptr(out),
                                                     kind, float(scale[0]), float(scale[1]), float(vrange[0]), float(vrange[1]),
                                                     ws, ws_bytes, _lib.stream_ptr()))
        return out

    def train_step(self, n, targets, idx=None, coords=None, weights=None, grid=None, offset=0,
                   loss="datal2", thr=0.0, beta=0.01, want_yhat=False):
        """zero_grad + forward + loss + backward of main.py:385-396 for one batch of n samples.
        Fills self.grads (canonical layout) and returns (loss [1] device tensor, yhat or None)."""
        self._require_gpu()
        self.sync_packed()
        dev = self.params.device
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
            self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        need = _lib.lib().brief_train_workspace_bytes(C.byref(self.desc), int(n))
        if need < 0:
            raise _lib.BriefError(_lib.lib().brief_last_error().decode())
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        yhat = torch.empty((n, self.data_channel), dtype=torch.float32, device=dev) if want_yhat else None
        g = None
        if coords is None:
            dims, lo, hi = grid
            g = self._grid(dims, lo, hi)
        b = _lib.BatchDesc(_dev_ptr(coords, torch.float32, "coords", dev), _dev_ptr(targets, torch.float32, "targets", dev),
                           _dev_ptr(weights, torch.float32, "weights", dev), _dev_ptr(idx, torch.int64, "idx", dev),
                           int(offset), int(n), 0, 0, 0)
        _lib.check(_lib.lib().brief_siren_train_step(
            C.byref(self.desc), _lib.ptr(self.packed), C.byref(g) if g is not None else None, C.byref(b),
            _lib.LOSS_KIND[loss], float(thr), float(beta), _lib.ptr(self.grads), _lib.ptr(self._loss), _lib.ptr(yhat),
            _lib.ptr(self._ws), self._ws.numel() * 4, _lib.stream_ptr()))
        return self._loss, yhat

    def fit_step(self, n, targets, opt_kind, s1, s2, lr, t, idx=None, weights=None, grid=None, offset=0,
                 loss="datal2", thr=0.0, beta=0.01, betas=(0.9, 0.999), eps=1e-8, rng=None):
        """train_step + optimizer update + refresh of the packed copy in one C-ABI call (three launches);
        bit-identical to the separate calls.  Returns the device loss tensor."""
        self._require_gpu()
        self.sync_packed()
        self.ensure_train_buffers(n)
        dims, lo, hi = grid
        g = self._grid(dims, lo, hi)
        pop, seed, step = rng if (rng is not None and idx is None) else (0, 0, 0)      # rng = (pop, seed, step): in-kernel sampling
        dev = self.params.device
        b = _lib.BatchDesc(None, _dev_ptr(targets, torch.float32, "targets", dev), _dev_ptr(weights, torch.float32, "weights", dev),
                           _dev_ptr(idx, torch.int64, "idx", dev), int(offset), int(n), int(pop), int(seed), int(step))
        _lib.check(_lib.lib().brief_siren_fit_step(
            C.byref(self.desc), _lib.ptr(self.params), _lib.ptr(self.packed), C.byref(g), C.byref(b),
            _lib.LOSS_KIND[loss], float(thr), float(beta), int(opt_kind), _lib.ptr(s1), _lib.ptr(s2),
            float(lr), betas[0], betas[1], eps, int(t), _lib.ptr(self.grads), _lib.ptr(self._loss),
            _lib.ptr(self._ws), self._ws.numel() * 4, _lib.stream_ptr()))
        return self._loss

    def ensure_train_buffers(self, n):
        """gradient / loss / workspace buffers for batches of n samples (allocated once, reused)."""
        self._require_gpu()
        dev = self.params.device
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
            self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        need = _lib.lib().brief_train_workspace_bytes(C.byref(self.desc), int(n))
        if need < 0:
            raise _lib.BriefError(_lib.lib().brief_last_error().decode())
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)

    # ---- budget -> width (utils/Networks.py:291-314)
    @staticmethod
    def calc_param_count(coords_channel, data_channel, features, layers, res=False, **kwargs):
        """utils/Networks.py:291-297: first layer + (layers-2) hidden F x F layers + head, weights and biases"""
        SIREN._no_res(res)
        F, hidden = features, layers - 2
        return int((coords_channel + 1) * F + hidden * (F + 1) * F + (F + 1) * data_channel)

    @staticmethod
    def calc_features(param_count, coords_channel, data_channel, layers, res=False, **kwargs):
        """utils/Networks.py:299-314: the positive root of hidden F^2 + (cin + 1 + hidden + cout) F + cout = P, rounded"""
        SIREN._no_res(res)
        hidden = layers - 2
        lin = coords_channel + 1 + hidden + data_channel
        if hidden == 0:
            return round((param_count - data_channel) / lin)
        return round((math.sqrt(lin * lin + 4 * hidden * (param_count - data_channel)) - lin) / (2 * hidden))

    @staticmethod
    def _no_res(res):
        if res:
            raise NotImplementedError("SIREN(res=True) is unsupported on the fused path")


def _dev_ptr(t, dtype, what, device):
    """data_ptr() of a tensor the C-ABI will read as `dtype`: wrong dtype / layout / device is an error here, not
    silently reinterpreted bits in the kernel"""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or t.dtype != dtype:
        raise _lib.BriefError("%s must be a %s tensor (got %s)" % (what, dtype, getattr(t, "dtype", type(t))))
    if not t.is_contiguous():
        raise _lib.BriefError("%s must be contiguous" % what)
    if t.device != device:
        raise _lib.BriefError("%s must live on %s (got %s)" % (what, device, t.device))
    return t.data_ptr()


def get_nnmodule_param_count(module):
    """utils/Networks.py:13-17"""
    return sum(int(np.prod(p.shape)) for p in module.state_dict().values())


# registry with the reference's names (utils/Networks.py:795-802).  Only SIREN exists on the fused
# path; every other phi.name of the reference raises instead of silently running something else.
ALLPHI = {"SIREN": SIREN}
ALL_CALC_PHI_FEATURES = {"SIREN": SIREN.calc_features}
ALL_CALC_PHI_PARAM_COUNT = {"SIREN": SIREN.calc_param_count}
ALL_CHECK_PARAM_COUNT = {}


def init_phi(kwargs):
    kwargs = copy.deepcopy(dict(kwargs))
    name = kwargs.pop("name")
    if name not in ALLPHI:
        raise NotImplementedError("Module.phi.name=%r is not available on the fused MI355X path (only SIREN)" % name)
    return ALLPHI[name](**kwargs)
