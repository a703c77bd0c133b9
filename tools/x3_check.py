"""BRIEF_PREC_BF16X3 against the oracle (f32 and f64) and against the fp32 path: one train step per shape, then step timing on the headline shape"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as O
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
for L, F, n in ((5, 256, 20000), (4, 128, 5000), (3, 200, 3333), (6, 96, 1000), (2, 64, 500)):
    rng = np.random.default_rng(F)
    x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    y = rng.uniform(0, 100, size=(n, 1)).astype(np.float32)
    d = O.make_desc(3, 1, L, F, 20.0)
    out = {}
    for prec in ("fp32", "bf16x3"):
        torch.manual_seed(L * 100 + F)
        m = SIREN(features=F, layers=L, w0=20.0, precision=prec)
        p = m.params.numpy().copy()
        m.to('cuda')
        loss, yh = m.train_step(n, torch.from_numpy(y).cuda(), coords=torch.from_numpy(x).cuda(), want_yhat=True)
        out[prec] = (loss.item(), yh.cpu().numpy(), m.grads.cpu().numpy())
    lo, go, yo, _ = O.loss_grad(d, p, x, y)
    l64, g64, y64, _ = O.loss_grad(d, p, x, y, f64=True)
    gw64, gb64 = O.unpack_params(d, g64)
    for prec in ("fp32", "bf16x3"):
        l, yh, g = out[prec]
        gw, gb = O.unpack_params(d, g)
        worst = max(max(relerr(gw[k], gw64[k]), relerr(gb[k], gb64[k])) for k in range(L))
        print("%dx%d n=%d %-7s vs f64 oracle: yhat %.2e loss %.2e worst gradient tensor %.2e" % (L - 1, F, n, prec, relerr(yh, y64), abs(l - l64) / l64, worst), flush=True)
for prec in ("fp32", "bf16x3"):
    torch.manual_seed(0)
    m = SIREN(features=256, layers=5, w0=20, precision=prec).to('cuda')
    tv = torch.rand(256 ** 3, 1, device='cuda') * 100
    fit = Fitter(m, tv, (256, 256, 256), sampler='randompoint', sample_size=100000)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        fit.run(100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fit.run(400); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 400
    print("4x256 100000 samples %-7s: %.4f ms/step = %.1f M voxels/s, loss %.4f" % (prec, ms, 100000 / ms / 1e3, float(m._loss)), flush=True)
