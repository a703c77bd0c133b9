"""accuracy of the HIP train step against the oracle's f32 AND f64 instantiations (one full-size step, 100 000 samples):
   BRIEF_LIB=... python tools/parity_report.py      prints yhat / loss / worst gradient tensor, and the oracle's own f32-vs-f64 distance"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import oracle as O
from test_gpu_fullsize import _full_size_case, _tensor_errs, N
for L, F, dims in ((5, 256, (256, 256, 256)), (3, 64, (64, 64, 64)), (5, 22, (64, 64, 64)), (7, 56, (128, 128, 128))):
    m, p, tgt, idx, x, y = _full_size_case(L, F, dims, seed=11)
    loss, yhat = m.train_step(N, tgt, idx=idx, grid=(dims, -1.0, 1.0), want_yhat=True)
    d = O.make_desc(3, 1, L, F, 20.0)
    lo, go, yo, _ = O.loss_grad(d, p, x, y)
    l64, g64, y64, _ = O.loss_grad(d, p, x, y, f64=True)
    g = m.grads.cpu().numpy(); yh = yhat.cpu().numpy()
    print("%dx%d: vs oracle f32: yhat %.2e loss %.2e grads %.2e | vs oracle f64: yhat %.2e loss %.2e grads %.2e | oracle f32 vs f64: yhat %.2e loss %.2e grads %.2e" % (
        L - 1, F, np.max(np.abs(yh - yo)) / np.max(np.abs(yo)), abs(loss.item() - lo) / lo, max(_tensor_errs(g, go, L, F)),
        np.max(np.abs(yh - y64)) / np.max(np.abs(y64)), abs(loss.item() - l64) / l64, max(_tensor_errs(g, g64.astype(np.float64), L, F)),
        np.max(np.abs(yo - y64)) / np.max(np.abs(y64)), abs(lo - l64) / l64, max(_tensor_errs(go.astype(np.float64), g64.astype(np.float64), L, F))), flush=True)
