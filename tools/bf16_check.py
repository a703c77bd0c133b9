"""scratch: bf16 path vs fp32 path on the same parameters (forward, gradients, a short fit)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
for (L, F, n) in [(3, 256, 1000), (5, 256, 5000), (5, 200, 777), (4, 512, 3000), (9, 512, 20000), (2, 300, 500)]:
    torch.manual_seed(L * 1000 + F)
    m32 = SIREN(features=F, layers=L, w0=20).to('cuda')
    m16 = SIREN(features=F, layers=L, w0=20, precision='bf16').to('cuda')
    m16.params.copy_(m32.params); m16._stale = True
    g = torch.Generator().manual_seed(n)
    x = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
    y = (torch.rand(n, 1, generator=g) * 100).cuda()
    o32 = m32.forward(x); o16 = m16.forward(x)
    torch.cuda.synchronize()
    l32, _ = m32.train_step(n, y, coords=x); g32 = m32.grads.clone()
    l16, _ = m16.train_step(n, y, coords=x); g16 = m16.grads.clone()
    torch.cuda.synchronize()
    # per-layer gradient agreement
    F_, cin = F, 3
    offs = [0, F_ * cin + F_]
    for l in range(L - 2): offs.append(offs[-1] + F_ * F_ + F_)
    offs.append(offs[-1] + F_ + 1)
    per = [rel(g16[offs[i]:offs[i + 1]], g32[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]
    print("L=%d F=%d n=%d: fwd rel %.2e (max abs %.3e)  loss %.6g vs %.6g  grads rel %.2e  per-layer %s" % (
        L, F, n, rel(o16, o32), float((o16 - o32).abs().max()), float(l16), float(l32), rel(g16, g32), ["%.1e" % v for v in per]), flush=True)
# short fit: bf16 vs fp32 loss curves
for prec in ("fp32", "bf16"):
    torch.manual_seed(0)
    m = SIREN(features=256, layers=5, w0=20, precision=prec).to('cuda')
    tv = torch.rand(64 ** 3, 1, generator=torch.Generator().manual_seed(3)).cuda() * 100
    fit = Fitter(m, tv, (64, 64, 64), sample_size=50000)
    losses = [float(fit.step()) for _ in range(300)]
    print(prec, ["%.2f" % v for v in losses[::50]], "%.3f" % losses[-1], flush=True)
