"""Is the falling PSNR of WIDE nets at lr = 1e-3 (profiles/r05_rate_distortion.md: 4x768 34.7 dB, 4x1024 25.4 dB after 20 000 steps against 41 dB at 384 features)
the model's training dynamics or this library's?  The same net, initial weights, optimizer (torch.optim.Adamax, lr 1e-3) and sampler statistics in PLAIN PyTorch
(nn.Linear + torch.sin on the GPU through rocBLAS, autograd) beside the fused path, on one textured 256^3 volume; and the fused path again at lr / 4.
    python tools/wide_dynamics_check.py [F] [steps] [nolow|low] [edge]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from brief_pytorch_amd import _lib
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.synthetic import make_volume_torch

F = int(sys.argv[1]) if len(sys.argv) > 1 else 768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
E, n, L, w0 = (int(sys.argv[4]) if len(sys.argv) > 4 else 256), 100000, 5, 20.0
dims, vox = (E, E, E), E ** 3
vol = make_volume_torch(dims, seed=42, detail=64)
vf = vol.to(torch.int32).to(torch.float32)
vmin, vmax = float(vf.min()), float(vf.max())
tv = ((vf - vmin) / (vmax - vmin) * 100.0).reshape(vox, 1).contiguous()

def psnr_of(dec_norm):      # dec_norm: [vox] in the 0 .. 100 normalisation
    d = (dec_norm.clamp(0, 100) / 100.0 * (vmax - vmin) + vmin).to(torch.float64) - vol.reshape(-1).to(torch.float64)
    return -10.0 * np.log10(float((d * d).mean()) / 65535.0 ** 2)

def fused(lr):
    torch.manual_seed(42)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=w0).to("cuda")
    fit = Fitter(m, tv, dims, sampler="randompoint", sample_size=n, seed=42, lr=lr)
    out = []
    for k in range(3):
        fit.run(steps // 3)
        out.append(psnr_of(m.decode_grid(dims).reshape(-1)))
    return out

def plain():
    torch.manual_seed(42)
    m = SIREN(coords_channel=3, data_channel=1, features=F, layers=L, w0=w0).to("cuda")      # the same initial weights
    ws = [m.net[l][0].weight.data.clone().requires_grad_(True) for l in range(L)]
    bs = [m.net[l][0].bias.data.clone().requires_grad_(True) for l in range(L)]
    opt = torch.optim.Adamax(ws + bs, lr=1e-3)
    lin = [torch.linspace(-1, 1, E, device="cuda") for _ in range(3)]
    def fwd(x):
        h = x
        for l in range(L - 1):
            h = torch.sin((w0 if l == 0 else 30.0) * (h @ ws[l].t() + bs[l]))
        return h @ ws[L - 1].t() + bs[L - 1]
    g = torch.Generator(device="cuda").manual_seed(7)
    out = []
    for k in range(3):
        for _ in range(steps // 3):
            idx = torch.randint(0, vox, (n,), device="cuda", generator=g)
            x = torch.stack([lin[0][idx // (E * E)], lin[1][(idx // E) % E], lin[2][idx % E]], -1)
            loss = ((fwd(x) - tv[idx]) ** 2).mean()
            opt.zero_grad(); loss.backward(); opt.step()
        with torch.no_grad():
            dec = torch.cat([fwd(torch.stack([lin[0][i // (E * E)], lin[1][(i // E) % E], lin[2][i % E]], -1)).reshape(-1)
                             for i in torch.arange(vox, device="cuda").split(1 << 20)])
        out.append(psnr_of(dec))
    return out

print("4x%d on a textured %d^3 volume, PSNR dB after %d / %d / %d steps" % (F, E, steps // 3, 2 * (steps // 3), steps))
print("  fused path, Adamax lr 1e-3   : %s" % " ".join("%.2f" % v for v in fused(1e-3)), flush=True)
print("  plain PyTorch, Adamax lr 1e-3: %s" % " ".join("%.2f" % v for v in plain()), flush=True)
if len(sys.argv) <= 3 or sys.argv[3] == 'low':
    print("  fused path, Adamax lr 2.5e-4 : %s" % " ".join("%.2f" % v for v in fused(2.5e-4)), flush=True)
