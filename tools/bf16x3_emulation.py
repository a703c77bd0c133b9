"""Would split-precision hidden GEMMs (round-2 verdict, stretch item: hi/lo bf16 splits of W and of the activations, three
v_mfma_f32_32x32x16_bf16 per product, f32 accumulation) meet the fp32 parity bands?  CPU emulation on the 4x256 SIREN (reference
init, w0 = 20, 20 000 random coordinates): forward error of yhat against the float64 evaluation, relative to max|y| (band: 2e-5),
for plain f32, bf16 x 1, x 3 (hi.hi + hi.lo + lo.hi), x 4 (+ lo.lo) and three-way splits x 6.      python tools/bf16x3_emulation.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from brief_pytorch_amd.networks import SIREN
torch.manual_seed(0)
L, F, w0 = 5, 256, 20.0
m = SIREN(features=F, layers=L, w0=w0)
Ws = [m.net[l][0].weight.data.clone() for l in range(L)]
bs = [m.net[l][0].bias.data.clone() for l in range(L)]
x = (torch.rand(20000, 3) * 2 - 1)

def bf(t): return t.to(torch.bfloat16).to(torch.float32)
def split(t, n):
    parts, r = [], t.clone()
    for _ in range(n):
        p = bf(r); parts.append(p); r = r - p
    return parts

def forward(mode):
    h = x.double() if mode == "f64" else x.clone()
    for l in range(L):
        W, b = Ws[l], bs[l]
        if mode == "f64":
            z = h @ W.double().T + b.double()
        elif mode == "f32" or l == 0 or l == L - 1:       # first layer and head stay f32 (as on the bf16 path)
            z = h.float() @ W.T + b
        else:
            n = {"x1": 1, "x3": 2, "x4": 2, "x6": 3}[mode]
            hp, wp = split(h.float(), n), split(W, n)
            z = torch.zeros(h.shape[0], W.shape[0])
            for i in range(n):
                for j in range(n):
                    if mode == "x3" and i + j > 1: continue
                    if mode == "x6" and i + j > 2: continue
                    z = z + hp[i] @ wp[j].T
            z = z + b
        if l < L - 1:
            om = w0 if l == 0 else 30.0
            h = torch.sin(om * z)
        else:
            h = z
    return h.double()

ref = forward("f64")
scale = ref.abs().max().item()
print("4x256 SIREN, 20 000 coordinates; max |yhat - yhat_f64| / max|y|   (fp32 parity band of the forward pass: 2e-5)")
for mode, label in (("f32", "f32 (what the fp32 path computes)"), ("x1", "bf16 x 1 (the bf16 path)"), ("x3", "bf16 x 3: hi.hi + hi.lo + lo.hi"),
                    ("x4", "bf16 x 4: + lo.lo"), ("x6", "three-way splits, 6 products")):
    e = (forward(mode) - ref).abs().max().item() / scale
    print("  %-40s %.2e" % (label, e))


# ---- gradients: manual backward with the same split products (dgrad: W^T and delta split; wgrad: delta and h split)
def mm(a, b, mode):
    """a @ b with the hidden-GEMM arithmetic of `mode` (a: [n,k], b: [k,m])"""
    if mode == "f64": return a.double() @ b.double()
    if mode == "f32": return a.float() @ b.float()
    n = {"x1": 1, "x3": 2, "x4": 2, "x6": 3}[mode]
    ap, bp = split(a.float(), n), split(b.float(), n)
    z = torch.zeros(a.shape[0], b.shape[1])
    for i in range(n):
        for j in range(n):
            if mode == "x3" and i + j > 1: continue
            if mode == "x6" and i + j > 2: continue
            z = z + ap[i] @ bp[j]
    return z

def loss_grads(mode, y):
    dt = torch.float64 if mode == "f64" else torch.float32
    hs, zs = [x.to(dt)], []
    for l in range(L):
        W, b = Ws[l].to(dt), bs[l].to(dt)
        hidden = 0 < l < L - 1
        z = (mm(hs[-1], W.T, mode) if hidden else hs[-1] @ W.T) + b
        zs.append(z)
        hs.append(torch.sin((w0 if l == 0 else 30.0) * z) if l < L - 1 else z)
    yh = hs[-1]
    n = yh.shape[0]
    g = 2.0 * (yh - y.to(dt)) / n
    gW, gb = [None] * L, [None] * L
    delta = g
    for l in range(L - 1, -1, -1):
        hidden = 0 < l < L - 1
        gW[l] = mm(delta.T, hs[l], mode) if hidden else delta.T @ hs[l]
        gb[l] = delta.sum(0)
        if l > 0:
            back = mm(delta, Ws[l].to(dt), mode) if hidden else delta @ Ws[l].to(dt)
            om = w0 if l - 1 == 0 else 30.0
            delta = back * (om * torch.cos(om * zs[l - 1]))
    return ((yh - y.to(dt)) ** 2).mean().item(), gW, gb

y = torch.rand(x.shape[0], 1) * 100
l64, gW64, gb64 = loss_grads("f64", y)
print("loss and worst gradient tensor (max-abs error relative to the tensor's max-abs) against float64   (bands: 1e-5 / 1e-4)")
for mode in ("f32", "x1", "x3", "x4", "x6"):
    lo, gW, gb = loss_grads(mode, y)
    worst = max(max(((gW[l].double() - gW64[l]).abs().max() / gW64[l].abs().max()).item(), ((gb[l].double() - gb64[l]).abs().max() / gb64[l].abs().max()).item()) for l in range(L))
    print("  %-6s loss %.2e   gradients %.2e" % (mode, abs(lo - l64) / l64, worst))
