"""decode throughput of the narrow nets BRIEF's YAMLs produce (forward-only k_fused<NT,false>): python tools/decode_narrow.py"""
import sys, torch, time
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
for (L, F) in ((5, 22), (5, 35), (7, 56), (3, 64), (5, 96), (5, 128), (5, 160), (5, 192), (5, 224), (5, 256)):
    torch.manual_seed(0)
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    dims = (256, 256, 256)
    out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0), out=out)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    n = 256 ** 3
    M = 3 * F + (L - 2) * F * F + F
    FPd = (F + 31) // 32 * 32
    Mp = 3 * FPd + (L - 2) * FPd * FPd + FPd
    print("%dx%d decode 256^3: %.2f ms  %.0f Mvox/s  %.1f TFLOP/s algorithmic (%.3f of 157.3), %.1f padded" % (L - 1, F, dt * 1e3, n / dt / 1e6, 2 * M * n / dt / 1e12, 2 * M * n / dt / 1e12 / 157.3, 2 * Mp * n / dt / 1e12), flush=True)
