"""decode throughput (forward-only kernels, fp32) over net widths: python tools/decode_widths.py L F1,F2,... (BRIEF_LIB selects the build)"""
import sys, time, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
L = int(sys.argv[1])
dims = (256, 256, 256)
for F in (int(v) for v in sys.argv[2].split(',')):
    torch.manual_seed(0)
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0), out=out)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    n = 256 ** 3
    M = 3 * F + (L - 2) * F * F + F
    print("fp32 %dx%d decode 256^3: %.1f ms  %.1f Mvox/s  %.1f TFLOP/s (%.3f of 157.3)" % (L - 1, F, dt * 1e3, n / dt / 1e6, 2 * M * n / dt / 1e12, 2 * M * n / dt / 157.3e12), flush=True)
