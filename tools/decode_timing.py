"""scratch: decode throughput (forward-only kernels) fp32 vs bf16 vs bf16x3"""
import sys, torch, time
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
for (L, F) in ((5, 256), (9, 512)):
    for prec in ('fp32', 'bf16') + (('bf16x3',) if F <= 256 else ()):
        torch.manual_seed(0)
        m = SIREN(features=F, layers=L, w0=20, precision=prec).to('cuda')
        dims = (256, 256, 256)
        out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0))
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): out = m.decode_grid(dims, out_kind='u16', scale=(0.0, 100.0), vrange=(0.0, 65535.0), out=out)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        n = 256 ** 3
        M = 3 * F + (L - 2) * F * F + F
        print("%s L=%d F=%d decode 256^3: %.1f ms  %.1f Mvox/s  %.1f TFLOP/s" % (prec, L, F, dt * 1e3, n / dt / 1e6, 2 * M * n / dt / 1e12), flush=True)
