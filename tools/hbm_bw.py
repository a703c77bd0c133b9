"""achievable HBM bandwidth on this box with plain torch kernels (streaming read: sum; read + write: copy_; write: fill_), for
pricing the load-bound kernels (k_wgrad_x3, k_wgrad16_big) against what the memory system delivers rather than the 8 TB/s nominal"""
import torch, time
dev = 'cuda'
for gb in (0.6, 1.4, 4.0):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn, bytes_ in (("read  (sum)", lambda: x.sum(), 4 * n), ("copy  (r+w)", lambda: y.copy_(x), 8 * n), ("write (fill)", lambda: y.fill_(1.0), 4 * n),
                             ("read  (bf16 view max)", lambda: x.view(torch.bfloat16).amax(), 4 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%.1f GB %-22s %.3f ms  %.2f TB/s" % (gb, name, ms, bytes_ / ms / 1e9), flush=True)
