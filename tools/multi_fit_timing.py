"""scratch: blocks of a DivideTask partition trained one after the other vs co-trained (brief_multi_fit)"""
import sys, time, torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter, MultiFitter
from brief_pytorch_amd.networks import SIREN
def mk(L, F, dims, sampler, n, seed):
    torch.manual_seed(seed)
    m = SIREN(features=F, layers=L, w0=20).to('cuda')
    pop = dims[0] * dims[1] * dims[2]
    tv = torch.rand(pop, 1, device='cuda') * 100
    return Fitter(m, tv, dims, sampler=sampler, sample_size=n, seed=seed)
def bench(tag, shapes, steps=400):
    fs = [mk(*s, seed=i) for i, s in enumerate(shapes)]
    for f in fs: f.run(20)
    torch.cuda.synchronize(); t0 = time.time()
    for f in fs: f.run(steps)
    torch.cuda.synchronize(); t_seq = time.time() - t0
    mf = MultiFitter(fs); mf.run(20)
    torch.cuda.synchronize(); t0 = time.time()
    mf.run(steps)
    torch.cuda.synchronize(); t_multi = time.time() - t0
    nb = len(shapes)
    print("%-44s sequential %.3f s   co-trained %.3f s   x%.2f   (%.1f us per block-step co-trained, %.1f sequential)" %
          (tag, t_seq, t_multi, t_seq / t_multi, t_multi / steps / nb * 1e6, t_seq / steps / nb * 1e6), flush=True)
bench("8 x (5x22, 32^3 full batch)", [(5, 22, (32, 32, 32), 'full', 0)] * 8)
bench("8 x (6x22, 32^3 full batch)", [(6, 22, (32, 32, 32), 'full', 0)] * 8)
bench("64 x (5x22, 16^3 full batch)", [(5, 22, (16, 16, 16), 'full', 0)] * 64, steps=200)
bench("1 x (5x22, 64^3 full batch)", [(5, 22, (64, 64, 64), 'full', 0)] * 1)
bench("8 x (5x35, 64^3 full batch)", [(5, 35, (64, 64, 64), 'full', 0)] * 8)
bench("8 x (5x35, 128^3, 100k samples)", [(5, 35, (128, 128, 128), 'randompoint', 100000)] * 8)
bench("4 x (7x56, 32x256x256, 100k samples)", [(7, 56, (32, 256, 256), 'randompoint', 100000)] * 4)
bench("20 x (5x35, 64^3 full) adaptive-like", [(5, 35, (64, 64, 64), 'full', 0)] * 20, steps=200)
bench("2 x (5x256, 256^3, 100k samples)", [(5, 256, (256, 256, 256), 'randompoint', 100000)] * 2, steps=200)
