"""how long after idle does the shader clock take to reach its steady value under the fp32 train step?  (clock-only diagnostic build)"""
import os, sys, time
os.environ.setdefault("BRIEF_LIB", os.path.abspath("brief_pytorch_amd/libbrief_hip_clock.so"))
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
torch.manual_seed(0)
n = 100000
m = SIREN(features=256, layers=5, w0=20).to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (256, 256, 256), sample_size=n)
rec_off = 2 * 3 * 256 * n
def clock():
    torch.cuda.synchronize()
    rec = m._ws[rec_off:rec_off + 2048 * 1056].view(2048, 1056)[:, 1040:1042].cpu().numpy()
    ok = rec[:, 1] > 0
    return float(np.median(rec[ok, 0] / rec[ok, 1])) * 0.1, float(np.median(rec[ok, 1])) / 100
torch.cuda.synchronize(); time.sleep(2.0)
done = 0
for upto in (1, 2, 5, 10, 20, 50, 100, 200, 400, 800):
    while done < upto:
        fit.step(); done += 1
    c, life = clock()
    print('after %4d steps since idle: %.3f GHz, workgroup lifetime %.0f us' % (done, c, life), flush=True)
