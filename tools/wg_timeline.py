"""per-workgroup start / end of k_fused<8,true> under the body + single-tile-tail launch plan (clock-only diagnostic build,
-DBRIEF_STAMPS=2 -> brief_pytorch_amd/libbrief_hip_clock.so):  BRIEF_TAIL_ROUNDS=r python tools/wg_timeline.py [n]"""
import os, sys
os.environ.setdefault("BRIEF_LIB", os.path.abspath("brief_pytorch_amd/libbrief_hip_clock.so"))
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tr = int(os.environ.get("BRIEF_TAIL_ROUNDS", "0"))
m = SIREN(features=256, layers=5, w0=20).to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (256, 256, 256), sample_size=n)
for _ in range(600): fit.step()
torch.cuda.synchronize()
FP, hidden = 256, 3
cap = 256 * int(os.environ.get("BRIEF_WG_PER_CU", "3"))      # resident workgroups (round 3: three per CU for the lean 8-tile kernel)
npad = (n + 31) // 32 * 32
ntiles = (n + 31) // 32
rounds = max(ntiles // cap - tr, 0) if tr > 0 else None
if rounds is None or rounds * cap >= ntiles: pers, ptiles, grid = cap, ntiles, cap
else: pers, ptiles = (cap if rounds else 0), rounds * cap; grid = pers + ntiles - ptiles
rec_off = 2 * hidden * FP * npad
rec = m._ws[rec_off:rec_off + grid * 4 * 1056].view(grid * 4, 1056).cpu().numpy()
r = rec[:, 1040:1045].reshape(grid, 4, 5)[:, 0]          # wave 0 of every workgroup: [cycles, ticks, start, xcc, hw_id]
life = r[:, 1] / 100.0
st = r[:, 2].copy()
st = (st - st[:min(grid, 256)].min()) % (1 << 24)        # 24-bit 100 MHz tick counter
start = st / 100.0
end = start + life
print('n=%d tail_rounds=%d: grid %d (persistent %d over %d tiles, %d single-tile); kernel span %.1f us' % (n, tr, grid, pers, ptiles, grid - pers, end.max()))
def cls(name, k):
    if k.sum(): print('  %-28s %4d wgs: start mean %6.1f (%6.1f..%6.1f)  life mean %6.1f (%6.1f..%6.1f)  end mean %6.1f max %6.1f' % (name, k.sum(), start[k].mean(), start[k].min(), start[k].max(), life[k].mean(), life[k].min(), life[k].max(), end[k].mean(), end[k].max()))
b = np.arange(grid)
for q in range(0, pers, 256): cls('persistent, slot %d' % (q // 256), (b >= q) & (b < q + 256) & (b < pers))
for q in range(0, max(grid - pers, 0), 256): cls('single-tile %d..%d' % (q, min(q + 256, grid - pers) - 1), (b >= pers + q) & (b < pers + q + 256))
h, e = np.histogram(end, bins=12)
print('  end-time histogram: ' + ' '.join('%.0f:%d' % (e[i + 1], h[i]) for i in range(len(h))))
busy = np.zeros(int(end.max()) + 2)
for s0, e0 in zip(start, end): busy[int(s0):int(e0) + 1] += 1
t = np.arange(len(busy))
print('  resident workgroups over time (us:count): ' + ' '.join('%d:%d' % (i, busy[i]) for i in range(0, len(busy), max(len(busy) // 24, 1))))
