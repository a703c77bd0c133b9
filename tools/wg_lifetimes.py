"""per-workgroup start / end of k_fused<8,true> (clock-only diagnostic build, -DBRIEF_STAMPS=2) in steady state: who finishes late?"""
import os, sys
os.environ.setdefault("BRIEF_LIB", os.path.abspath("brief_pytorch_amd/libbrief_hip_clock.so"))
import torch, numpy as np
sys.path.insert(0, '.')
from brief_pytorch_amd.networks import SIREN
from brief_pytorch_amd.fit import Fitter
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = SIREN(features=256, layers=5, w0=20).to('cuda')
tv = torch.rand(256 ** 3, 1, device='cuda') * 100
fit = Fitter(m, tv, (256, 256, 256), sample_size=n)
for _ in range(600): fit.step()
torch.cuda.synchronize()
FP, hidden = 256, 3
npad = (n + 31) // 32 * 32
rec_off = 2 * hidden * FP * npad
rec = m._ws[rec_off:rec_off + 2048 * 1056].view(2048, 1056).cpu().numpy()
r = rec[:, 1040:1045].reshape(512, 4, 5)[:, 0]          # wave 0 of every workgroup: [cycles, ticks, start, xcc, hw_id]
life = r[:, 1] / 100.0
start = (r[:, 2] - r[:, 2].min()) / 100.0
end = start + life
xcc = r[:, 3].astype(int)
hw = r[:, 4].astype(int)
cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
ntiles = (n + 31) // 32
cnt = np.array([len(range(b, ntiles, 512)) for b in range(512)])
print('n=%d: kernel span (first start -> last end) %.1f us; start spread %.1f us; lifetime mean %.1f min %.1f max %.1f; end mean %.1f' % (n, end.max(), start.max(), life.mean(), life.min(), life.max(), end.mean()))
for c in sorted(set(cnt)):
    k = cnt == c
    print('  %d tiles: %3d workgroups, lifetime mean %.1f (min %.1f max %.1f), end mean %.1f max %.1f' % (c, k.sum(), life[k].mean(), life[k].min(), life[k].max(), end[k].mean(), end[k].max()))
print('  by XCC: ' + '  '.join('%d: n=%d life %.0f end<=%.0f' % (x, (xcc == x).sum(), life[xcc == x].mean(), end[xcc == x].max()) for x in range(8)))
print('  first slot (block < 256) life %.1f, second slot life %.1f; start of second slot %.1f' % (life[:256].mean(), life[256:].mean(), start[256:].mean()))
key = xcc * 1000 + se * 100 + sh * 20 + cu
pairs = {}
for b in range(512): pairs.setdefault(key[b], []).append(b)
sizes = np.bincount([len(v) for v in pairs.values()])
print('  workgroups per physical CU (count of CUs): ' + ', '.join('%d wg: %d' % (i, c) for i, c in enumerate(sizes) if c))
order = np.argsort(-end)[:12]
print('  latest finishers: ' + '; '.join('b%d x%d se%d cu%d life %.0f start %.0f' % (b, xcc[b], se[b], cu[b] + 16 * sh[b], life[b], start[b]) for b in order))
d = [abs(life[v[0]] - life[v[1]]) for v in pairs.values() if len(v) == 2]
if d: print('  |lifetime difference| inside a co-resident pair: mean %.1f max %.1f us' % (np.mean(d), np.max(d)))
pm = np.array([np.mean(life[v]) for v in pairs.values()])
print('  per-CU mean lifetime: min %.1f max %.1f std %.1f' % (pm.min(), pm.max(), pm.std()))
