"""sha256 of the parameters after a short fit, per net width — run once per library build (BRIEF_LIB=...) and diff the outputs: a change that
claims 'same bits' (a re-vectorised k_reduce, a re-ordered launch plan) must print identical lines.
    python tools/lib_checksum.py L F1,F2,... [steps] [n]"""
import hashlib, sys
import torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN

L = int(sys.argv[1])
Fs = [int(v) for v in sys.argv[2].split(',')]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
torch.manual_seed(1)
tv = torch.rand(128 ** 3, 1, device='cuda') * 100
for F in Fs:
    for opt in ('Adamax', 'Adam'):
        torch.manual_seed(7)
        m = SIREN(features=F, layers=L, w0=20).to('cuda')
        f = Fitter(m, tv, (128, 128, 128), sampler='randompoint', sample_size=n, optimizer=opt, lr=1e-3, seed=5)
        f.run(steps)
        torch.cuda.synchronize()
        h = hashlib.sha256(m.params.detach().cpu().numpy().tobytes()).hexdigest()[:16]
        hp = hashlib.sha256(m.packed.detach().cpu().numpy().tobytes()).hexdigest()[:16] if getattr(m, 'packed', None) is not None else '-'
        print("L=%d F=%d %s: params %s packed %s" % (L, F, opt, h, hp))
