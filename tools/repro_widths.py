"""Run-to-run bit-reproducibility of whole fits over net widths (a race in a kernel shows up as a parameter that differs between two identical fits):
    python tools/repro_widths.py L F1,F2,... [steps] [n]"""
import sys
import torch
sys.path.insert(0, '.')
from brief_pytorch_amd.fit import Fitter
from brief_pytorch_amd.networks import SIREN

L = int(sys.argv[1])
Fs = [int(v) for v in sys.argv[2].split(',')]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
torch.manual_seed(1)
tv = torch.rand(128 ** 3, 1, device='cuda') * 100
bad = 0
for F in Fs:
    outs = []
    for rep in range(2):
        torch.manual_seed(7)
        m = SIREN(features=F, layers=L, w0=20).to('cuda')
        fit = Fitter(m, tv, (128, 128, 128), sampler='randompoint', sample_size=n, seed=3)
        fit.run(steps)
        dec = m.decode_grid((64, 64, 64), out_kind='f32')
        torch.cuda.synchronize()
        outs.append((m.params.clone(), dec.clone()))
    same_p = torch.equal(outs[0][0], outs[1][0])
    same_d = torch.equal(outs[0][1], outs[1][1])
    finite = bool(torch.isfinite(outs[0][0]).all())
    print("%dx%d: %d steps of %d samples twice: parameters %s, decode %s%s" % (L - 1, F, steps, n, "identical" if same_p else "DIFFER", "identical" if same_d else "DIFFER",
                                                                                 "" if finite else "  NON-FINITE"), flush=True)
    bad += (not same_p) or (not same_d) or (not finite)
sys.exit(1 if bad else 0)
