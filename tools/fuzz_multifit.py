"""Randomised check of the two 'identical results' claims of the fit drivers: (1) Fitter.run(k) == k x Fitter.step(), (2) a group of
fits trained together (MultiFitter / brief_multi_fit, HIP streams) == each fit run on its own — bitwise, for random groups of nets
(narrow k_small nets, general fp32 nets, bf16 nets, mixed), random samplers, batch sizes and step counts.
    python tools/fuzz_multifit.py [groups] [seed]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from brief_pytorch_amd.fit import Fitter, MultiFitter
from brief_pytorch_amd.networks import SIREN

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)

def make(spec, seed):
    L, F, prec, dims, sampler, n, opt = spec
    torch.manual_seed(seed)
    m = SIREN(features=F, layers=L, w0=20, precision=prec).to('cuda')
    g = torch.Generator(device='cpu').manual_seed(seed + 1000)
    tv = (torch.rand(int(np.prod(dims)), 1, generator=g) * 100).cuda()
    return Fitter(m, tv, dims, sampler=sampler, sample_size=n, optimizer=opt, lr=1e-3, seed=seed,
                  scheduler={"name": "MultiStepLR", "milestones": [7, 13], "gamma": 0.5})

bad = 0
for gi in range(groups):
    k = int(rng.integers(1, 7)) if gi % 3 else int(rng.integers(8, 80))      # every third group: many jobs (several k_small_group launches of up to 64)
    specs = []
    for _ in range(k):
        F = int(rng.choice([5, 22, 33, 56, 64, 96, 130, 200, 256, 340, 527, 1100])) if k < 8 else int(rng.choice([5, 22, 22, 30, 33, 56, 64, 96]))
        L = int(rng.integers(3, 8))
        prec = 'bf16' if (96 <= F <= 512 and rng.random() < 0.3) else 'fp32'      # (the bf16 path stops at 512 features)
        dims = tuple(int(v) for v in rng.choice([8, 12, 16, 20, 24], size=3))
        sampler = str(rng.choice(['full', 'randompoint']))
        n = int(rng.choice([100, 1000, 3333, 5000, 9000, 20000])) if sampler == 'randompoint' else 0
        specs.append((L, F, prec, dims, sampler, n, str(rng.choice(['Adamax', 'Adam']))))
    steps = int(rng.integers(3, 25))
    alone = [make(s, 10 * gi + i) for i, s in enumerate(specs)]
    for f in alone: f.run(steps)
    stepped = [make(s, 10 * gi + i) for i, s in enumerate(specs)]
    for f in stepped:
        for _ in range(steps): f.step()
    together = [make(s, 10 * gi + i) for i, s in enumerate(specs)]
    MultiFitter(together).run(steps)
    torch.cuda.synchronize()
    for i, s in enumerate(specs):
        pa, ps, pt = alone[i].m.params, stepped[i].m.params, together[i].m.params
        ok = torch.equal(pa, ps) and torch.equal(pa, pt) and bool(torch.isfinite(pa).all())
        if not ok:
            bad += 1
            print("FAIL group %d fit %d %s steps %d: run-vs-step %s, alone-vs-together %s" % (gi, i, s, steps, torch.equal(pa, ps), torch.equal(pa, pt)), flush=True)
    if gi % 5 == 0:
        print("ok   group %d: %d fits, %d steps" % (gi, k, steps), flush=True)
print("%d groups, %d mismatching fits" % (groups, bad))
sys.exit(1 if bad else 0)
