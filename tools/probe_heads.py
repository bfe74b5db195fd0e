"""Run the register-resident heads alone at the metric shape (for rocprofv3 PMC runs / timing)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import fused
from tools.kbench import timeit
dev = torch.device("cuda:0"); R, S, Kh = 8192, 128, 43; N = R * S
g = torch.Generator().manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev).requires_grad_(True)
enc = r(16, N, 2); hray = r(R, Kh)
Pn = [r(64, 32, k=.2), r(64, k=.1), r(64, 64, k=.1), r(64, k=.1)]
Pc = [r(64, Kh + 64, k=.1), r(64, k=.1), r(64, 64 + Kh + 64, k=.1), r(64, k=.1), r(3, 64, k=.1), r(3, k=.1)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def step():
    geo, _, dens = fused.neck(enc, *Pn)
    rgb = fused.rgb_head(hray, geo, S, *Pc)
    (rgb.sum() + dens.sum()).backward()
t, _ = timeit(step, iters=iters)
print("neck+rgb fwd+bwd+wgrad us", t)
