"""Run the register-resident heads alone at the metric shape: per-kernel timing and error against fp64.
usage: python tools/probe_heads.py [--lib tag] [iters]"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _libsel
from emernerf_amd import fused
from tools.kbench import timeit
import torch.nn.functional as F
dev = torch.device("cuda:0"); R, S, Kh = 8192, 128, 49; N = R * S
g = torch.Generator().manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev).requires_grad_(True)
enc = r(16, N, 2); hray = r(R, Kh)
Pn = [r(64, 32, k=.2), r(64, k=.1), r(64, 64, k=.1), r(64, k=.1)]
Pc = [r(64, Kh + 64, k=.1), r(64, k=.1), r(64, 64 + Kh + 64, k=.1), r(64, k=.1), r(3, 64, k=.1), r(3, k=.1)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res = {"lib": _libsel.TAG}
def fwd_neck(): return fused.neck(enc, *Pn)
geo, _, dens = fwd_neck()
def fwd_rgb(): return fused.rgb_head(hray, geo.detach().requires_grad_(True), S, *Pc)
res["neck_fwd_us"] = timeit(lambda: fwd_neck(), iters=iters)[0]
res["rgb_fwd_us"] = timeit(lambda: fwd_rgb(), iters=iters)[0]
def step():
    geo, _, dens = fused.neck(enc, *Pn)
    rgb = fused.rgb_head(hray, geo, S, *Pc)
    (rgb.sum() + dens.sum()).backward()
res["neck+rgb fwd+bwd+wgrad_us"] = timeit(step, iters=iters)[0]
# ---- error vs fp64 on the first 32 rays (torch fp64 on the GPU)
n = 32 * S
with torch.no_grad():
    x = enc[:, :n].permute(1, 0, 2).reshape(n, 32).double()
    h = torch.relu(F.linear(x, Pn[0].double(), Pn[1].double()))
    gref = F.linear(h, Pn[2].double(), Pn[3].double())
    hr = hray[:32].double().repeat_interleave(S, 0)
    xin = torch.cat([hr, gref], -1)
    a1 = torch.relu(F.linear(xin, Pc[0].double(), Pc[1].double()))
    a2 = torch.relu(F.linear(torch.cat([a1, xin], -1), Pc[2].double(), Pc[3].double()))
    rgb_ref = torch.sigmoid(F.linear(a2, Pc[4].double(), Pc[5].double()))
    geo_h = fused.neck(enc[:, :n].contiguous(), *Pn)[0]
    rgb_h = fused.rgb_head(hray[:32].contiguous(), geo_h, S, *Pc)
    res["geo_max_abs_err"] = float((geo_h.double() - gref).abs().max()); res["geo_scale"] = float(gref.abs().max())
    res["geo_rms_rel"] = float(((geo_h.double() - gref).pow(2).mean() / gref.pow(2).mean()).sqrt())
    res["rgb_max_abs_err"] = float((rgb_h.double() - rgb_ref).abs().max())
print(json.dumps(res))
