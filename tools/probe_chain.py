"""Time the fused MLP chains one by one at the metric shape (and serve as a rocprofv3 target)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import fused, _lib
from tools.kbench import timeit
which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda:0")
R, S = 8192, 128
N = R * S
g = torch.Generator().manual_seed(0)
def P(*shape, s=1.0): return (torch.randn(*shape, generator=g) * s).to(dev).requires_grad_(True)
res = {}
if which in ("all", "base"):
    enc, W0, b0, W1, b1 = P(16, N, 2), P(64, 32, s=0.2), P(64, s=0.1), P(64, 64, s=0.12), P(64, s=0.1)
    res["base_fwd_us"], _ = timeit(lambda: fused.base_mlp(enc, W0, b0, W1, b1), iters=10)
    f, d = fused.base_mlp(enc, W0, b0, W1, b1)
    gf, gd = torch.randn_like(f), torch.randn_like(d)
    res["base_bwd_us"], _ = timeit(lambda: torch.autograd.backward([f, d], [gf, gd], retain_graph=True), iters=5)
if which in ("all", "rgb", "rgbfwd"):
    hray, geo = P(R, 49), P(N, 64)
    Ws = [P(64, 113, s=0.1), P(64, s=0.1), P(64, 177, s=0.08), P(64, s=0.1), P(3, 64, s=0.12), P(3, s=0.1)]
    res["rgb_fwd_us"], _ = timeit(lambda: fused.rgb_head(hray, geo, S, *Ws), iters=10)
    if which != "rgbfwd":
        rgb = fused.rgb_head(hray, geo, S, *Ws)
        go = torch.randn_like(rgb)
        res["rgb_bwd_us"], _ = timeit(lambda: rgb.backward(go, retain_graph=True), iters=5)
if which in ("all", "dens"):
    enc, W0, b0, W1, b1 = P(8, N, 1), P(64, 8, s=0.3), P(64, s=0.1), P(1, 64, s=0.12), P(1, s=0.1)
    with torch.no_grad():
        res["density_fwd_nograd_us"], _ = timeit(lambda: fused.density_mlp(enc, W0, b0, W1, b1), iters=10)
    res["density_fwd_us"], _ = timeit(lambda: fused.density_mlp(enc, W0, b0, W1, b1), iters=10)
macs = {"base_fwd_us": 32 * 64 + 64 * 64, "rgb_fwd_us": 113 * 64 + 177 * 64 + 64 * 3, "density_fwd_nograd_us": 8 * 64 + 64}
for k, m in macs.items():
    if k in res: res[k.replace("_us", "_TF")] = 2.0 * m * N / res[k] / 1e6
print(json.dumps(res, indent=1))

if which in ("all", "detail"):
    # per-entry-point breakdown of one rgb-head backward and one base backward
    from emernerf_amd import _lib as L_
    import torch
    hray, geo = P(R, 49), P(N, 64)
    Ws = [P(64, 113, s=0.1), P(64, s=0.1), P(64, 177, s=0.08), P(64, s=0.1), P(3, 64, s=0.12), P(3, s=0.1)]
    rgb = fused.rgb_head(hray, geo, S, *Ws); go = torch.randn_like(rgb)
    enc, W0, b0, W1, b1 = P(16, N, 2), P(64, 32, s=0.2), P(64, s=0.1), P(64, 64, s=0.12), P(64, s=0.1)
    f, dn = fused.base_mlp(enc, W0, b0, W1, b1); gf, gd = torch.randn_like(f), torch.randn_like(dn)
    for _ in range(2):
        rgb.backward(go, retain_graph=True); torch.autograd.backward([f, dn], [gf, gd], retain_graph=True)
    for name, fn in (("rgb_bwd", lambda: rgb.backward(go, retain_graph=True)),
                     ("base_bwd", lambda: torch.autograd.backward([f, dn], [gf, gd], retain_graph=True))):
        t = L_.KernelTimer(["emer_mlp_chain", "emer_wgrad_segmented"]); L_.TIMER = t
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); L_.TIMER = None
        print(name, "total_us", round(e0.elapsed_time(e1) * 1e3), {k: [round(x) for x in v] for k, v in t.elapsed_us().items()})
