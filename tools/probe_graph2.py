import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd.trainer import Trainer, synthetic_rays
from emernerf_amd.render_utils import render_rays
dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
tr.step_count = 1000
for s in range(1000):
    tr.requires_grad_fn(s)
data = synthetic_rays(8192, dev, seed=1000)
jit = torch.full((8192,), 0.37, device=dev)
tr.estimator.jitter_fn = lambda n, d: jit
def fwd_bwd(pg):
    tr.flat.zero_grad()
    results = render_rays(radiance_field=tr.model, proposal_estimator=tr.estimator, proposal_networks=tr.props,
                          data_dict=data, cfg=tr.rcfg, proposal_requires_grad=pg)
    loss = tr.losses(results, data)
    (loss * tr.loss_scale).backward()
    return loss.detach()
for _ in range(5):
    fwd_bwd(False)
torch.cuda.synchronize()
ref = tr.flat.grads.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd_bwd(False)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd_bwd(False)
names = [(n, p) for n, p in tr.model.named_parameters()]
def report(tag):
    torch.cuda.synchronize()
    d = (tr.flat.grads - ref).abs()
    worst = max(((float((p.grad - ref[o:o + p.numel()].view(p.shape)).abs().max()), n) for (n, p), (pp, o) in zip(names, tr.flat._plist[:len(names)])), key=lambda t: t[0])
    print(tag, "max abs diff", float(d.max()), "worst param", worst, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
for i in range(40):
    g.replay()
    if mode == "sync":
        torch.cuda.synchronize()
    if i % 8 == 7:
        report(f"replay {i+1} ({mode})")
