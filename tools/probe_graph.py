"""Can a whole training step (forward + backward, no optimizer) be captured into a hipGraph, and what does replay cost?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd.trainer import Trainer, synthetic_rays
from emernerf_amd.render_utils import render_rays

dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
tr.set_step(1000)
data = synthetic_rays(8192, dev, seed=1000)
for _ in range(12):
    tr.train_step(data)
torch.cuda.synchronize()

def fwd_bwd(prop_grad):
    tr.flat.zero_grad()
    results = render_rays(radiance_field=tr.model, proposal_estimator=tr.estimator, proposal_networks=tr.props,
                          data_dict=data, cfg=tr.rcfg, proposal_requires_grad=prop_grad)
    if prop_grad:
        tr.estimator.compute_loss(results["extras"]["trans"], loss_scaler=tr.loss_scale).backward()
    loss = tr.losses(results, data)
    (loss * tr.loss_scale).backward()
    return loss.detach()

jit = torch.full((8192,), 0.37, device=dev)
tr.estimator.jitter_fn = lambda n, d: jit  # deterministic jitter: graph and eager gradients must then agree
ref = {}
for pg in (False, True):
    fwd_bwd(pg)
    torch.cuda.synchronize()
    ref[pg] = tr.flat.grads.clone()
graphs = {}
side = torch.cuda.Stream()
for pg in (False, True):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd(pg)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd_bwd(pg)
    graphs[pg] = (g, out)
    print("captured prop_grad =", pg, flush=True)
torch.cuda.synchronize()
# eager reference gradient vs replayed gradient on the same jitter?  (jitter differs per replay; compare magnitudes only)
for pg in (False, True):
    g, out = graphs[pg]
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    d = (tr.flat.grads - ref[pg]).abs().max() / ref[pg].abs().max()
    print("graph vs eager grad max rel diff", float(d), "ref |grad| sum", float(ref[pg].abs().sum()))
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"prop_grad={pg}: replay+adam {dt*1e3:.3f} ms/step, loss {float(out):.6f}, |grad| {float(tr.flat.grads.abs().sum()):.4e}")
t0 = time.perf_counter()
for _ in range(50):
    tr.train_step(data)
torch.cuda.synchronize()
print(f"eager train_step {(time.perf_counter()-t0)/50*1e3:.3f} ms/step (1 in 6 with prop grad)")
