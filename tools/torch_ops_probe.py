"""Which torch operators (and which autograd nodes) are behind the at::native launches of a step: one eager step under torch.profiler.
usage: python tools/torch_ops_probe.py [kind] [rays]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from emernerf_amd.trainer import Trainer, synthetic_rays
kind = sys.argv[1] if len(sys.argv) > 1 else "flow"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
tr = Trainer(kind=kind, device=dev)
tr.set_step(1000)
kw = dict(num_cams=3, feature_dim=64) if kind == "feature" else {}
data = synthetic_rays(rays, dev, seed=1, **kw)
for _ in range(4): tr.train_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and e.cpu_parent is not None:
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name); par = par.cpu_parent
        # only leaf aten ops (those that launch): no aten child with device time
        if not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
            rows.append((e.device_time_total, e.name, str(e.input_shapes)[:70], " <- ".join(chain)[:110]))
rows.sort(reverse=True)
for t, n, shp, ch in rows[:40]:
    print(f"{t:8.1f} us  {n:28s} {shp:70s} {ch}")
