"""Which torch ops (and from which source lines) launch the small glue kernels of a training step."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd.trainer import Trainer, synthetic_rays

dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
tr.step_count = 1001
for s in range(1001):
    tr.requires_grad_fn(s)
data = synthetic_rays(8192, dev, seed=1000)
for _ in range(8):
    tr.train_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(6):
        tr.train_step(data)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::mul", "aten::sum", "aten::clone",
        "aten::contiguous", "aten::zeros", "aten::index_select", "aten::embedding", "aten::addmm", "aten::mm")
cnt = collections.Counter()
for e in prof.events():
    if e.name in want and e.device_time_total > 0 or (e.name in want and any(k.device_time > 0 for k in getattr(e, "kernels", []))):
        frames = [f for f in (e.stack or []) if "/root/repo" in f or "emernerf_amd" in f]
        where = frames[0].split("/")[-1] if frames else (e.stack[0] if e.stack else "?")
        cnt[(e.name, where[:90])] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c / 6:6.1f}/step  {name:18s} {where}")
