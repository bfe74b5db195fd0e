import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from emernerf_amd.trainer import Trainer, synthetic_rays
dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev, table_init=None)
datas = [synthetic_rays(8192, dev, seed=s) for s in range(4)]
torch.cuda.synchronize(); t0 = time.perf_counter()
losses = []
for i in range(1500):
    out = tr.train_step(datas[i % 4])
    if i % 250 == 0:
        losses.append(float(out["loss"]))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = tr.flat.params
print("steps/s", 1500 / dt, "losses", [round(l, 5) for l in losses], "finite params", bool(torch.isfinite(p).all()), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
