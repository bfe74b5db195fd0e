"""Is the step host-bound?  Enqueue time (no sync) vs GPU time per step, with and without the kernel timer."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import _lib
from emernerf_amd.trainer import Trainer, synthetic_rays

dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
tr.set_step(1000)
data = synthetic_rays(8192, dev, seed=1000)
for _ in range(6):
    tr.train_step(data)
torch.cuda.synchronize()
for label in ("no timer",):
    enq = []
    t0 = time.perf_counter()
    for _ in range(24):
        a = time.perf_counter()
        tr.train_step(data)
        enq.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(label, "enqueue ms/step", 1e3 * sum(enq) / 24, "min", 1e3 * min(enq), "total ms/step", 1e3 * (t2 - t0) / 24, "tail sync ms", 1e3 * (t2 - t1))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    tr.train_step(data)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
