"""Stability soak: N optimizer steps of a model (argv[2]: static | dynamic | flow | feature, default static; argv[3]: rays) from scratch (training schedule from step 0: the proposal nets train on
every early step), eager and hipGraph replay; the loss must fall, the parameters stay finite, the memory high-water mark flat."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd.trainer import Trainer, synthetic_rays  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
kind = sys.argv[2] if len(sys.argv) > 2 else "static"
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
kw = dict(num_cams=3, feature_dim=64) if kind == "feature" else {}
for use_graph in (False, True):
    tr = Trainer(kind=kind, device=dev, table_init=None, use_graph=use_graph)
    datas = [synthetic_rays(rays, dev, seed=s, **kw) for s in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses, mem = [], []
    for i in range(steps):
        out = tr.train_step(datas[i % 4])
        if i % (steps // 6) == 0:
            losses.append(float(out["loss"]))
            mem.append(round(torch.cuda.max_memory_allocated() / 1e9, 2))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = tr.flat.params
    print(kind, rays, "graph" if use_graph else "eager", "steps/s", round(steps / dt, 1), "losses", [round(v, 5) for v in losses], "finite params",
          bool(torch.isfinite(p).all()), "mem GB", mem)
    assert torch.isfinite(p).all() and losses[-1] < losses[0]
    del tr
    torch.cuda.empty_cache()
