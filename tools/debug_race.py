import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emernerf_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ws = torch.empty(4 + 2048, dtype=torch.int32, device=dev)
seed = torch.zeros(1, dtype=torch.int64, device=dev)
for n in (64, 4096, 100000, 1000000, 7680000):
    big = torch.rand(n, generator=g).to(dev)
    big[::3] = 0.0
    k = min(2048, n // 4)
    o = torch.full((k,), -1, dtype=torch.int64, device=dev)
    _lib.call("emer_sample_importance", ops._ptr(big), n, ops._ptr(seed), 5, k, ops._ptr(ws), ops._ptr(o), ops._stream(big))
    torch.cuda.synchronize()
    oc = o.cpu()
    print(n, k, "state", ws[:4].cpu().tolist(), "distinct", len(set(oc.tolist())), "min", int(oc.min()), "max", int(oc.max()), flush=True)
