import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emernerf_amd import fused
dev = torch.device("cuda:0")
def timeit(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:60s} {a.elapsed_time(b) / n * 1000:8.2f} us")
H, K0, C = 64, 49, 3
for M in (64, 512, 2048, 8192):
    a1, a2, x = (torch.randn(M, w, device=dev) for w in (H, H, K0))
    d2, d1, d0 = (torch.randn(M, w, device=dev) for w in (C, H, H))
    dw2, dw1, dw0 = torch.zeros(C, H, device=dev), torch.zeros(H, H + K0, device=dev), torch.zeros(H, K0, device=dev)
    db2, db1, db0 = torch.zeros(C, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    timeit(f"M={M} sky 3 jobs", lambda: fused.ray_wgrad([(d2, [(a2, H, 0)], dw2, db2), (d1, [(a1, H, 0), (x, K0, H)], dw1, db1), (d0, [(x, K0, 0)], dw0, db0)], x))
    timeit(f"M={M} job dW1 only (64 x 114)", lambda: fused.ray_wgrad([(d1, [(a1, H, 0), (x, K0, H)], dw1, db1)], x))
    timeit(f"M={M} job dW0 only (64 x 50)", lambda: fused.ray_wgrad([(d0, [(x, K0, 0)], dw0, db0)], x))
    timeit(f"M={M} job dW2 only (3 x 65)", lambda: fused.ray_wgrad([(d2, [(a2, H, 0)], dw2, db2)], x))
