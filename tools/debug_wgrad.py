"""Which entries of dW / db differ from fp64 for one wgrad shape (debug aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import fused
dev = torch.device("cuda:0")
for (M, N, K, lm) in [(1000, 64, 32, True), (1024, 64, 32, True), (1000, 64, 64, False), (4099, 64, 128, False), (33, 64, 8, True), (1000, 1, 64, False), (777, 64, 40, True)]:
    g = torch.Generator().manual_seed(M)
    dp = torch.randn(M, N, generator=g).to(dev)
    if lm:
        F = 2 if K % 2 == 0 and K <= 32 else (4 if K % 4 == 0 else 1)
        F = 1 if K == 8 else F
        x = torch.randn(K // F, M, F, generator=g).to(dev)
        segs = [fused.seg_lm(x, 0)]
        xr = x.permute(1, 0, 2).reshape(M, K)
    else:
        x = torch.randn(M, K, generator=g).to(dev)
        segs = [fused.seg(x, 0, K)]
        xr = x
    dw, db = fused.wgrad(dp, segs, K)
    rw = dp.double().T @ xr.double(); rb = dp.double().sum(0)
    ew = (dw.double() - rw).abs() / rw.abs().max(); eb = (db.double() - rb).abs() / rb.abs().max()
    print(f"M={M} N={N} K={K} lm={lm}: dW max rel {ew.max().item():.2e}  db max rel {eb.max().item():.2e}  bad db idx {torch.nonzero(eb > 1e-4).flatten().tolist()[:40]}")
