import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import fused
from tools.kbench import timeit
dev = torch.device("cuda:0"); R, S = 8192, 128; N = R * S
which = sys.argv[1] if len(sys.argv) > 1 else "both"
a2 = torch.randn(N, 64, device=dev); dp2 = torch.randn(N, 3, device=dev)
a1 = torch.randn(N, 64, device=dev); g = torch.randn(N, 64, device=dev); hr = torch.randn(R, 49, device=dev); dp1 = torch.randn(N, 64, device=dev)
if which in ("both", "small"):
    print("dW2 (3x64) us", timeit(lambda: fused.wgrad(dp2, [fused.seg(a2, 0, 64)], 64), iters=5)[0])
if which in ("both", "big"):
    print("dW1 (64x177) us", timeit(lambda: fused.wgrad(dp1, [fused.seg(a1, 0, 64), fused.seg(hr, 64, 49, row_div=S), fused.seg(g, 113, 64)], 177), iters=5)[0])
