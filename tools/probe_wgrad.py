"""Time emer_wgrad_segmented on the shapes of one training step (1M rows)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _libsel
from emernerf_amd import fused
from tools.kbench import timeit
dev = torch.device("cuda:0"); R, S = 8192, 128; N = R * S
a2 = torch.randn(N, 64, device=dev); dp2 = torch.randn(N, 3, device=dev)
a1 = torch.randn(N, 64, device=dev); g = torch.randn(N, 64, device=dev); dp1 = torch.randn(N, 64, device=dev)
enc = torch.randn(16, N, 2, device=dev); d1 = torch.randn(N, 1, device=dev); enc1 = torch.randn(8, N, 1, device=dev)
fa = torch.randn(N, device=dev); fb = torch.rand(N, device=dev)
cases = {
    "dW2 3x64": lambda: fused.wgrad(dp2, [fused.seg(a2, 0, 64)], 64),
    "dW1 64x128 [a1|geo]": lambda: fused.wgrad(dp1, [fused.seg(a1, 0, 64), fused.seg(g, 64, 64)], 128, want_bias=False),
    "dW0g 64x64": lambda: fused.wgrad(dp1, [fused.seg(g, 0, 64)], 64, want_bias=False),
    "neck dW1 64x64 +fix+bias": lambda: fused.wgrad(dp1, [fused.seg(a1, 0, 64)], 64, col0=fa),
    "neck dW0 64x32 lm": lambda: fused.wgrad(dp1, [fused.seg_lm(enc, 0)], 32),
    "prop dW1 1x64": lambda: fused.wgrad(d1, [fused.seg(a1, 0, 64)], 64),
    "prop dW0 64x8 lm": lambda: fused.wgrad(dp1, [fused.seg_lm(enc1, 0)], 8),
}
for k, f in cases.items():
    print(f"{k:28s} {timeit(f, iters=8)[0]:8.1f} us (includes workspace alloc + reduce)")
# accuracy against fp64 (the dW1 case: 1M-row reduction)
dw, _ = fused.wgrad(dp1, [fused.seg(a1, 0, 64), fused.seg(g, 64, 64)], 128, want_bias=False)
ref = dp1.double().T @ torch.cat([a1, g], 1).double()
err = (dw.double() - ref).abs().max().item(); sc = ref.abs().max().item()
print(f"lib {_libsel.TAG}: dW1 max abs err {err:.3e} / max |dW| {sc:.3e} = {err / sc:.2e}; rms rel {((dw.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item():.2e}")
