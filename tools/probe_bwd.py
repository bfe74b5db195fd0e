"""Run ONLY the sliced grid-backward on a 1-level grid (for rocprofv3 PMC runs)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import _lib, ops
from tools.kbench import synth_rays
res = int(sys.argv[1]) if len(sys.argv) > 1 else 562
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
R, S, D, F, T = 8192, 128, 3, 2, 19
N = R * S
o, d = synth_rays(R, dev)
aabb = torch.tensor([-20.0, -40.0, 0.0, 80.0, 40.0, 20.0], device=dev)
cdf = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1)
s, t = ops.importance_sample(cdf, cdf, S, torch.rand(R, device=dev), stot=(0.1, 1000.0, "uniform_lindisp"))
x, _ = ops.ray_points(o, d, t[:, :-1].contiguous(), t[:, 1:].contiguous(), aabb, True)
x = x.view(N, D)
d1 = _lib.make_grid_desc(D, 1, F, T, res, 1.0)
dlm = torch.randn(1, N, F, device=dev)
g1 = torch.zeros(d1.n_entries * F, device=dev)
mk = ops.slice_masks(d1, x)
for _ in range(iters):
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(d1), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(g1), N, ops._stream(x))
torch.cuda.synchronize()
print("done")
