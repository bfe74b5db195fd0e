"""Per-level time of the sliced grid backward on the TRAINING sample distribution (proposal-resampled points)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import _lib, ops
from emernerf_amd.trainer import Trainer, synthetic_rays
from tools.kbench import timeit
dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
data = synthetic_rays(8192, dev, seed=1000)
captured = {}
orig = tr.model.contract_points
def hook(p):
    out = orig(p)
    captured["x"] = out.detach().reshape(-1, 3).contiguous()
    return out
tr.model.contract_points = hook
tr.train_step(data)
x = captured["x"]
N, D, F, T = x.shape[0], 3, 2, 19
full = tr.model.xyz_encoder.tcnn_encoding.desc
tot = 0.0
for l in range(full.n_levels):
    r = int(full.res[l])
    d1 = _lib.make_grid_desc(D, 1, F, T, r, 1.0)
    dlm = torch.randn(1, N, F, device=dev)
    g1 = torch.zeros(d1.n_entries * F, device=dev)
    mk = ops.slice_masks(d1, x)
    t, _ = timeit(lambda: _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(d1), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk),
                                    ops._ptr(g1), N, ops._stream(x)), iters=5)
    tot += t
    print(f"level {l:2d} res {r:5d} hashed {int(d1.hashed[0])}  {t:7.1f} us")
print("sum", round(tot, 1))
