import ctypes, json, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from emernerf_amd import _lib, ops
from emernerf_amd.trainer import Trainer, synthetic_rays, capture_main_grid_positions
dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
data = synthetic_rays(8192, dev, seed=1000)
for _ in range(2): tr.train_step(data)
x = capture_main_grid_positions(tr, data)   # [1M, 3] positions of the final round (proposal-resampled)
del tr
def timeit(fn, iters=12):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[iters // 2]
for (D, L, base, mx, T, F) in [(3, 8, 16, 512, 20, 1), (3, 8, 16, 2048, 20, 1)]:
    growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
    desc = _lib.make_grid_desc(D, L, F, T, base, growth)
    p = torch.rand(desc.n_entries * F, device=dev) - 0.5
    N = x.shape[0]
    dlm = torch.randn(L, N, F, device=dev)
    dx = torch.empty(N, D, device=dev)
    st = ops._stream(x)
    f = timeit(lambda: ops.hashgrid_fwd_raw(desc, x, p, level_major=True))
    b = timeit(lambda: _lib.call("emer_hashgrid_bwd_input", ctypes.byref(desc), ops._ptr(x), ops._ptr(p), 0, ops._ptr(dlm), F, N * F, ops._ptr(dx), N, st))
    print(json.dumps({"grid": [D, L, base, mx, T, F], "fwd_level_major_blocks_us": round(f, 1), "all_levels_per_thread_gather_us": round(b, 1)}))
