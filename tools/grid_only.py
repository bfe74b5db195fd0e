"""Launch only the main-grid forward (+bitmaps) and the owner-computes backward on the TRAINING sample distribution
(proposal-resampled points of a real step), a few times each: the target of tools/pmc_grid.sh (SQ / LDS / TCP counter
passes) and a quick A/B timer.  argv: [--uniform] [--iters K] [--grid D,L,base,max,T,F]"""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emernerf_amd._lib as _L0
if "--lib" in sys.argv:  # A/B: emernerf_amd/lib/libemernerf_<tag>.so built by tools/build_variant.sh
    _L0.LIB_PATH = os.path.join(os.path.dirname(_L0.LIB_PATH), f"libemernerf_{sys.argv[sys.argv.index('--lib') + 1]}.so")
from emernerf_amd import _lib, ops
from emernerf_amd.trainer import Trainer, synthetic_rays
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--uniform", action="store_true")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--grid", default="3,16,16,2048,19,2")
ap.add_argument("--table-init", type=float, default=None)
ap.add_argument("--lib", default=None)
ap.add_argument("--split-sweep", action="store_true", help="time the backward as two level-range launches [k, L) + [0, k) for every k")
args = ap.parse_args()
dev = torch.device("cuda:0")
D, L, base, mx, T, F = (int(v) for v in args.grid.split(","))
growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
desc = _lib.make_grid_desc(D, L, F, T, base, growth)
N = 8192 * 128
if args.uniform:
    x = torch.rand(N, D, device=dev)
else:
    tr = Trainer(kind="static", device=dev, table_init=args.table_init)
    data = synthetic_rays(8192, dev, seed=1000)
    from emernerf_amd.trainer import capture_main_grid_positions
    for _ in range(2):
        tr.train_step(data)
    cap = {"x": capture_main_grid_positions(tr, data)}
    x = cap["x"]
    if D == 4:
        x = torch.cat([x, data["normed_timestamps"][:, None].expand(-1, 128).reshape(-1, 1)], -1).contiguous()
    del tr
p = torch.rand(desc.n_entries * F, device=dev) - 0.5
dlm = torch.randn(L, N, F, device=dev)
grad = torch.empty(desc.n_entries * F, device=dev)
def fwd():
    return ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
_, mk = fwd()
def bwd():
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(grad), N, ops._stream(x))
def timeit(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]
f_us, b_us = timeit(fwd, args.iters), timeit(bwd, args.iters)
if args.split_sweep:
    def part(a, b):
        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(grad), N, a, b, ops._stream(x))
    rows = []
    for k in range(1, L):
        hi_us, lo_us = timeit(lambda: part(k, L), args.iters), timeit(lambda: part(0, k), args.iters)
        rows.append({"k": k, "first_us": round(hi_us, 1), "second_us": round(lo_us, 1), "sum_us": round(hi_us + lo_us, 1),
                     "first_bytes_frac": round(1.0 - float(desc.offset[k]) / desc.n_entries, 3)})
    print(json.dumps({"single_us": round(b_us, 1), "split": rows}))
f0_us = timeit(lambda: ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=False), args.iters)
fb = 4 * D + (2 ** D) * L * F * 4 + L * F * 4
bb = 4 * D + L * F * 4 + 2 * (2 ** D) * L * F * 4
print(json.dumps({"lib": args.lib or "base", "grid": args.grid, "dist": "uniform" if args.uniform else "training", "fwd_us": round(f_us, 1), "fwd_nomask_us": round(f0_us, 1), "bwd_us": round(b_us, 1), "grad_abs_sum": float(grad.double().abs().sum()),
                  "pair_frac_of_8TBps": round((fb + bb) * N / ((f_us + b_us) * 1e-6) / 8e12, 4)}))
