"""Work-item timeline of the owner-computes grid backward (debug build with -DEMER_SLICED_TRACE).

Builds emernerf_amd/lib/libemernerf_trace.so from hashgrid.hip with the trace macro, runs the backward on the training
sample distribution and prints: kernel span, per-level item statistics, busy / idle time per persistent workgroup
(what the tail costs) and the 12 longest items.  Usage (GPU box): python tools/trace_sliced.py [--uniform]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import emernerf_amd._build as B
B.build()
extra = "".join(f"#define {d.replace('=', ' ')}\\n" for d in os.environ.get("EMER_TRACE_DEFS", "").split())  # e.g. "EMER_DENSE_ITEMS=1024"
subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "trace", "hashgrid.hip", f"1s|^|#define EMER_SLICED_TRACE 1\\n{extra}|"])
import emernerf_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libemernerf_trace.so")
import numpy as np
import torch
from emernerf_amd import ops
from emernerf_amd.trainer import Trainer, synthetic_rays

dev = torch.device("cuda:0")
gs = sys.argv[sys.argv.index("--grid") + 1] if "--grid" in sys.argv else "3,16,16,2048,19,2"   # e.g. --grid 4,10,32,8192,18,4 (xyzt tables)
D, Lv, base, mx, T, F = (int(v) for v in gs.split(","))
growth = float(np.exp((np.log(mx) - np.log(base)) / (Lv - 1)))
desc = L.make_grid_desc(D, Lv, F, T, base, growth)
N = 8192 * 128
if "--uniform" in sys.argv:
    x = torch.rand(N, D, device=dev)
else:
    tr = Trainer(kind="static", device=dev, table_init=0.3 if "--clustered" in sys.argv else None)
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
dlm = torch.randn(Lv, N, F, device=dev)
grad = torch.empty(desc.n_entries * F, device=dev)
_, mk = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
lib = L.load()
buf = torch.zeros(8 + 4 * 8192, device=dev, dtype=torch.int64)
lib.emer_debug_sliced_trace.argtypes = [ctypes.c_void_p]
def run():
    L.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(grad), N, ops._stream(x))
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.emer_debug_sliced_trace(ctypes.c_void_p(buf.data_ptr())) == 0
run()
torch.cuda.synchronize()
lib.emer_debug_sliced_trace(None)
b = buf.cpu().numpy().astype(np.uint64)
n = int(b[0])
rec = b[8:8 + 4 * n].reshape(n, 4)
level = (rec[:, 0] & 0xFF).astype(int); slc = ((rec[:, 0] >> 8) & 0xFFFF).astype(int); rng = ((rec[:, 0] >> 24) & 0xFFFF).astype(int)
blk = (rec[:, 0] >> 40).astype(int)
t0 = rec[:, 1].astype(np.float64); t1 = rec[:, 2].astype(np.float64)
TICK = 0.01  # wall_clock64: 100 MHz -> 10 ns
start = t0.min()
dur = (t1 - t0) * TICK
print(f"items {n}; kernel span {(t1.max() - start) * TICK:.1f} us; sum of item time {dur.sum() / 256:.1f} us per workgroup (256 workgroups)")
print("level res hashed items   mean_us   max_us   first_start  last_end   hits/item(wave0 x16)")
for l in range(Lv):
    m = level == l
    if not m.any():
        continue
    print(f"{l:5d} {int(desc.res[l]):5d} {int(desc.hashed[l]):3d} {int(m.sum()):6d} {dur[m].mean():9.1f} {dur[m].max():8.1f} {(t0[m].min() - start) * TICK:10.1f} {(t1[m].max() - start) * TICK:10.1f} {rec[m, 3].astype(np.float64).mean() * 16:12.0f}")
busy = np.zeros(blk.max() + 1); last = np.zeros(blk.max() + 1)
for i in range(n):
    busy[blk[i]] += dur[i]; last[blk[i]] = max(last[blk[i]], (t1[i] - start) * TICK)
print(f"workgroup busy us: min {busy.min():.1f} mean {busy.mean():.1f} max {busy.max():.1f}; last-finish us: min {last.min():.1f} mean {last.mean():.1f} max {last.max():.1f}")
for x8 in range(8):
    mm = (np.arange(len(busy)) % 8) == x8
    print(f"  xcd {x8}: busy mean {busy[mm].mean():.1f}  finish mean {last[mm].mean():.1f} max {last[mm].max():.1f}")
order = np.argsort(-dur)[:12]
print("longest items (level, slice, range, us, start_us):", [(int(level[i]), int(slc[i]), int(rng[i]), round(float(dur[i]), 1), round(float((t0[i] - start) * TICK), 1)) for i in order])
