"""[r6] Time emer_hashgrid_fwd (with slice bitmaps, as a training step asks for them) of one grid on proposal-like positions.
usage: python tools/r06_fwd_probe.py [--lib tag] --grid D,L,base,max,T,F [--n 1048576]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _libsel  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from emernerf_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="3,8,16,512,20,1")
ap.add_argument("--n", type=int, default=1 << 20)
a = ap.parse_args()
D, L, base, mx, T, F = [int(v) for v in a.grid.split(",")]
growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
desc = _lib.make_grid_desc(D, L, F, T, base, growth)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# ray-ordered positions: 128 consecutive samples share a ray (neighbouring samples fall into neighbouring cells, as in a step)
R, S = a.n // 128, 128
o = torch.rand(R, 1, D, generator=g) * 0.5 + 0.25
d = torch.nn.functional.normalize(torch.randn(R, 1, D, generator=g), dim=-1)
t = torch.sort(torch.rand(R, S, 1, generator=g) ** 2, dim=1).values * 0.5
x = (o + d * t).clamp(0.0, 1.0).reshape(-1, D).contiguous().to(dev)
p = ((torch.rand(desc.n_entries * F, generator=g) - 0.5)).to(dev)
for _ in range(5):
    ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out, m = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"{_libsel.TAG:6s} grid {a.grid:22s} n {a.n}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us  checksum {float(out.double().sum()):.6f}")
