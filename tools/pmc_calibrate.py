"""Known-byte-count dispatches for calibrating FETCH_SIZE / WRITE_SIZE on this rocprofv3 (MI355X_MICROARCH.md, HBM
section: gfx950 halves wide streaming reads; other widths must be calibrated).  Three access shapes that bracket the
grid kernels': a 16 B/lane streaming copy (torch), an 8 B/lane streaming copy, and the library's own level-major
<-> row-major transpose (8 B/lane reads, strided writes).  1 GiB per buffer, far beyond the 256 MiB Infinity Cache."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import ops

dev = torch.device("cuda:0")
n = 1 << 28                      # 1 GiB of fp32
a = torch.empty(n, device=dev).uniform_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)                   # elementwise copy kernel, 16 B/lane: reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
L, N, F = 16, 1 << 23, 2         # [L][N][F] fp32 = 1 GiB
lm = a.view(L, N, F)
for _ in range(3):
    ops.layout_transpose(lm, L, N, F, True)   # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("calibration done")
