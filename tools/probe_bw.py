"""Reference points for HBM streaming on this box: torch reduction (read-only), copy (read+write), fill (write-only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit
dev = torch.device("cuda:0")
for mb in (256, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev).uniform_(); b = torch.empty_like(a)
    t, _ = timeit(lambda: a.sum(), iters=10); print(f"{mb} MiB sum   : {mb*1.048576e6/t/1e6:.2f} TB/s ({t:.0f} us)")
    t, _ = timeit(lambda: b.copy_(a), iters=10); print(f"{mb} MiB copy  : {2*mb*1.048576e6/t/1e6:.2f} TB/s ({t:.0f} us)")
    t, _ = timeit(lambda: b.fill_(1.0), iters=10); print(f"{mb} MiB fill  : {mb*1.048576e6/t/1e6:.2f} TB/s ({t:.0f} us)")
    t, _ = timeit(lambda: torch.add(a, b, out=b), iters=10); print(f"{mb} MiB add   : {3*mb*1.048576e6/t/1e6:.2f} TB/s ({t:.0f} us)")
