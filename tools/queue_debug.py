"""Debug harness for the second-pair queue of the grid backward: guarded build (EMER_QUEUE_DEBUG) vs the no-queue build."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import emernerf_amd._build as B
B.build()
sh = os.path.join(ROOT, "tools", "build_variant.sh")
subprocess.check_call(["bash", sh, "qdbg", "hashgrid.hip", "1s|^|#define EMER_QUEUE_DEBUG 1\\n|"])
subprocess.check_call(["bash", sh, "noq", "hashgrid.hip", "1s|^|#define EMER_PAIR_QUEUE 0\\n|"])
libdir = os.path.join(ROOT, "emernerf_amd", "lib")
dev = torch.device("cuda:0")
from emernerf_amd import _lib, ops
D, L, base, mx, T, F = 3, 16, 16, 2048, 19, 2
growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
desc = _lib.make_grid_desc(D, L, F, T, base, growth)
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192 * 128
g = torch.Generator().manual_seed(0)
x = torch.rand(N, 3, generator=g).to(dev)
p = (torch.rand(desc.n_entries * F, generator=g) - 0.5).to(dev)
dlm = torch.randn(L, N, F, generator=g).to(dev)
_, mk = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
outs = {}
TAGS = ("noq", "qdbg") if "--hip" not in sys.argv else ("noq", "hip")
for tag in TAGS:
    lib = ctypes.CDLL(os.path.join(libdir, f"libemernerf_{tag}.so"))
    fn = lib.emer_hashgrid_bwd_params_sliced
    fn.argtypes = _lib.SIGNATURES["emer_hashgrid_bwd_params_sliced"]; fn.restype = ctypes.c_int
    if tag == "qdbg":
        lib.emer_debug_queue.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        lib.emer_debug_queue(N, None, 1)
    grad = torch.zeros(desc.n_entries * F, device=dev)
    rc = fn(ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(grad), N, ops._stream(x))
    torch.cuda.synchronize()
    print(tag, "rc", rc, flush=True)
    if tag == "qdbg":
        out = (ctypes.c_uint32 * 8)()
        lib.emer_debug_queue(N, out, 0)
        print("queue dbg [bad, appended, drained, -, last_e, head, count, -]:", list(out), flush=True)
    outs[tag] = grad
d = (outs["noq"] - outs[TAGS[1]]).abs().max().item()
print("max |noq - queue| =", d, "scale", outs["noq"].abs().max().item())
