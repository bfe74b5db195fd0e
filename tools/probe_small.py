"""Timing of the per-ray kernels (csrc/rayinputs.hip) in isolation: HIP events over back-to-back launches."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emernerf_amd import _lib

dev = torch.device("cuda:0")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(name, fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:50s} {a.elapsed_time(b) / n * 1000:8.2f} us")


R, E, P = 8192, 16, 33
for n_emb in (1, 50, 1000):
    for stride in (1, 128):
        idx_full = torch.randint(0, n_emb, (R, stride), device=dev)
        idx = idx_full[:, 0]
        ga, gb = torch.randn(R, P + E, device=dev), torch.randn(R, P + E, device=dev)
        dw = torch.zeros(n_emb, E, device=dev)
        timeit(f"embed_grad n_emb={n_emb} idx_stride={stride}",
               lambda: _lib.call("emer_embed_grad", ga[:, P:].data_ptr(), P + E, gb[:, P:].data_ptr(), P + E, idx.data_ptr(), idx.stride(0), R, n_emb, E,
                                 dw.data_ptr(), st()))
dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
idx = torch.randint(0, 50, (R,), device=dev)
w = torch.randn(50, E, device=dev)
o1, o2 = torch.empty(R, P + E, device=dev), torch.empty(R, P + E, device=dev)
timeit("ray_inputs_fwd", lambda: _lib.call("emer_ray_inputs_fwd", dirs.data_ptr(), 3, idx.data_ptr(), 1, w.data_ptr(), 50, E, 4, R, o1.data_ptr(), P + E,
                                           o2.data_ptr(), P + E, st()))
H, Kh, NG = 64, 49, 64
W0, W1 = torch.randn(H, Kh + NG, device=dev), torch.randn(H, H + Kh + NG, device=dev)
b0, b1 = torch.randn(H, device=dev), torch.randn(H, device=dev)
h = torch.randn(R, Kh, device=dev); rb = torch.empty(R, 2 * H, device=dev)
wb = W1[:, H:]
timeit("ray_pre_fwd", lambda: _lib.call("emer_ray_pre_fwd", h.data_ptr(), Kh, R, Kh, H, W0.data_ptr(), W0.stride(0), b0.data_ptr(), wb.data_ptr(), W1.stride(0),
                                        b1.data_ptr(), rb.data_ptr(), 2 * H, st()))
s0, s1, dh = torch.randn(R, H, device=dev), torch.randn(R, H, device=dev), torch.empty(R, Kh, device=dev)
timeit("ray_pre_bwd", lambda: _lib.call("emer_ray_pre_bwd", s0.data_ptr(), s1.data_ptr(), H, R, Kh, H, W0.data_ptr(), W0.stride(0), wb.data_ptr(), W1.stride(0),
                                        dh.data_ptr(), Kh, st()))
W2, b2 = torch.randn(3, H, device=dev), torch.randn(3, device=dev)
a1, a2, out = torch.empty(R, H, device=dev), torch.empty(R, H, device=dev), torch.empty(R, 3, device=dev)
timeit("ray_head_fwd", lambda: _lib.call("emer_ray_head_fwd", rb.data_ptr(), 2 * H, R, W1.data_ptr(), W1.stride(0), W2.data_ptr(), b2.data_ptr(), 3, _lib.ACT_SIGMOID,
                                         a1.data_ptr(), a2.data_ptr(), out.data_ptr(), st()))
dout = torch.randn(R, 3, device=dev); d2, d1, d0 = torch.empty(R, 3, device=dev), torch.empty(R, H, device=dev), torch.empty(R, H, device=dev)
timeit("ray_head_bwd", lambda: _lib.call("emer_ray_head_bwd", dout.data_ptr(), out.data_ptr(), a1.data_ptr(), a2.data_ptr(), R, W1.data_ptr(), W1.stride(0), W2.data_ptr(), 3,
                                         _lib.ACT_SIGMOID, d2.data_ptr(), d1.data_ptr(), d0.data_ptr(), st()))
x = torch.empty(1 << 20, device=dev)
timeit("torch fill 4 MB (launch floor reference)", lambda: x.fill_(1.0))

# emer_ray_wgrad: per-job cost and scaling with the row count (one workgroup per 256-row chunk and job)
from emernerf_amd import fused
C = 3
for M in (64, 2048, 8192):
    a1, a2, x = (torch.randn(M, w, device=dev) for w in (H, H, Kh))
    e2, e1, e0 = (torch.randn(M, w, device=dev) for w in (C, H, H))
    dw2, dw1, dw0 = torch.zeros(C, H, device=dev), torch.zeros(H, H + Kh, device=dev), torch.zeros(H, Kh, device=dev)
    db2, db1, db0 = torch.zeros(C, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    timeit(f"ray_wgrad M={M}: sky head, 3 jobs", lambda: fused.ray_wgrad([(e2, [(a2, H, 0)], dw2, db2), (e1, [(a1, H, 0), (x, Kh, H)], dw1, db1),
                                                                       (e0, [(x, Kh, 0)], dw0, db0)], x))
    timeit(f"ray_wgrad M={M}: 64 x 114 job alone", lambda: fused.ray_wgrad([(e1, [(a1, H, 0), (x, Kh, H)], dw1, db1)], x))
