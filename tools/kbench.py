"""Per-kernel micro-benchmark on the metric shape (8192 rays x 128 samples), MI355X only.

Not the judged bench (that is bench.py); this is the inner-loop tool used to iterate on kernels.
Times with HIP events on torch's current stream (the stream every kernel is launched on).
"""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd import _lib, ops  # noqa: E402


def synth_rays(R, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.stack([torch.rand(R, generator=g) * 60, torch.rand(R, generator=g) * 4 - 2, torch.rand(R, generator=g) + 1.5], -1)
    d = torch.nn.functional.normalize(torch.tensor([1.0, 0.0, 0.0]) + 0.6 * torch.randn(R, 3, generator=g), dim=-1)
    return o.to(device), d.to(device)


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--grid", default="3,16,16,2048,19,2")
    ap.add_argument("--clustered", action="store_true", help="cluster samples near a surface (trained-like)")
    ap.add_argument("--per-level", action="store_true", help="time each level of the grid as a 1-level grid")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    R, S = args.rays, args.samples
    N = R * S
    D, L, base, mx, T, F = (int(v) for v in args.grid.split(","))
    growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
    desc = _lib.make_grid_desc(D, L, F, T, base, growth)
    o, d = synth_rays(R, dev)
    aabb = torch.tensor([-20.0, -40.0, 0.0, 80.0, 40.0, 20.0], device=dev)
    cdf = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1)
    jit = torch.rand(R, device=dev)
    if args.clustered:
        m = 65
        edges = torch.linspace(0, 1, m, device=dev)[None].repeat(R, 1)
        w = torch.full((R, m - 1), 1e-3, device=dev)
        hit = torch.randint(4, 40, (R,), device=dev)
        w[torch.arange(R), hit] = 1.0
        c = torch.cat([torch.zeros(R, 1, device=dev), torch.cumsum(w, -1)], -1)
        c = c / c[:, -1:]
        s, t = ops.importance_sample(edges, c, S, jit, stot=(0.1, 1000.0, "uniform_lindisp"))
    else:
        s, t = ops.importance_sample(cdf, cdf, S, jit, stot=(0.1, 1000.0, "uniform_lindisp"))
    ts, te = t[:, :-1].contiguous(), t[:, 1:].contiguous()
    times = torch.rand(R, device=dev) if D == 4 else None
    x, _ = ops.ray_points(o, d, ts, te, aabb, True, times=times)
    x = x.view(N, D)
    if args.per_level:
        import ctypes
        full = desc
        rows = []
        for l in range(L):
            r = int(full.res[l])
            d1 = _lib.make_grid_desc(D, 1, F, T, r, 1.0)
            assert int(d1.res[0]) == r, (r, int(d1.res[0]))
            p = torch.rand(d1.n_entries * F, device=dev) - 0.5
            dlm = torch.randn(1, N, F, device=dev)
            g1 = torch.zeros(d1.n_entries * F, device=dev)
            f_us, _ = timeit(lambda: ops.hashgrid_fwd_raw(d1, x, p, level_major=True), iters=10)
            a_us, _ = timeit(lambda: _lib.call("emer_hashgrid_bwd_params", ctypes.byref(d1), ops._ptr(x), ops._ptr(dlm), F, N * F,
                                               ops._ptr(g1), 0, N, ops._stream(x)), iters=5)
            mk = ops.slice_masks(d1, x)
            s_us, _ = timeit(lambda: _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(d1), ops._ptr(x), ops._ptr(dlm), F,
                                               N * F, ops._ptr(mk), ops._ptr(g1), N, ops._stream(x)), iters=5)
            rows.append({"level": l, "res": r, "entries": int(d1.n_entries), "hashed": int(d1.hashed[0]), "fwd_us": round(f_us, 1),
                         "bwd_atomic_us": round(a_us, 1), "bwd_sliced_us": round(s_us, 1)})
            print(rows[-1], flush=True)
        return
    res = {"shape": [R, S], "grid": [D, L, base, mx, T, F], "n_params": desc.n_entries * F, "clustered": args.clustered}
    inside = float((x[:, :3] != 0).any(-1).float().mean())
    res["frac_inside"] = inside
    for dt_name, dt in (("f32", torch.float32), ("f16", torch.float16)):
        p = ((torch.rand(desc.n_entries * F, device=dev) - 0.5)).to(dt)
        sp = 4 if dt == torch.float32 else 2
        med, best = timeit(lambda: ops.hashgrid_fwd_raw(desc, x, p, level_major=True))
        b_fwd = 4 * D + (2 ** D) * L * F * sp + L * F * 4
        res[f"fwd_{dt_name}_us"] = med
        res[f"fwd_{dt_name}_algGBps"] = b_fwd * N / med / 1e3
        med_rm, _ = timeit(lambda: ops.hashgrid_fwd_raw(desc, x, p, level_major=False))
        res[f"fwd_rowmajor_{dt_name}_us"] = med_rm
        dlm = torch.randn(L, N, F, device=dev)
        grad = torch.zeros(desc.n_entries * F, device=dev, dtype=dt)
        import ctypes

        def bwd():
            _lib.call("emer_hashgrid_bwd_params", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(grad),
                      ops._dtype_tag(grad), N, ops._stream(x))
        if not (dt == torch.float16 and F % 2):
            med, best = timeit(bwd)
            b_bwd = 4 * D + L * F * 4 + 2 * (2 ** D) * L * F * sp
            res[f"bwd_params_{dt_name}_us"] = med
            res[f"bwd_params_{dt_name}_algGBps"] = b_bwd * N / med / 1e3
        if dt == torch.float32:
            g2 = torch.empty(desc.n_entries * F, device=dev)
            _, mk = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
            med_m, _ = timeit(lambda: ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True))
            res["fwd_with_masks_f32_us"] = med_m

            def bwd_sl():
                _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F,
                          ops._ptr(mk), ops._ptr(g2), N, ops._stream(x))
            med, best = timeit(bwd_sl)
            res["bwd_params_sliced_f32_us"] = med
            res["bwd_params_sliced_f32_algGBps"] = (4 * D + L * F * 4 + 2 * (2 ** D) * L * F * 4) * N / med / 1e3
            grad.zero_(); bwd(); torch.cuda.synchronize()
            res["sliced_vs_atomic_maxrel"] = float((g2 - grad).abs().max() / grad.abs().max())
            dx = torch.empty_like(x)

            def bwd_in():
                _lib.call("emer_hashgrid_bwd_input", ctypes.byref(desc), ops._ptr(x), ops._ptr(p), 0, ops._ptr(dlm), F, N * F,
                          ops._ptr(dx), N, ops._stream(x))
            res["bwd_input_f32_us"], _ = timeit(bwd_in)
            med, _ = timeit(lambda: grad.zero_())
            res["zero_grad_f32_us"] = med
    lm = torch.randn(L, N, F, device=dev)
    res["transpose_us"], _ = timeit(lambda: ops.layout_transpose(lm, L, N, F, True))
    # MLP layers at N rows
    for (K, Nn, act) in [(32, 64, "relu"), (64, 64, None), (113, 64, "relu"), (177, 64, "relu"), (64, 3, "sigmoid"), (8, 64, "relu"), (64, 1, "trunc_exp")]:
        xx = torch.randn(N, K, device=dev, requires_grad=True)
        W = torch.randn(Nn, K, device=dev, requires_grad=True)
        b = torch.zeros(Nn, device=dev, requires_grad=True)
        med, _ = timeit(lambda: ops.linear(xx, W, b, act), iters=10)
        res[f"linear_fwd_{K}x{Nn}_us"] = med
        res[f"linear_fwd_{K}x{Nn}_TF"] = 2.0 * N * K * Nn / med / 1e6
        y = ops.linear(xx, W, b, act)
        go = torch.randn_like(y)

        def lb():
            y.backward(go, retain_graph=True)
        med, _ = timeit(lb, iters=5)
        res[f"linear_bwd_{K}x{Nn}_us"] = med
    # compositing
    sg = torch.rand(R, S, device=dev, requires_grad=True)
    res["render_weights_fwd_us"], _ = timeit(lambda: ops.render_weights(ts, te, sg))
    rgb = torch.rand(R, S, 3, device=dev)
    w = torch.rand(R, S, device=dev)
    res["accumulate3_fwd_us"], _ = timeit(lambda: ops.accumulate_along_rays(w, rgb))
    res["importance_sample_us"], _ = timeit(lambda: ops.importance_sample(s, s, S, jit, stot=(0.1, 1000.0, "uniform_lindisp")))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
