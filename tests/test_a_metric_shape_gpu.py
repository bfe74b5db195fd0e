"""Parity at the BENCHMARKED shape (collected first on purpose: the file name sorts before the other GPU tests).

The kernel tests in test_kernels_gpu.py run a few thousand samples, which never reaches the code the benchmark
exercises: at N = 8192 x 128 = 1 048 576 samples the owner-computes grid backward makes 16 trips over its bitmap per
work item (prefetched words), dense levels use 128-word ranges merged with atomics, tables larger than 64 LDS slices
share bitmaps (``gsub > 0``), the fused heads run 4096-row weight-gradient blocks.  Here every hot kernel is compared
with the CPU oracle (oracle/emer_oracle.c, OpenMP) at that size on the BASELINE.json grids, on uniform AND on
proposal-resampled ("training") sample positions, and one full optimizer step's gradients are compared with
oracle/ref_path.py parameter by parameter.

Tolerances (same as the small tests): grid forward 2e-6 abs; grid gradients 2e-5 x max; step gradients 2e-3 x max.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_METRIC = 8192 * 128

GRIDS = {
    # name: (D, L, base, max, log2T, F)            reference
    "cfg2_static": (3, 16, 16, 2048, 19, 2),       # HashEncoder defaults, encodings.py:110-118 (BASELINE configs[1])
    "default_static": (3, 10, 16, 8192, 20, 4),    # default_config.yaml:62-69  (256 LDS slices -> shared bitmaps)
    "dynamic_xyzt": (4, 10, 32, 8192, 18, 4),      # default_config.yaml:70-77
    "flow_xyzt": (4, 10, 16, 4096, 18, 4),         # radiance_field.py:916-923 (has a DENSE level 0; the dynamic grid has none)
    "prop1": (3, 8, 16, 2048, 20, 1),              # default_config.yaml:51-58 (second proposal net)
}

_CACHE = {}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _training_positions():
    """Contracted sample positions of a REAL proposal-resampled training batch (8192 rays x 128 samples): what
    PropNetEstimator.sampling + emer_ray_points hand to the main grid inside a step (clustered along each ray)."""
    if "x" not in _CACHE:
        from emernerf_amd.trainer import Trainer, capture_main_grid_positions, synthetic_rays
        dev = _dev()
        tr = Trainer(kind="static", device=dev, table_init=0.5, seed=11)
        data = synthetic_rays(8192, dev, seed=1000)
        cap = {"x": capture_main_grid_positions(tr, data)}
        torch.cuda.synchronize()
        _CACHE["x"] = cap["x"].cpu()
        _CACHE["t"] = data["normed_timestamps"].cpu()
        del tr
        torch.cuda.empty_cache()
    return _CACHE["x"], _CACHE["t"]


def _positions(dist: str, D: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    if dist == "uniform":
        x = torch.rand(N_METRIC, D, generator=g)
        # edge cases of the small tests, plus coordinates OUTSIDE [0, 1] (tcnn wraps them through the index
        # arithmetic; ADVICE r1: the paired-corner fast path must not assume in-range cells)
        x[0] = 0.0
        x[1] = 1.0 - 2 ** -24
        x[2] = 2 ** -20
        x[3, 0] = 0.9671
        x[4] = 1.0
        x[5, 0] = -0.37
        x[6, 0] = 1.61
        x[7] = torch.tensor([2.5, -1.25, 0.5, 0.3][:D])
        x[70000:70064, 0] = -0.013  # a whole wave of out-of-range hits somewhere in the middle of the stream
        return x
    x3, t = _training_positions()
    assert x3.shape[0] == N_METRIC
    if D == 3:
        return x3.clone()
    tt = t[:, None].expand(-1, 128).reshape(-1, 1)
    return torch.cat([x3, tt], -1).contiguous()


def _mk(oracle, name):
    from emernerf_amd import _lib
    D, L, base, mx, T, F = GRIDS[name]
    meta = oracle.grid_meta_from_encoder_args(D, L, base, mx, T, F)
    desc = _lib.make_grid_desc(D, L, F, T, base, meta.per_level_scale)
    return meta, desc


@pytest.mark.parametrize("dist", ["uniform", "training"])
@pytest.mark.parametrize("name", list(GRIDS))
def test_hashgrid_metric_shape(hip_lib, oracle, name, dist):
    """fwd, owner-computes bwd_params (through the bitmaps the forward emitted) and bwd_input at N = 1 048 576."""
    from emernerf_amd import ops
    meta, desc = _mk(oracle, name)
    D, L, F = meta.n_dims, meta.n_levels, meta.n_features
    assert ops.sliced_supported(desc), "every shipped grid must take the owner-computes backward"
    x = _positions(dist, D, seed=21)
    g = torch.Generator().manual_seed(22)
    p = torch.rand(meta.n_params, generator=g) - 0.5
    dout = torch.randn(N_METRIC, L * F, generator=g)
    dout[9] = 0.0
    dev = _dev()
    xd = x.to(dev).requires_grad_(D == 4)   # the xyzt grids are the ones that need input gradients (flow configs)
    pd = p.to(dev).requires_grad_(True)
    # level-major tensors, exactly what the fused heads exchange with the grid kernels inside a step
    lm = ops.hashgrid_encode_lm(xd, pd, desc)
    dlm = dout.view(N_METRIC, L, F).permute(1, 0, 2).contiguous().to(dev)
    lm.backward(dlm)
    torch.cuda.synchronize()
    got_fwd = lm.detach().permute(1, 0, 2).reshape(N_METRIC, L * F).cpu().numpy()
    ref_fwd = oracle.hashgrid_fwd(meta, x, p)
    np.testing.assert_allclose(got_fwd, ref_fwd, rtol=0, atol=2e-6)
    del got_fwd, ref_fwd
    ref_dp = oracle.hashgrid_bwd_params(meta, x, dout)
    got_dp = pd.grad.cpu().numpy()
    scale = np.abs(ref_dp).max()
    err = np.abs(got_dp - ref_dp)
    assert err.max() <= 2e-5 * scale, f"{name}/{dist}: grid gradient max err {err.max():.3e} vs scale {scale:.3e} at {err.argmax()}"
    # per level too: a coarse level's large sums must not hide a broken fine level
    for l in range(L):
        a, b = int(meta.offset[l]) * F, (int(meta.offset[l]) + int(meta.size[l])) * F
        sl = np.abs(ref_dp[a:b]).max()
        assert np.abs(got_dp[a:b] - ref_dp[a:b]).max() <= 2e-5 * max(sl, 1e-30), f"{name}/{dist}: level {l}"
    if D == 4:
        ref_dx = oracle.hashgrid_bwd_input(meta, x, p, dout)
        sx = np.abs(ref_dx).max()
        np.testing.assert_allclose(xd.grad.cpu().numpy(), ref_dx, rtol=0, atol=2e-5 * sx)


def test_hashgrid_sliced_is_run_to_run_stable(hip_lib, oracle):
    """Two launches of the owner-computes backward on the same inputs agree to the last few ulp (accumulation is in
    double inside the LDS; only the fp32 rounding of differently ordered double sums can differ)."""
    from emernerf_amd import _lib, ops
    meta, desc = _mk(oracle, "cfg2_static")
    L, F = meta.n_levels, meta.n_features
    dev = _dev()
    x = _positions("training", 3, 0).to(dev)
    g = torch.Generator().manual_seed(5)
    p = (torch.rand(meta.n_params, generator=g) - 0.5).to(dev)
    dlm = torch.randn(L, N_METRIC, F, generator=g).to(dev)
    _, mk = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
    outs = []
    for _ in range(2):
        grad = torch.empty(meta.n_params, device=dev)
        _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N_METRIC * F, ops._ptr(mk),
                  ops._ptr(grad), N_METRIC, ops._stream(x))
        outs.append(grad)
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs().max().item()
    assert d <= 1e-6 * outs[0].abs().max().item()


@pytest.mark.parametrize("name", ["cfg2_static", "flow_xyzt", "prop1"])
def test_hashgrid_sliced_level_ranges_partition_the_launch(hip_lib, oracle, name):
    """emer_hashgrid_bwd_params_sliced_levels over [k, L) and then [0, k) writes, into a buffer poisoned with NaN, the same table
    gradient as the single launch (every entry of a range written by its launch and by no other; the data-parallel trainer
    starts the collective of the first range between the two), for every split point, and the empty range is a no-op."""
    from emernerf_amd import _lib, ops
    meta, desc = _mk(oracle, name)
    D, L, F = meta.n_dims, meta.n_levels, meta.n_features
    dev = _dev()
    N = 1 << 17
    x = _positions("training", D, 0)[:N].contiguous().to(dev)
    g = torch.Generator().manual_seed(6)
    p = (torch.rand(meta.n_params, generator=g) - 0.5).to(dev)
    dlm = torch.randn(L, N, F, generator=g).to(dev)
    _, mk = ops.hashgrid_fwd_raw(desc, x, p, level_major=True, want_masks=True)
    st = ops._stream(x)
    one = torch.empty(meta.n_params, device=dev)
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk), ops._ptr(one), N, st)
    for k in sorted({1, L // 2, L - 1}):
        two = torch.full((meta.n_params,), float("nan"), device=dev)
        cut = int(meta.offset[k]) * F
        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk),
                  ops._ptr(two), N, k, L, st)
        torch.cuda.synchronize()
        assert torch.isnan(two[:cut]).all(), f"{name}: launch over levels [{k}, {L}) wrote below the level-{k} offset"
        assert not torch.isnan(two[cut:]).any()
        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk),
                  ops._ptr(two), N, k, k, st)   # empty range
        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), ops._ptr(x), ops._ptr(dlm), F, N * F, ops._ptr(mk),
                  ops._ptr(two), N, 0, k, st)
        torch.cuda.synchronize()
        assert not torch.isnan(two).any()
        d = (one - two).abs().max().item()
        assert d <= 1e-6 * one.abs().max().item(), f"{name}: split at level {k}: {d:.3e}"


@pytest.mark.parametrize("name", ["cfg2_static", "flow_xyzt", "default_static", "prop1"])
def test_hashgrid_sliced_add_accumulates(hip_lib, oracle, name):
    """emer_hashgrid_bwd_params_sliced_add (a table's second evaluation in a step) adds its gradient to what the buffer holds: after
    sliced(x1, d1) and sliced_add(x2, d2) the buffer equals the two single-launch gradients summed -- on the levels written with plain
    stores as on the ones merged with atomics (dense levels, the xyzt tables' half-size tail items)."""
    from emernerf_amd import _lib, ops
    meta, desc = _mk(oracle, name)
    D, L, F = meta.n_dims, meta.n_levels, meta.n_features
    dev = _dev()
    N1, N2 = 1 << 16, (1 << 15) + 77
    xs = _positions("training", D, 0)
    x1, x2 = xs[:N1].contiguous().to(dev), xs[N1:N1 + N2].contiguous().to(dev)
    g = torch.Generator().manual_seed(8)
    p = (torch.rand(meta.n_params, generator=g) - 0.5).to(dev)
    d1, d2 = torch.randn(L, N1, F, generator=g).to(dev), torch.randn(L, N2, F, generator=g).to(dev)
    _, m1 = ops.hashgrid_fwd_raw(desc, x1, p, level_major=True, want_masks=True)
    _, m2 = ops.hashgrid_fwd_raw(desc, x2, p, level_major=True, want_masks=True)
    st = ops._stream(x1)
    a, b = torch.empty(meta.n_params, device=dev), torch.empty(meta.n_params, device=dev)
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x1), ops._ptr(d1), F, N1 * F, ops._ptr(m1), ops._ptr(a), N1, st)
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x2), ops._ptr(d2), F, N2 * F, ops._ptr(m2), ops._ptr(b), N2, st)
    both = torch.full((meta.n_params,), float("nan"), device=dev)
    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), ops._ptr(x1), ops._ptr(d1), F, N1 * F, ops._ptr(m1), ops._ptr(both), N1, st)
    _lib.call("emer_hashgrid_bwd_params_sliced_add", ctypes.byref(desc), ops._ptr(x2), ops._ptr(d2), F, N2 * F, ops._ptr(m2), ops._ptr(both), N2, st)
    torch.cuda.synchronize()
    want = a.double() + b.double()
    assert float((both.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


# ------------------------------------------------------------------------------------------ fused heads
@pytest.mark.parametrize("rgbw", ["tile", "recompute", "streamed"])
def test_heads_metric_rows(hip_lib, monkeypatch, rgbw):
    """neck / rgb head forward, data gradients and weight gradients at 1 048 576 rows (8192 rays x 128 samples)
    against fp64 torch evaluated on the GPU, every row and every weight gradient compared.  ``rgbw``: the rgb head's layer-0 / 1
    weight gradients inside the backward kernel (the default; "recompute": with the hidden activations recomputed there instead of
    stored by the forward, emer_rgb_head_bwd_recompute) or as the round-3 streamed passes."""
    from emernerf_amd import fused
    monkeypatch.setattr(fused, "FUSED_RGB_WGRAD", rgbw != "streamed")
    monkeypatch.setattr(fused, "RGB_RECOMPUTE", 1 if rgbw == "recompute" else 0)
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    R, S, Kh, L, Fe = 8192, 128, 49, 16, 2
    N = R * S
    rnd = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(dev)  # noqa: E731
    enc = rnd(L, N, Fe)
    wn = [rnd(64, L * Fe, s=0.2), rnd(64, s=0.1), rnd(128, 64, s=0.1), rnd(128, s=0.1)]
    hray = rnd(R, Kh)
    wc = [rnd(64, Kh + 64, s=0.1), rnd(64, s=0.1), rnd(64, 64 + Kh + 64, s=0.1), rnd(64, s=0.1), rnd(3, 64, s=0.1), rnd(3, s=0.1)]
    gw = rnd(N, 3)
    gd = rnd(N, s=0.1)
    for t in wn + wc + [enc, hray]:
        t.requires_grad_(True)
    # ReLU hinges: with 3 x 64M hidden pre-activations some land within rounding distance of zero, and the branch fp32
    # takes there changes that row's gradients by O(1) of their size.  The fp64 reference therefore uses the product's
    # OWN relu masks (the post-ReLU activations its forward saved for the backward, captured through autograd's
    # saved-tensor hooks): same function, no hinge ambiguity, every row compared.
    saved = []

    def pack(t):
        saved.append(t)
        return t
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        geo, sem, dens = fused.neck(enc, *wn)
        rgb = fused.rgb_head(hray, geo, S, *wc)
    acts = [t for t in saved if t.shape == (N, 64) and t.data_ptr() != geo.data_ptr()]
    if rgbw == "recompute":
        # [r6] the recomputing rgb backward saves no hidden activation either (it recomputes them bitwise): its masks come from a forward
        # of the stored-activation path on the same inputs
        assert len(acts) == 0, [tuple(t.shape) for t in saved]
        saved.clear()
        monkeypatch.setattr(fused, "RGB_RECOMPUTE", 0)
        with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
            rgb_s = fused.rgb_head(hray, geo.detach().requires_grad_(True), S, *wc)
        monkeypatch.setattr(fused, "RGB_RECOMPUTE", 1)
        assert torch.equal(rgb_s, rgb), "both rgb head paths share the forward kernel"
        acts = [t for t in saved if t.shape == (N, 64) and t.data_ptr() != geo.data_ptr() and t.requires_grad is False and t.grad_fn is None][-2:]
        del rgb_s
    if len(acts) == 2:
        # [r5] the 128-output neck's backward recomputes its hidden layer too (bitwise the forward's values) and saves none: its mask
        # comes from a forward of the unfused path on the same inputs
        saved.clear()
        monkeypatch.setattr(fused, "FUSED_WGRAD", False)
        with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
            geo_u, _, _ = fused.neck(enc, *wn)
        monkeypatch.setattr(fused, "FUSED_WGRAD", True)
        assert torch.equal(geo_u, geo), "both neck paths share the forward kernel"
        h1 = [t for t in saved if t.shape == (N, 64) and t.data_ptr() != geo_u.data_ptr()]
        assert len(h1) == 1
        acts = h1 + acts
        del geo_u
    assert len(acts) == 3, [tuple(t.shape) for t in saved]  # h1 (neck), a1, a2 (rgb head)
    m_h1, m_a1, m_a2 = [(t > 0).double() for t in acts]
    ((rgb * gw).sum() + (dens * gd).sum()).backward()
    torch.cuda.synchronize()

    # fp64 reference on the GPU through plain torch ops (checker only; the product never calls these)
    e64 = enc.detach().double().permute(1, 0, 2).reshape(N, L * Fe).requires_grad_(True)
    wn64 = [t.detach().double().requires_grad_(True) for t in wn]
    wc64 = [t.detach().double().requires_grad_(True) for t in wc]
    h64 = hray.detach().double().requires_grad_(True)
    h1 = (e64 @ wn64[0].T + wn64[1]) * m_h1
    feats = h1 @ wn64[2].T + wn64[3]
    geo64 = feats[:, :64]
    dens64 = torch.exp(geo64[:, 0] - 1)
    hs = h64[:, None, :].expand(R, S, Kh).reshape(N, Kh)
    inp = torch.cat([hs, geo64], -1)
    a1 = (inp @ wc64[0].T + wc64[1]) * m_a1
    a2 = (torch.cat([a1, inp], -1) @ wc64[2].T + wc64[3]) * m_a2
    rgb64 = torch.sigmoid(a2 @ wc64[4].T + wc64[5])
    ((rgb64 * gw.double()).sum() + (dens64 * gd.double()).sum()).backward()

    def close(name, a, b, rtol=2e-4, sa=5e-5):
        a, b = a.detach().double(), b.detach().double()
        scale = b.abs().max().item()
        err = (a - b).abs()
        bad = err > (sa * scale + rtol * b.abs())
        assert not bad.any().item(), f"{name}: max err {err.max().item():.3e} (scale {scale:.3e}), {int(bad.sum())} outside"

    close("rgb", rgb, rgb64, rtol=1e-4, sa=2e-5)
    close("density", dens, dens64, rtol=1e-4, sa=2e-5)
    close("geo", geo, geo64, rtol=1e-4, sa=2e-5)
    close("denc", enc.grad.permute(1, 0, 2).reshape(N, L * Fe), e64.grad)
    close("dhray", hray.grad, h64.grad)
    for i, (a, b) in enumerate(zip(wn, wn64)):  # (the semantic half has no consumer here: exact zeros on both sides)
        close(f"neck dW{i}", a.grad, b.grad)
    for i, (a, b) in enumerate(zip(wc, wc64)):
        close(f"rgb dW{i}", a.grad, b.grad)


# ------------------------------------------------------------------------------------------ whole step
def _ref_from_trainer(oracle, tr):
    """oracle/ref_path.RefPath holding copies of the trainer's parameters (reference state_dict names)."""
    from oracle.train_parity import ref_from_trainer
    return ref_from_trainer(tr)


@pytest.mark.parametrize("kind", ["static", "dynamic", "flow", "feature"])
def test_full_step_gradients_vs_oracle(hip_lib, oracle, kind):
    """One optimizer step's gradients at 2048 rays x 128 samples (flow: 1024 x 128 -- seven xyzt evaluations per sample on
    the CPU oracle; proposal rounds 128 + 64, proposal nets training): every parameter's gradient in Trainer.flat.grads vs
    oracle/ref_path.py (torch-CPU autograd on the C oracle) with the stratified jitter -- and, for the flow model, the
    temporal-aggregation noise -- replayed on both sides (feature model: + semantic features, feature heads, learnable PE map,
    feature sky head, three cameras).  Same losses as Trainer.losses.  The flow case runs the batched
    xyzt evaluations (3N dynamic, N + 2N flow) with input gradients through both grids."""
    import torch.nn.functional as Fn
    from oracle.ref_path import pixel_step_loss, prop_loss
    from emernerf_amd.trainer import Trainer, synthetic_rays
    dev = _dev()
    R, S = (1024 if kind in ("flow", "feature") else 2048), 128
    tr = Trainer(kind=kind, device=dev, num_samples=S, prop_samples=(128, 64), table_init=0.3, seed=7)
    if kind in ("dynamic", "flow", "feature"):
        # With +-0.3 tables static + dynamic density saturates every ray (opacity 0.9999 .. 1, measured on the oracle), where the
        # sky term -log(1 - opacity) and its gradient 1 / (1 - opacity) are decided by the last bits of a sum.  The sky loss stays
        # ON: the density biases are lowered by 3 instead (opacity 0.3 .. 0.8 over these rays), so that the sky-BCE gradient
        # through a static + dynamic opacity is compared at scale.
        with torch.no_grad():
            tr.model.base_mlp[2].bias[0] -= 3.0
            tr.model.dynamic_base_mlp[2].bias[0] -= 3.0
    ref = _ref_from_trainer(oracle, tr)
    data = synthetic_rays(R, dev, seed=77, **(dict(num_cams=3, feature_dim=64) if kind == "feature" else {}))
    assert float(data["sky_masks"].sum()) > 0.1 * R
    cpu = {k: v.cpu() for k, v in data.items()}
    g = torch.Generator().manual_seed(9)
    jit = [torch.rand(R, generator=g) for _ in range(3)]
    it = iter([j.to(dev) for j in jit])
    tr.estimator.jitter_fn = lambda n, d: next(it)
    noise = torch.rand(R, S, 1, generator=g)
    tr.model._noise = lambda like: noise.to(like.device)
    loss_hip = tr._forward_backward(data, prop_grad=True)
    torch.cuda.synchronize()

    res = ref.render_rays(cpu, S, [128, 64], jitters=jit, requires_grad=True, noise_fn=lambda like: noise)
    if kind != "static":
        assert float(res["opacity"].max()) < 0.9999, "rays must not saturate (the sky term is part of this comparison)"
    pl = prop_loss(ref.cache, res["extras"]["trans"], 1024.0)
    pl.backward()
    loss = pixel_step_loss(res, cpu)   # the reference's expressions (train_emernerf.py:655-716), oracle/ref_path.py
    (loss * 1024.0).backward()
    np.testing.assert_allclose(float(loss_hip), float(loss), rtol=1e-4)

    checked, worst = 0, (0.0, "")
    gmax = max(float(v.grad.abs().max()) for v in ref.t.values() if v.grad is not None and v.numel() < 100000)  # (MLP weights)
    for prefix, mod in [("model/", tr.model)] + [(f"prop{i}/", p) for i, p in enumerate(tr.props)]:
        for k, q in mod.named_parameters():
            want = ref.t[prefix + k].grad
            got = q.grad.detach().cpu()
            if want is None:  # never reached by the graph (e.g. proposal net 0: the reference's late-binding lambda)
                assert float(got.abs().max()) == 0.0, f"{prefix + k}: gradient where the reference has none"
                continue
            scale = float(want.abs().max())
            err = float((got - want).abs().max())
            # 4e-4 of the parameter's own gradient scale (+ 1e-3 of the step's largest MLP gradient: parameters whose whole gradient
            # is that small are compared on that floor); measured on MI355X: 4e-5 .. 8e-5 over the four models (hinge flips of a
            # few pre-activations within an ulp of zero among 10^5 .. 10^6 rows)
            assert err <= 4e-4 * (scale + 1e-3 * gmax), f"{prefix + k}: max err {err:.3e} vs scale {scale:.3e}"
            worst = max(worst, (err / (scale + 1e-3 * gmax), prefix + k))
            checked += 1
    print(f"\n[{kind}] worst gradient error / (own scale + 1e-3 gmax): {worst[0]:.2e} at {worst[1]}")
    assert checked >= 15


# ------------------------------------------------------------------------------- INTEGRATION.md section 2 shims
def test_nerfacc_compat_against_oracle(hip_lib, oracle):
    """The reference-side binding points (emernerf_amd.nerfacc_compat: what INTEGRATION.md section 2 swaps in for
    `import nerfacc`) called directly, the way the reference calls them (render_utils.py:35-43,73-77,103-105;
    nerfacc_prop_net.py:153,165-172): values vs the C oracle, and the gradient of the reference's own
    ``weights = trans * alphas`` formulation vs fp64 autograd."""
    from emernerf_amd import nerfacc_compat as NC
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    R, S = 513, 128
    e = torch.sort(torch.rand(R, S + 1, generator=g) * 40, -1).values
    ts, te = e[:, :-1].contiguous(), e[:, 1:].contiguous()
    sg = torch.rand(R, S, generator=g) * 0.5
    rw, rT, ra = oracle.render_weights(ts, te, sg)
    sgd = sg.to(dev).requires_grad_(True)
    trans, alphas = NC.render_transmittance_from_density(ts.to(dev), te.to(dev), sgd)
    np.testing.assert_allclose(trans.detach().cpu().numpy(), rT, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(alphas.detach().cpu().numpy(), ra, rtol=2e-5, atol=1e-7)
    weights = trans * alphas  # render_utils.py:73-77
    vals = torch.rand(R, S, 3, generator=g)
    rgb = NC.accumulate_along_rays(weights, values=vals.to(dev))
    opa = NC.accumulate_along_rays(weights, values=None)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), oracle.accumulate(rw, vals), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(opa.detach().cpu().numpy(), oracle.accumulate(rw), rtol=1e-4, atol=1e-6)
    gr, go = torch.randn(R, 3, generator=g), torch.randn(R, 1, generator=g)
    ((rgb * gr.to(dev)).sum() + (opa * go.to(dev)).sum()).backward()
    s64 = sg.double().requires_grad_(True)
    sdt = s64 * (te - ts).double()
    cum = torch.cumsum(sdt, -1)
    T64 = torch.exp(-torch.cat([torch.zeros_like(cum[:, :1]), cum[:, :-1]], -1))
    w64 = T64 * (1 - torch.exp(-sdt))
    (((w64[..., None] * vals.double()).sum(1) * gr.double()).sum() + (w64.sum(1, keepdim=True) * go.double()).sum()).backward()
    got, want = sgd.grad.cpu().double(), s64.grad
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max()), "d(trans*alphas)/dsigma through the shim"
    w2, T2, a2 = NC.render_weight_from_density(ts.to(dev), te.to(dev), sg.to(dev))
    np.testing.assert_allclose(w2.cpu().numpy(), rw, rtol=2e-5, atol=1e-7)

    # importance_sampling / searchsorted (nerfacc_prop_net.py:148-175,349-359)
    m, n = 129, 64
    w = torch.rand(R, m - 1, generator=g) ** 4 + 1e-3
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(w, -1)], -1)
    cdf = cdf / cdf[:, -1:]
    vals_s = torch.sort(torch.rand(R, m, generator=g), -1).values
    jit = torch.rand(R, generator=g)
    iv, smp = NC.importance_sampling(NC.RayIntervals(vals=vals_s.to(dev)), cdf.to(dev), n, stratified=True, jitter=jit.to(dev))
    ref_s = oracle.importance_sample(vals_s, cdf, n, jit)
    assert np.array_equal(iv.vals.cpu().numpy().view(np.uint32), ref_s.view(np.uint32)), "sample offsets must be bit-exact"
    np.testing.assert_allclose(smp.vals.cpu().numpy(), (ref_s[:, :-1] + ref_s[:, 1:]) * 0.5, rtol=1e-6)
    iv0, _ = NC.importance_sampling(NC.RayIntervals(vals=vals_s.to(dev)), cdf.to(dev), n, stratified=False)
    assert np.array_equal(iv0.vals.cpu().numpy().view(np.uint32), oracle.importance_sample(vals_s, cdf, n, None).view(np.uint32))
    il, ir = NC.searchsorted(NC.RayIntervals(vals=vals_s.to(dev)), iv)
    q, key = iv.vals.cpu().numpy(), vals_s.numpy()
    want_r = np.stack([np.searchsorted(key[r], q[r], side="right") for r in range(R)])
    assert np.array_equal(ir.cpu().numpy(), np.clip(want_r, 0, m - 1))
    assert np.array_equal(il.cpu().numpy(), np.clip(want_r - 1, 0, m - 1))


@pytest.mark.parametrize("typ", ["uniform", "lindisp", "uniform_lindisp", "sqrt", "log", "uniform_lindisp_0"])
def test_stot_all_transforms(hip_lib, oracle, typ):
    """Every entry of the reference's TRANSFROM_DICT (nerfacc_prop_net.py:298-314): bit-exact vs the oracle except
    "log" (libm exp/log: 4 ulp)."""
    from emernerf_amd import ops
    s = torch.rand(4096, generator=torch.Generator().manual_seed(8))
    got = ops.stot(s.to(_dev()), 0.5, 300.0, typ).cpu().numpy()
    want = oracle.stot(s, 0.5, 300.0, typ)
    if typ == "log":
        np.testing.assert_allclose(got, want, rtol=5e-7)
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------------------- half-precision tables (BASELINE configs[1])
def test_fp16_table_mode_metric_shape(hip_lib, oracle):
    """tcnn's half-precision tables as BASELINE.md 2.2 states them: the fp32 MASTER is cast to fp16 per call, the encode gathers
    fp16 entries and accumulates in fp32, the gradient is accumulated in fp32 by the owner-computes backward straight into the
    master's .grad.  At N = 1 048 576 on the cfg-2 grid: forward vs the oracle on the fp16-ROUNDED table (2e-6 abs, the fp32
    tolerance), table gradient vs the oracle (it does not depend on the table: 2e-5 x max)."""
    from emernerf_amd import ops
    meta, desc = _mk(oracle, "cfg2_static")
    D, L, F = meta.n_dims, meta.n_levels, meta.n_features
    x = _positions("training", D, seed=31)
    g = torch.Generator().manual_seed(32)
    p = torch.rand(meta.n_params, generator=g) - 0.5
    dout = torch.randn(N_METRIC, L * F, generator=g)
    dev = _dev()
    pd = p.to(dev).requires_grad_(True)
    lm = ops.hashgrid_encode_lm(x.to(dev), pd, desc, table_dtype=torch.float16)
    assert lm.dtype == torch.float32
    dlm = dout.view(N_METRIC, L, F).permute(1, 0, 2).contiguous().to(dev)
    lm.backward(dlm)
    torch.cuda.synchronize()
    got = lm.detach().permute(1, 0, 2).reshape(N_METRIC, L * F).cpu().numpy()
    ref = oracle.hashgrid_fwd(meta, x, p.half().float())
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    assert pd.grad.dtype == torch.float32
    ref_dp = oracle.hashgrid_bwd_params(meta, x, dout)
    assert np.abs(pd.grad.cpu().numpy() - ref_dp).max() <= 2e-5 * np.abs(ref_dp).max()


def test_fp16_table_trainer_step(hip_lib):
    """Trainer(table_dtype="f16"): every encoder reads fp16 copies, the flat fp32 buffers stay the masters; the step trains
    (finite, decreasing loss) and its first-step gradients agree with the fp32-table trainer to fp16 table rounding."""
    from emernerf_amd.trainer import Trainer, synthetic_rays
    dev = _dev()
    data = synthetic_rays(1024, dev, seed=5)
    jit = torch.full((1024,), 0.41, device=dev)
    gs = {}
    for td in ("f32", "f16"):
        tr = Trainer(kind="static", device=dev, num_samples=64, prop_samples=(64, 32), table_init=0.3, seed=2, table_dtype=td)
        tr.estimator.jitter_fn = lambda n, d: jit
        tr._forward_backward(data, prop_grad=False)
        tr._exchange_grads(False)
        a, b = tr.flat.ranges["main"]
        gs[td] = tr.flat.grads[a:b].clone()
        if td == "f16":
            assert tr.flat.params.dtype == torch.float32
            l0 = float(tr.train_step(data)["loss"])
            for _ in range(6):
                l1 = float(tr.train_step(data)["loss"])
            assert np.isfinite([l0, l1]).all() and l1 < l0
    rel = float((gs["f16"] - gs["f32"]).abs().max() / gs["f32"].abs().max())
    assert rel < 2e-2, rel   # tables differ by fp16 rounding (2^-11 relative): gradients follow to that order
