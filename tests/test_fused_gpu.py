"""Fused heads (register-resident emer_neck_* / emer_rgb_head_*, generic emer_mlp_chain, emer_wgrad_segmented) vs fp64
torch references of the reference's heads."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(name, got, want, rtol=1e-4, scale_atol=2e-5):
    got, want = got.detach().double().cpu().numpy(), want.detach().double().cpu().numpy()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=scale_atol * max(np.abs(want).max(), 1e-30), err_msg=name)


@pytest.mark.parametrize("L,Fe,NG,N", [(16, 2, 64, 1000), (10, 4, 128, 777), (4, 2, 64, 16), (8, 1, 64, 33)])
def test_base_mlp(hip_lib, L, Fe, NG, N):
    from emernerf_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(L + NG + N)
    K0, H = L * Fe, 64
    enc = torch.randn(L, N, Fe, generator=g)
    W0, b0 = torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g) * 0.1
    W1, b1 = torch.randn(NG, H, generator=g) / H ** 0.5, torch.randn(NG, generator=g) * 0.1
    t = [v.to(dev).requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    feats, dens = fused.base_mlp(*t)
    r = [v.double().requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    x = r[0].permute(1, 0, 2).reshape(N, K0)  # row-major view of the level-major encoding
    f_ref = F.linear(torch.relu(F.linear(x, r[1], r[2])), r[3], r[4])
    d_ref = torch.exp(f_ref[:, 0] - 1)
    _close("feats", feats, f_ref); _close("density", dens, d_ref)
    gf, gd = torch.randn(N, NG, generator=g), torch.randn(N, generator=g)
    (feats * gf.to(dev)).sum().add((dens * gd.to(dev)).sum()).backward()
    ((f_ref * gf.double()).sum() + (d_ref * gd.double()).sum()).backward()
    for name, a, b in zip(("denc", "dW0", "db0", "dW1", "db1"), t, r):
        _close(name, a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)
    # only one of the two outputs carries gradient
    t2 = [v.detach().clone().requires_grad_(True) for v in t]
    _, dens2 = fused.base_mlp(*t2)
    dens2.sum().backward()
    assert t2[1].grad is not None and torch.isfinite(t2[0].grad).all()


@pytest.mark.parametrize("L,Fe,NG,N,which", [(16, 2, 128, 1000, "all"), (16, 2, 128, 2048, "geo"), (10, 4, 128, 777, "all"),
                                              (4, 2, 64, 16, "all"), (8, 1, 64, 33, "dens"), (3, 8, 64, 100, "all"),
                                              (16, 4, 128, 50, "sem"), (10, 4, 128, 20000, "all"), (16, 2, 128, 9000, "sem")])
@pytest.mark.parametrize("fusedw", [True, False])
def test_neck_register_resident(hip_lib, monkeypatch, L, Fe, NG, N, which, fusedw):
    """emer_neck_fwd / emer_neck_bwd (+ emer_wgrad_segmented) and emer_neck_bwd_fused (weight gradients accumulated inside
    the backward kernel): split outputs, ragged row counts, every combination of live output gradients."""
    from emernerf_amd import fused
    monkeypatch.setattr(fused, "FUSED_WGRAD", fusedw)
    dev = torch.device("cuda:0")
    assert fused.neck_supported(L, Fe, 64, NG)
    g = torch.Generator().manual_seed(L * 7 + NG + N)
    K0, H = L * Fe, 64
    enc = torch.randn(L, N, Fe, generator=g)
    W0, b0 = torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g) * 0.1
    W1, b1 = torch.randn(NG, H, generator=g) / H ** 0.5, torch.randn(NG, generator=g) * 0.1
    t = [v.to(dev).requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    geo, sem, dens = fused.neck(*t)
    r = [v.double().requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    x = r[0].permute(1, 0, 2).reshape(N, K0)
    f_ref = F.linear(torch.relu(F.linear(x, r[1], r[2])), r[3], r[4])
    d_ref = torch.exp(f_ref[:, 0] - 1)
    _close("geo", geo, f_ref[:, :64]); _close("density", dens, d_ref)
    assert (sem is None) == (NG == 64)
    if sem is not None:
        _close("sem", sem, f_ref[:, 64:])
    gg, gs, gd = torch.randn(N, 64, generator=g), torch.randn(N, 64, generator=g), torch.randn(N, generator=g)
    loss, loss_ref = 0.0, 0.0
    if which in ("all", "geo"):
        loss, loss_ref = loss + (geo * gg.to(dev)).sum(), loss_ref + (f_ref[:, :64] * gg.double()).sum()
    if which in ("all", "sem") and sem is not None:
        loss, loss_ref = loss + (sem * gs.to(dev)).sum(), loss_ref + (f_ref[:, 64:] * gs.double()).sum()
    if which in ("all", "dens"):
        loss, loss_ref = loss + (dens * gd.to(dev)).sum(), loss_ref + (d_ref * gd.double()).sum()
    loss.backward(); loss_ref.backward()
    for name, a, b in zip(("denc", "dW0", "db0", "dW1", "db1"), t, r):
        _close(name, a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)


@pytest.mark.parametrize("fusedw", [True, False])  # weight gradients inside the backward kernel (L F <= 16) / separate passes
@pytest.mark.parametrize("L,N,Fe", [(8, 1000, 1), (8, 17, 1), (4, 256, 1), (8, 70000, 1), (4, 333, 4), (7, 100, 2), (12, 50, 2)])
def test_density_mlp(hip_lib, monkeypatch, L, N, Fe, fusedw):
    from emernerf_amd import fused
    monkeypatch.setattr(fused, "FUSED_WGRAD", fusedw)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(L + N)
    enc = torch.randn(L, N, Fe, generator=g)
    L_rows, L = L, L * Fe  # K0 = L * F input features
    W0, b0 = torch.randn(64, L, generator=g) / L ** 0.5, torch.randn(64, generator=g) * 0.1
    W1, b1 = torch.randn(1, 64, generator=g) / 8, torch.randn(1, generator=g) * 0.1
    t = [v.to(dev).requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    dens = fused.density_mlp(*t)
    r = [v.double().requires_grad_(True) for v in (enc, W0, b0, W1, b1)]
    d_ref = torch.exp(F.linear(torch.relu(F.linear(r[0].permute(1, 0, 2).reshape(N, L), r[1], r[2])), r[3], r[4])[:, 0] - 1)
    _close("density", dens, d_ref)
    gd = torch.randn(N, generator=g)
    (dens * gd.to(dev)).sum().backward()
    (d_ref * gd.double()).sum().backward()
    for name, a, b in zip(("denc", "dW0", "db0", "dW1", "db1"), t, r):
        _close(name, a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)
    with torch.no_grad():  # inference path: no activations saved
        _close("density_nograd", fused.density_mlp(*[v.detach() for v in t]), d_ref)


# weight gradients: all inside the backward kernel, only the output layer's
# inside it (round 3: layers 0 / 1 streamed), or every one as a separate pass
# [r6] "all" / "all_a2": with the hidden activations (both / a2 only) RECOMPUTED in that kernel; "all_stored": read back from the forward's stores (the default)
@pytest.mark.parametrize("fusedw", ["all", "all_a2", "all_stored", "w2_only", False])
@pytest.mark.parametrize("R,S,Kh,NG,ld", [(16, 64, 49, 64, 64), (5, 16, 49, 64, 128), (3, 128, 33, 64, 64), (1, 16, 49, 64, 64), (5000, 16, 49, 64, 64),
                                          (1031, 32, 49, 64, 64), (2, 96, 17, 64, 192)])
def test_rgb_head(hip_lib, monkeypatch, R, S, Kh, NG, ld, fusedw):
    from emernerf_amd import fused
    monkeypatch.setattr(fused, "FUSED_WGRAD", bool(fusedw))
    monkeypatch.setattr(fused, "FUSED_RGB_WGRAD", fusedw in ("all", "all_a2", "all_stored"))
    monkeypatch.setattr(fused, "RGB_RECOMPUTE", {"all": 1, "all_a2": 2}.get(fusedw, 0))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R * S + Kh)
    N, H, K0 = R * S, 64, Kh + NG
    hray = torch.randn(R, Kh, generator=g)
    feats = torch.randn(N, ld, generator=g)  # geo is the first NG columns of a wider feature tensor when ld > NG
    Ws = [torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g) * 0.1,
          torch.randn(H, H + K0, generator=g) / (H + K0) ** 0.5, torch.randn(H, generator=g) * 0.1,
          torch.randn(3, H, generator=g) / H ** 0.5, torch.randn(3, generator=g) * 0.1]
    hd = hray.to(dev).requires_grad_(True)
    fd = feats.to(dev).requires_grad_(True)
    wd = [w.to(dev).requires_grad_(True) for w in Ws]
    rgb = fused.rgb_head(hd, fd[:, :NG], S, *wd)
    h64, f64 = hray.double().requires_grad_(True), feats.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in Ws]
    inp = torch.cat([h64.repeat_interleave(S, dim=0), f64[:, :NG]], -1)  # [dir-PE | emb | geo], radiance_field.py:644
    x = torch.relu(F.linear(inp, w64[0], w64[1]))
    x = torch.relu(F.linear(torch.cat([x, inp], -1), w64[2], w64[3]))   # mlp.py:41-42 skip connection
    ref = torch.sigmoid(F.linear(x, w64[4], w64[5]))
    _close("rgb", rgb, ref)
    go = torch.randn(N, 3, generator=g)
    (rgb * go.to(dev)).sum().backward()
    (ref * go.double()).sum().backward()
    _close("dhray", hd.grad, h64.grad, rtol=2e-4, scale_atol=5e-5)
    _close("dfeats", fd.grad, f64.grad, rtol=2e-4, scale_atol=5e-5)
    for i, (a, b) in enumerate(zip(wd, w64)):
        _close(f"dW{i}", a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)


@pytest.mark.parametrize("R,S,Kh,ld", [(16, 64, 49, 64), (5, 16, 49, 128), (1031, 32, 49, 64), (4100, 128, 49, 64), (2, 96, 17, 192), (1, 16, 49, 64)])
def test_rgb_head_recomputed_activations_are_bitwise_the_stored_ones(hip_lib, monkeypatch, R, S, Kh, ld):
    """[r6] emer_rgb_head_bwd_recompute (a1 / a2 recomputed from geo and the per-ray pre-activations with the forward's fragments in the
    forward's order) against emer_rgb_head_bwd_fused on the forward's stored activations: every gradient bit for bit, with and without the
    forward riding along in the neck's launch being irrelevant here (plain rgb_head calls).  4100 x 128: more rays than resident waves
    (several rays per wave, the per-ray pre-activation buffers alternate), 1 x 16: one tile."""
    from emernerf_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R + S)
    N, H, NG, K0 = R * S, 64, 64, Kh + 64
    hray = torch.randn(R, max(Kh, 1), generator=g)[:, :Kh].contiguous().to(dev)
    feats = torch.randn(N, ld, generator=g).to(dev)
    Ws = [torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g) * 0.1, torch.randn(H, H + K0, generator=g) / (H + K0) ** 0.5,
          torch.randn(H, generator=g) * 0.1, torch.randn(3, H, generator=g) / H ** 0.5, torch.randn(3, generator=g) * 0.1]
    go = torch.randn(N, 3, generator=g).to(dev)
    res = {}
    for mode in (1, 2, 0):   # both recomputed / a2 only / both stored
        monkeypatch.setattr(fused, "RGB_RECOMPUTE", mode)
        hd, fd = hray.clone().requires_grad_(True), feats.clone().requires_grad_(True)
        wd = [w.to(dev).requires_grad_(True) for w in Ws]
        rgb = fused.rgb_head(hd, fd[:, :NG], S, *wd)
        assert rgb.grad_fn is not None
        (rgb * go).sum().backward()
        w0g, b0g, w1g, b1g, w2g, b2g = [w.grad for w in wd]
        # bit for bit: the colours, dgeo and (through the per-ray sums s0 / s1 and emer_ray_pre_bwd) dhray.  The weight gradients leave the
        # kernel as per-workgroup partials that are bitwise equal too, but their final reduction (linear_dw_reduce_multi over ranges of
        # workgroups, emer_ray_wgrad over chunks of rays) meets in float atomics above ~8 workgroups / 256 rays: order-dependent in the
        # last bits from run to run of EITHER path, so they are compared to 1e-5 of their scale
        res[mode] = {"exact": [rgb.detach(), hd.grad, fd.grad],
                     "close": [w0g, b0g, w1g, b1g, w2g, b2g]}
    for mode in (1, 2):
        for i, (a, b) in enumerate(zip(res[mode]["exact"], res[0]["exact"])):
            assert torch.equal(a, b), f"mode {mode}, tensor {i}: recomputed != stored (max |diff| {float((a - b).abs().max()):.3e})"
        for i, (a, b) in enumerate(zip(res[mode]["close"], res[0]["close"])):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30, f"mode {mode}, weight gradient {i}"
            if R <= 16:   # one workgroup range, one chunk of rays: nothing meets in an atomic
                assert torch.equal(a, b), f"mode {mode}, weight gradient {i}: recomputed != stored"


def test_grad_sinks_equal_autograd_accumulation(hip_lib):
    """fused.grad_sinks(): weight gradients accumulated straight into an existing .grad == what autograd's
    AccumulateGrad produces (old .grad + dW), for the neck, the proposal density MLP and the rgb head."""
    from emernerf_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    R, S, Kh, L, Fe = 6, 32, 43, 16, 2
    N = R * S
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(dev)

    def params():
        gg = torch.Generator().manual_seed(11)
        r = lambda *shape, s=1.0: torch.nn.Parameter((torch.randn(*shape, generator=gg) * s).to(dev))
        return {"n": [r(64, L * Fe, s=0.2), r(64, s=0.1), r(128, 64, s=0.1), r(128, s=0.1)],
                "d": [r(64, 8, s=0.3), r(64, s=0.1), r(1, 64, s=0.1), r(1, s=0.1)],
                "c": [r(64, Kh + 64, s=0.1), r(64, s=0.1), r(64, 64 + Kh + 64, s=0.1), r(64, s=0.1), r(3, 64, s=0.1), r(3, s=0.1)]}

    enc, enc1, hray, geo = mk(L, N, Fe), mk(8, N, 1), mk(R, Kh), mk(N, 64)
    w = [mk(N, 64), mk(N), mk(N), mk(N, 3)]

    def run(P):
        geo_o, sem_o, dens = fused.neck(enc, *P["n"])
        d2 = fused.density_mlp(enc1, *P["d"])
        rgb = fused.rgb_head(hray, geo, S, *P["c"])
        ((geo_o * w[0]).sum() + (dens * w[1]).sum() + (d2 * w[2]).sum() + (rgb * w[3]).sum()).backward()

    ref = params()
    run(ref)  # plain autograd: .grad created by AccumulateGrad
    got = params()
    pre = {}
    for k, ps in got.items():
        for i, p in enumerate(ps):
            p.grad = torch.full_like(p, 0.25)  # a pre-existing gradient the sinks must ADD to
            pre[(k, i)] = p.grad.data_ptr()
    with fused.grad_sinks(True):
        run(got)
        fused.join_side_stream()  # a Trainer created earlier in this process may have enabled the wgrad side stream
    assert not fused._USE_GRAD_SINKS, "the sink opt-in must not leak out of its scope"
    for k in ref:
        for i, (a, b) in enumerate(zip(got[k], ref[k])):
            assert a.grad.data_ptr() == pre[(k, i)], "gradient was not accumulated in place"
            _close(f"{k}{i}", a.grad - 0.25, b.grad, rtol=2e-4, scale_atol=5e-5)
    # the same through the weight-gradient side stream
    got2 = params()
    for ps in got2.values():
        for p in ps:
            p.grad = torch.full_like(p, 0.25)
    old_side, fused.SIDE_STREAM = fused.SIDE_STREAM, torch.cuda.Stream()
    try:
        with fused.grad_sinks(True):
            run(got2)
            fused.join_side_stream()
    finally:
        fused.SIDE_STREAM = old_side
    for k in ref:
        for i, (a, b) in enumerate(zip(got2[k], ref[k])):
            _close(f"side {k}{i}", a.grad - 0.25, b.grad, rtol=2e-4, scale_atol=5e-5)


@pytest.mark.parametrize("N,K0,act", [(8192, 43, "sigmoid"), (100, 27, "sigmoid"), (33, 43, "none"), (64, 70, "sigmoid")])  # K0 = 70: the general chain path
def test_skip_mlp3(hip_lib, N, K0, act):
    """fused.skip_mlp3 (per-ray sky head) vs fp64 torch: MLP(3 layers, skip at 1) + sigmoid."""
    from emernerf_amd import fused, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + K0 + 7)  # (seed N + K0 has a hidden pre-activation ~1e-8 from zero in row 7275)
    H = 64
    vals = [torch.randn(N, K0, generator=g), torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g) * 0.1,
            torch.randn(H, H + K0, generator=g) / (H + K0) ** 0.5, torch.randn(H, generator=g) * 0.1,
            torch.randn(3, H, generator=g) / 8, torch.randn(3, generator=g) * 0.1]
    t = [v.to(dev).requires_grad_(True) for v in vals]
    r = [v.double().requires_grad_(True) for v in vals]
    out = fused.skip_mlp3(*t, final_act=_lib.ACT_SIGMOID if act == "sigmoid" else _lib.ACT_NONE)
    x = r[0]
    h = torch.relu(F.linear(x, r[1], r[2]))
    h = torch.relu(F.linear(torch.cat([h, x], -1), r[3], r[4]))
    ref = F.linear(h, r[5], r[6])
    if act == "sigmoid":
        ref = torch.sigmoid(ref)
    _close("out", out, ref)
    w = torch.randn(N, 3, generator=g)
    (out * w.to(dev)).sum().backward(); (ref * w.double()).sum().backward()
    # dx is checked row by row: a hidden pre-activation within ~1e-7 of zero may take the other ReLU branch in fp32
    # than in the fp64 reference (about one row in 10^4 with these sizes), which legitimately changes that row
    got, want = t[0].grad.double().cpu(), r[0].grad
    bad_rows = int(((got - want).abs() > 2e-4 * want.abs() + 5e-5 * float(want.abs().max())).any(1).sum())
    assert bad_rows <= max(1, N // 4096), f"dx: {bad_rows} rows differ"
    for name, a, b in list(zip(("dx", "dW0", "db0", "dW1", "db1", "dW2", "db2"), t, r))[1:]:
        _close(name, a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)


@pytest.fixture(params=[True, False], ids=["fusedw", "streamedw"])
def rmlp_wgrad_mode(request):
    """Both backward paths of the plain heads: weight gradients inside the backward kernel (emer_rmlp_bwd_fused, the default for
    stacks with <= 16 outputs) and emer_rmlp_bwd + streamed weight gradients."""
    from emernerf_amd import fused
    prev, fused.FUSED_RMLP_WGRAD = fused.FUSED_RMLP_WGRAD, request.param
    yield request.param
    fused.FUSED_RMLP_WGRAD = prev


@pytest.mark.parametrize("dims,sig,N", [((64, 64, 1), True, 1000), ((40, 64, 64, 6), False, 777), ((64, 64, 64, 64), False, 4096),
                                        ((43, 32, 5), False, 17), ((64, 64, 1), True, 10000), ((64, 64, 64, 3), True, 5000), ((32, 64, 16), False, 3001)])
def test_seq_mlp(hip_lib, dims, sig, N, rmlp_wgrad_mode):
    """fused.seq_mlp (shadow / flow / dino heads) vs fp64 torch, including dx."""
    from emernerf_amd import fused, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(dims) + N + 3)
    x = torch.randn(N, dims[0], generator=g)
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    Bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    assert fused.seq_mlp_supported(Ws)
    t = [v.to(dev).requires_grad_(True) for v in [x] + Ws + Bs]
    r = [v.double().requires_grad_(True) for v in [x] + Ws + Bs]
    n = len(Ws)
    out = fused.seq_mlp(t[0], t[1:1 + n], t[1 + n:], _lib.ACT_SIGMOID if sig else _lib.ACT_NONE)
    h = r[0]
    for i in range(n):
        h = F.linear(h, r[1 + i], r[1 + n + i])
        if i + 1 < n:
            h = torch.relu(h)
    ref = torch.sigmoid(h) if sig else h
    _close("out", out, ref)
    w = torch.randn(N, dims[-1], generator=g)
    (out * w.to(dev)).sum().backward(); (ref * w.double()).sum().backward()
    for i, (a, b) in enumerate(zip(t, r)):
        _close(f"grad{i}", a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)


@pytest.mark.parametrize("L,F,dims,N", [(10, 4, (64, 64, 6), 1000), (10, 4, (64, 6), 333), (4, 4, (64, 64, 64), 50), (16, 4, (64, 64, 3), 4099),
                                         (10, 4, (64, 64, 6), 262144)])
def test_seq_mlp_level_major(hip_lib, L, F, dims, N, rmlp_wgrad_mode):
    """fused.seq_mlp_lm (flow MLP on the level-major xyzt encoding, radiance_field.py:359-389) vs fp64 torch on the
    row-major view of the same encoding, including the level-major input gradient."""
    from emernerf_amd import fused, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(L * 100 + N)
    K0 = L * F
    enc = torch.randn(L, N, F, generator=g)
    widths = (K0,) + dims
    Ws = [torch.randn(widths[i + 1], widths[i], generator=g) / widths[i] ** 0.5 for i in range(len(dims))]
    Bs = [torch.randn(widths[i + 1], generator=g) * 0.1 for i in range(len(dims))]
    assert fused.rmlp_supported(Ws, K0, F)
    t = [v.to(dev).requires_grad_(True) for v in [enc] + Ws + Bs]
    r = [v.double().requires_grad_(True) for v in [enc] + Ws + Bs]
    n = len(Ws)
    # The reference takes the PRODUCT's ReLU masks (its saved post-ReLU activations): a pre-activation within fp32 rounding of
    # zero may legitimately get either sign, and one flipped unit moves a whole row of dW by O(1) -- at 262144 rows a handful
    # of such units exist for any pair of correct fp32 evaluations (same device as tests/test_a_metric_shape_gpu.py).
    saved = []

    def pack(x):
        saved.append(x)
        return x
    # (the fused backward recomputes the hidden layers -- bitwise the forward's values -- and saves none: the masks then come from a
    # forward of the other path on the same inputs)
    mode, fused.FUSED_RMLP_WGRAD = fused.FUSED_RMLP_WGRAD, False
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda x: x):
        out = fused.seq_mlp_lm(t[0], t[1:1 + n], t[1 + n:])
    fused.FUSED_RMLP_WGRAD = mode
    if mode and fused.rmlp_bwd_fused_supported(Ws, K0, F, N):
        out_f = fused.seq_mlp_lm(t[0], t[1:1 + n], t[1 + n:])
        assert torch.equal(out_f, out), "the two paths share the forward kernel"
        out = out_f
    acts = [x for x in saved if x.dim() == 2 and tuple(x.shape) == (N, 64) and x.data_ptr() != out.data_ptr()]
    assert len(acts) == n - 1, [tuple(x.shape) for x in saved]
    h = r[0].permute(1, 0, 2).reshape(N, K0)
    for i in range(n):
        h = torch.nn.functional.linear(h, r[1 + i], r[1 + n + i])
        if i + 1 < n:
            mask = (acts[i] > 0).double().cpu()
            flips = int((mask != (h.detach() > 0).double()).sum())
            assert flips <= 2 + N // 4000, flips      # masks agree except at the kink
            h = h * mask
    _close("out", out, h)
    w = torch.randn(N, dims[-1], generator=g)
    (out * w.to(dev)).sum().backward(); (h * w.double()).sum().backward()
    for i, (a, b) in enumerate(zip(t, r)):
        _close(f"grad{i}", a.grad, b.grad, rtol=2e-4, scale_atol=5e-5)


def test_seq_mlp_routes_to_register_resident_kernels(hip_lib):
    """The shipped head shapes must not fall back to the LDS-staged chain."""
    from emernerf_amd import fused
    mk = lambda *d: [torch.empty(d[i + 1], d[i]) for i in range(len(d) - 1)]
    assert fused.rmlp_supported(mk(64, 64, 1), 64, 0)          # shadow head
    assert fused.rmlp_supported(mk(64, 64, 64, 64), 64, 0)     # feature heads
    assert fused.rmlp_supported(mk(40, 64, 64, 6), 40, 4)      # flow MLP on the xyzt grid (L10 x F4)
    assert not fused.rmlp_supported(mk(43, 32, 5), 43, 0)


def test_forwards_outside_autograd_recording_match(hip_lib):
    """Under torch.no_grad() the heads detach their inputs and skip what only a backward needs (hidden activations, the grid
    backward's bitmaps); the values must be bit-identical to the recorded forward."""
    from emernerf_amd import fused, ops, _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    R, S, Kh, H = 64, 32, 49, 64
    hr, geo = torch.randn(R, Kh, generator=g).to(dev), torch.randn(R * S, H, generator=g).to(dev)
    ws = [torch.randn(H, Kh + H, generator=g) / 10, torch.randn(H, generator=g) / 10, torch.randn(H, H + Kh + H, generator=g) / 12,
          torch.randn(H, generator=g) / 10, torch.randn(3, H, generator=g) / 8, torch.randn(3, generator=g) / 10]
    ws = [w.to(dev).requires_grad_(True) for w in ws]
    want = fused.rgb_head(hr, geo, S, *ws)
    with torch.no_grad():
        got = fused.rgb_head(hr, geo, S, *ws)
    assert torch.equal(want, got) and not got.requires_grad
    enc = torch.randn(8, R * S, 1, generator=g).to(dev)
    dw = [torch.randn(H, 8, generator=g) / 3, torch.randn(H, generator=g) / 10, torch.randn(1, H, generator=g) / 8, torch.randn(1, generator=g) / 10]
    dw = [w.to(dev).requires_grad_(True) for w in dw]
    want = fused.density_mlp(enc, *dw)
    with torch.no_grad():
        got = fused.density_mlp(enc, *dw)
    assert torch.equal(want, got) and not got.requires_grad


@pytest.mark.parametrize("L,Fe,R,S", [(16, 2, 64, 32), (10, 4, 7, 16), (4, 2, 3, 128)])
def test_field_forward_rides_along(hip_lib, L, Fe, R, S):
    """emer_field_fwd (neck + rgb head in one launch, fused.RgbRider) against the two separate launches: geometry features,
    density and colour bit-identical; gradients of every parameter identical too (the backward is the separate kernels' in
    both cases, fed by the saved activations)."""
    from emernerf_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(L * 100 + R)
    N, H, Kh = R * S, 64, 49
    enc = (torch.randn(L, N, Fe, generator=g) * 0.5).to(dev)
    neck = [torch.randn(H, L * Fe, generator=g) / (L * Fe) ** 0.5, torch.randn(H, generator=g) * 0.1, torch.randn(H, H, generator=g) / 8, torch.randn(H, generator=g) * 0.1]
    head = [torch.randn(H, Kh + H, generator=g) / 10, torch.randn(H, generator=g) / 10, torch.randn(H, H + Kh + H, generator=g) / 12,
            torch.randn(H, generator=g) / 10, torch.randn(3, H, generator=g) / 8, torch.randn(3, generator=g) / 10]
    hray = torch.randn(R, Kh, generator=g).to(dev)
    wrgb = torch.randn(N, 3, generator=g).to(dev)
    wden = torch.randn(N, generator=g).to(dev)

    def run(ride):
        pn = [w.clone().to(dev).requires_grad_(True) for w in neck]
        ph = [w.clone().to(dev).requires_grad_(True) for w in head]
        hr = hray.clone().requires_grad_(True)
        fused.clear_riders()
        rider = fused.RgbRider(hr, S, ph, True) if ride else None
        geo, _, dens = fused.neck(enc, *pn, rider=rider)
        if ride:
            assert rider.out is not None, "the rider was not taken (shape not covered?)"
        rgb = fused.rgb_head(hr, geo, S, *ph)
        if ride:
            assert rgb.data_ptr() == rider.out.data_ptr(), "rgb_head launched its own forward"
        ((rgb * wrgb).sum() + (dens * wden).sum() + 0.01 * geo.sum()).backward()
        return [geo, dens, rgb] + [p.grad for p in pn + ph] + [hr.grad]

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a[:3], b[:3])):
        assert torch.equal(x, y), f"output {i} differs"
    for i, (x, y) in enumerate(zip(a[3:], b[3:])):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6 * float(y.abs().max())), f"gradient {i} differs"  # atomics in the reductions
    # a query with other tensors must not pick the parked results up
    pn = [w.clone().to(dev) for w in neck]
    ph = [w.clone().to(dev) for w in head]
    fused.clear_riders()
    rider = fused.RgbRider(hray, S, ph, False)
    with torch.no_grad():
        geo, _, _ = fused.neck(enc, *pn, rider=rider)
        other = [w.clone() for w in ph]
        rgb = fused.rgb_head(hray, geo, S, *other)
    assert rgb.data_ptr() != rider.out.data_ptr() and torch.equal(rgb, rider.out)
