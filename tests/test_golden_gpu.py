"""Path-level parity on MI355X: emernerf_amd's RadianceField / DensityField / PropNetEstimator / render_rays
against golden vectors recorded from the REFERENCE's own Python (tests/golden/make_golden.py).

Each case rebuilds the model from the recorded state_dict (reference parameter names), replays the
recorded stratified jitter and temporal-aggregation noise, and compares every output, the loss and the
recorded gradients.  Tolerance: composited quantities within 1e-4 relative (north star), written below.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden import make_golden as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

RTOL, ATOL = 1e-4, 2e-5  # composited outputs (rgb, depth, opacity, features ...)


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return {k: z[k] for k in z.files}


def _build(case, gold, dev):
    from emernerf_amd.prop_net import PropNetEstimator
    from emernerf_amd.radiance_field import build_density_field, build_radiance_field_from_cfg
    kw = {**G.CASES, **G.TINTERP_CASES, **G.SHIPPED_CASES}[case]
    cfg = G.model_cfg(kw["kind"], kw.get("tinterp", False), kw.get("grid", "toy"))
    torch.manual_seed(0)
    model = build_radiance_field_from_cfg(cfg, verbose=False)
    props = [build_density_field(aabb=G.AABB, unbounded=True, **k) for k in G.prop_kw(kw.get("grid", "toy"))]
    seed = int(gold["table_seed"])
    for prefix, m in [("model/", model)] + [(f"prop{i}/", p) for i, p in enumerate(props)]:
        sd = {}
        for k, v in m.state_dict().items():
            if k.endswith("tcnn_encoding.params"):
                sd[k] = G.table_values(prefix + k, v.numel(), seed)
            else:
                sd[k] = torch.from_numpy(gold["state/" + prefix + k])
        m.load_state_dict(sd)  # strict: every reference key must exist here and vice versa
        m.to(dev)
    model.time_diff = 1 / cfg.num_train_timesteps
    est = PropNetEstimator(None, None).to(dev)
    return cfg, model, props, est


def _check(name, got, want, rtol=RTOL, atol=ATOL):
    got = got.detach().float().cpu().numpy()
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    scale = max(float(np.abs(want).max()), 1e-6)
    err = np.abs(got - want)
    bad = err > (atol * max(scale, 1.0) + rtol * np.abs(want))
    assert not bad.any(), f"{name}: max err {err.max():.3e} (scale {scale:.3e}), {bad.sum()} / {bad.size} outside tolerance"


def _check_digest(name, got, gold, rtol=2e-3, flips=0.0):
    """``flips``: fraction of entries that may sit outside the elementwise tolerance -- 0 for the toy cases.  At the SHIPPED sizes a
    recording holds ~10^5 ReLU pre-activations, the grid encoding of a second correct implementation differs by ~1e-6, and the handful
    of units within that distance of zero take the other branch: each moves ONE sample's contribution of one weight-gradient row (and
    of the table entries under that sample).  Those entries are counted and bounded (<= ``flips`` of the tensor), everything else
    keeps the tolerance of the toy cases, and a dense tensor must agree to 2e-2 in the L2 norm (a digest: its norm to ``rtol``)."""
    t = got.detach().float().cpu()

    def close(a, want, scale, what, l2=False):
        bad = np.abs(a - want) > rtol * scale + rtol * np.abs(want)
        assert bad.mean() <= flips, f"{what}: {int(bad.sum())} of {bad.size} entries outside tolerance (max err {np.abs(a - want).max():.3e}, scale {scale:.3e})"
        if flips and l2:   # one flipped unit of one of the ~500 samples of these recordings is worth up to ~1e-2 of a weight gradient's norm
            err = float(np.linalg.norm((a - want).astype(np.float64))) / max(float(np.linalg.norm(want.astype(np.float64))), 1e-30)
            assert err <= 2e-2, f"{what}: relative L2 error {err:.3e}"
    if name in gold:
        want = gold[name]
        close(t.numpy(), want, max(float(np.abs(want).max()), 1e-12), name, l2=True)
    else:
        flat = t.reshape(-1).double()
        want = gold[name + "@sample"]
        scale = max(float(gold[name + "@norm"]) / np.sqrt(flat.numel()) * 10, 1e-12)
        close(flat[G.digest_indices(flat.numel())].float().numpy(), want, scale, name + "@sample")
        np.testing.assert_allclose(float(flat.norm()), float(gold[name + "@norm"]), rtol=rtol, err_msg=name + "@norm")
        if name + "@top_idx" in gold:   # [r6] the entries of largest magnitude, by position
            idx, val = torch.from_numpy(gold[name + "@top_idx"]), gold[name + "@top_val"]
            close(flat[idx].float().numpy(), val, float(np.abs(val).max()), name + "@top")


@pytest.mark.parametrize("case", list(G.CASES) + list(G.TINTERP_CASES) + list(G.SHIPPED_CASES))
def test_render_rays_matches_reference(hip_lib, case):
    """(``*_shipped_*`` / ``static_encdefaults_train``: the reference recorded at the SHIPPED grid hyper-parameters -- static xyz
    D3/L10/F4/T2^20, xyzt D4/L10/F4/T2^18, proposal nets L8/F1/T2^20 of configs/default_config.yaml:51-77 and the HashEncoder defaults
    D3/L16/F2/T2^19 of encodings.py:110-118 that bench.py times: 256-slice bitmaps, the run-reduced / wide-pair / tail-split paths of the
    owner-computes backward and the dense-level LDS staging all sit behind these tables, none behind the toy ones.)
    (flow_eval_tinterp: evaluation with enable_temporal_interpolation on rays whose timestamps lie between the training timesteps --
    the reference's temporal_interpolation of the flow field, radiance_field.py:359-389,844-905, recorded and compared like every other case)"""
    from emernerf_amd.render_utils import render_rays
    dev = torch.device("cuda:0")
    gold = _load(case)
    kw = {**G.CASES, **G.TINTERP_CASES, **G.SHIPPED_CASES}[case]
    cfg, model, props, est = _build(case, gold, dev)
    train = kw["mode"] == "train"
    model.train(train); est.train(train)
    for p in props:
        p.train(train)
    prefix = "lidar_" if kw.get("lidar") else ""
    data = {k[len("data/"):]: torch.from_numpy(v).to(dev) for k, v in gold.items() if k.startswith("data/")}
    jitters = [torch.from_numpy(gold[f"jitter/{i}"]).to(dev) for i in range(sum(k.startswith("jitter/") for k in gold))]
    noises = [torch.from_numpy(gold[f"noise/{i}"]).to(dev) for i in range(sum(k.startswith("noise/") for k in gold))]
    jit_it, noise_it = iter(jitters), iter(noises)
    est.jitter_fn = lambda n, d: next(jit_it)
    if noises:
        model._noise = lambda like: next(noise_it).reshape(*like.shape[:-1], 1)
    rcfg = G.render_cfg(list(kw["prop_samples"]), kw["num_samples"])
    results = render_rays(radiance_field=model, proposal_estimator=est, proposal_networks=props, data_dict=data, cfg=rcfg,
                          proposal_requires_grad=train, return_decomposition=not train, prefix=prefix)
    assert next(jit_it, None) is None and next(noise_it, None) is None, "all recorded randomness must be consumed"

    out_keys = {k[len("out/"):] for k in gold if k.startswith("out/")}
    ex_keys = {k[len("extras/"):] for k in gold if k.startswith("extras/")}
    assert set(results) - {"extras"} == out_keys, f"result keys differ: {set(results) ^ out_keys}"
    assert set(results["extras"]) == ex_keys, f"extras keys differ: {set(results['extras']) ^ ex_keys}"
    for k in sorted(out_keys):
        if k == "median_depth":  # index-valued (the sample at which the accumulated weight crosses one half)
            g, w = results[k].detach().cpu().numpy().reshape(-1), gold["out/" + k].reshape(-1)
            same = np.isclose(g, w, rtol=1e-4)
            assert same.mean() >= 0.97, f"{k}: only {same.mean():.3f} of the rays agree"
            tv, wt = gold["extras/t_vals"], gold["extras/weights"]
            if not same.all():
                # a ray that disagrees may only have slipped to the NEIGHBOURING sample, and only where the recorded cumulative
                # weight sits within rounding of one half at that boundary (sample indices are compared, not depths)
                assert tv.shape[0] == g.shape[0], "median depth differs on a chunked render whose extras cover the last chunk only"
                ig = np.abs(tv - g[:, None]).argmin(1)
                iw = np.abs(tv - w[:, None]).argmin(1)
                assert (np.abs(ig - iw)[~same] <= 1).all(), f"{k}: a ray moved by more than one sample"
                cw = np.cumsum(wt.astype(np.float64), axis=1)
                lo = np.minimum(ig, iw)
                edge = np.abs(cw[np.arange(len(lo)), lo] - 0.5)
                assert (edge[~same] <= 1e-5).all(), f"{k}: a ray slipped although its cumulative weight is {edge[~same].max():.2e} away from 0.5"
            continue
        _check("out/" + k, results[k], gold["out/" + k])
    for k in sorted(ex_keys):
        _check("extras/" + k, results["extras"][k], gold["extras/" + k], rtol=2e-4, atol=5e-5)

    if not train:
        return
    flips = 0.01 if kw.get("grid", "toy") != "toy" else 0.0   # shipped sizes: a counted handful of ReLU flips (see _check_digest)
    prop_loss = est.compute_loss(results["extras"]["trans"], loss_scaler=1024)
    np.testing.assert_allclose(float(prop_loss), float(gold["prop_loss"]), rtol=2e-3)
    for p in props:
        p.zero_grad()
    prop_loss.backward()
    for i, p in enumerate(props):
        for k, q in p.named_parameters():
            assert (q.grad is not None) == bool(gold[f"prop_has_grad/{i}/{k}"]), f"prop{i}.{k}: grad presence differs"
            key = f"prop_grad/{i}/{k}"
            if q.grad is not None and (key in gold or key + "@sample" in gold):
                _check_digest(key, q.grad, gold, rtol=5e-3, flips=flips)
    loss = G.golden_loss(results, data, prefix)
    np.testing.assert_allclose(float(loss), float(gold["loss"]), rtol=1e-4)
    model.zero_grad()
    loss.backward()
    named = dict(model.named_parameters())
    checked = 0
    for k in G.GRAD_KEYS:
        key = "grad/" + k
        if key in gold or key + "@sample" in gold:
            assert named[k].grad is not None, f"{k}: no gradient"
            _check_digest(key, named[k].grad, gold, flips=flips)
            checked += 1
    assert checked >= 5


# ------------------------------------------------------------------------------------------ N3: the evaluation render loop
@pytest.mark.parametrize("case", list(G.RENDER_PIXELS_CASES))
def test_render_pixels_matches_reference_loop(hip_lib, case):
    """emernerf_amd.video_utils.render_pixels against a RECORDING of the reference's own radiance_fields/video_utils.py
    render_pixels / render (:50-468, run unmodified over the import shims on a stub split; make_golden.py): the same keys,
    every per-image list (rgbs, depths, opacities, static / dynamic decomposition with the green-screen blend, shadow
    variants, flows, ground truth) and the psnr the loop computes."""
    from emernerf_amd.prop_net import PropNetEstimator
    from emernerf_amd.radiance_field import build_density_field, build_radiance_field_from_cfg
    from emernerf_amd.video_utils import render_pixels
    dev = torch.device("cuda:0")
    gold = _load(case)
    kw = G.RENDER_PIXELS_CASES[case]
    cfg = G.model_cfg(kw["kind"])
    torch.manual_seed(0)
    model = build_radiance_field_from_cfg(cfg, verbose=False)
    props = [build_density_field(aabb=G.AABB, unbounded=True, **k) for k in G.PROP_KW]
    seed = int(gold["table_seed"])
    for prefix, m in [("model/", model)] + [(f"prop{i}/", p) for i, p in enumerate(props)]:
        sd = {k: (G.table_values(prefix + k, v.numel(), seed) if k.endswith("tcnn_encoding.params")
                  else torch.from_numpy(gold["state/" + prefix + k])) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        m.to(dev)
    model.time_diff = 1 / cfg.num_train_timesteps
    est = PropNetEstimator(None, None).to(dev)
    n_img = len({k.split("/")[0] for k in gold if k.startswith("image")})
    images = [{k[len(f"image{i}/"):]: torch.from_numpy(v).to(dev) for k, v in gold.items() if k.startswith(f"image{i}/")} for i in range(n_img)]
    rcfg = G.render_cfg([24, 16], 16, chunk=25)
    out = render_pixels(rcfg, model, est, G.GoldenSplit(images), proposal_networks=props, compute_metrics=True, vis_indices=[0, 2],
                        return_decomposition=True)
    want_lists = sorted({k.split("/")[1] for k in gold if k.startswith("res/")})
    got_lists = sorted(k for k, v in out.items() if isinstance(v, list) and len(v) > 0)
    assert got_lists == want_lists, (got_lists, want_lists)
    # keys the reference returns even when empty, and its scalars
    for k in ("rgbs", "static_rgbs", "dynamic_rgbs", "depths", "opacities", "static_depths", "static_opacities", "dynamic_depths",
              "dynamic_opacities", "psnr", "ssim", "feat_psnr", "masked_psnr", "masked_ssim", "masked_feat_psnr"):
        assert k in out, k
    np.testing.assert_allclose(out["psnr"], float(gold["scalar/psnr"]), rtol=1e-4)
    for k in want_lists:
        assert len(out[k]) == 2, k
        for j in range(2):
            got, want = np.asarray(out[k][j]), gold[f"res/{k}/{j}"]
            assert got.shape == want.shape, (k, got.shape, want.shape)
            if k == "median_depths":
                # index-valued: a pixel may slip to the NEIGHBOURING sample when its cumulative weight crosses one half within
                # rounding; the loop does not return the samples, so the neighbour rule is checked on this path's own eval-mode
                # samples of the same image (sample placement is bit-exact against the oracle: tests/test_kernels_gpu.py)
                same = np.isclose(got, want, rtol=1e-4)
                assert same.mean() >= 0.97, f"{k}[{j}]: only {same.mean():.3f} of the pixels agree"
                if not same.all():
                    from emernerf_amd.render_utils import render_rays
                    big = G.render_cfg([24, 16], 16, chunk=1 << 20)
                    with torch.no_grad():
                        one = render_rays(radiance_field=model, proposal_estimator=est, proposal_networks=props, data_dict=images[[0, 2][j]],
                                          cfg=big, return_decomposition=True)
                    tv = one["extras"]["t_vals"].reshape(-1, 16).cpu().numpy()
                    ig = np.abs(tv - got.reshape(-1, 1)).argmin(1)
                    iw = np.abs(tv - want.reshape(-1, 1)).argmin(1)
                    assert (np.abs(ig - iw)[~same.reshape(-1)] <= 1).all(), f"{k}[{j}]: a pixel moved by more than one sample"
                continue
            scale = max(float(np.abs(want).max()), 1.0)
            err = np.abs(got - want)
            assert (err <= 2e-5 * scale + 1e-4 * np.abs(want)).all(), f"{k}[{j}]: max err {err.max():.3e}"


def test_pixel_source_matches_reference_recording(hip_lib):
    """N2 against a recording of the reference's own datasets/base/pixel_source.py (tests/golden/make_golden.py::run_pixel_source_case):
    ``get_rays`` (:39-76) on recorded pixels / cameras, the batch ``get_train_rays`` (:666-731) assembled for the pixels IT drew
    (its torch.randint draws are part of the recording; our sampler draws its own), and ``get_render_rays`` (:733-826) of one image,
    key for key.  Origins, pixel coordinates, gathered colours / masks / ids: bit-exact; directions: 2e-7 (the reference normalises
    with ``/ (norm + 1e-8)`` in torch's evaluation order)."""
    from emernerf_amd.pixel_source import PixelSource, get_rays
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(HERE, "golden", "pixel_source.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    src = PixelSource(t("src/images"), t("src/cam_to_worlds"), t("src/intrinsics"), t("src/sky_masks"), t("src/normalized_timestamps"), t("src/cam_ids"))
    H, W = src.HEIGHT, src.WIDTH
    # get_rays
    cam = t("get_rays/cam")
    o, d, n = get_rays(t("get_rays/x"), t("get_rays/y"), src.cam_to_worlds[cam], src.intrinsics[cam])
    np.testing.assert_array_equal(o.cpu().numpy(), z["get_rays/origins"])
    np.testing.assert_allclose(d.cpu().numpy(), z["get_rays/viewdirs"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(n.cpu().numpy(), z["get_rays/direction_norm"], rtol=2e-7)
    # the training batch of the recorded pixel draws
    pc = z["train/pixel_coords"]
    y, x = np.rint(pc[:, 0] * H).astype(np.int64), np.rint(pc[:, 1] * W).astype(np.int64)
    got = src._gather(t("train/img_idx"), torch.from_numpy(y).to(dev), torch.from_numpy(x).to(dev))
    want_keys = {k.split("/", 1)[1] for k in z.files if k.startswith("train/")}
    assert set(got.keys()) == want_keys, (sorted(got.keys()), sorted(want_keys))
    for k in want_keys:
        a, b = got[k].cpu().numpy(), z["train/" + k]
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        if k in ("viewdirs", "direction_norms"):
            np.testing.assert_allclose(a, b, rtol=2e-7, atol=2e-7, err_msg=k)
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)
    # every pixel of image 4, image-shaped
    rr = src.get_render_rays(4)
    want_keys = {k.split("/", 1)[1] for k in z.files if k.startswith("render4/")}
    assert set(rr.keys()) == want_keys, (sorted(rr.keys()), sorted(want_keys))
    for k in want_keys:
        a, b = rr[k].cpu().numpy(), z["render4/" + k]
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        if k in ("viewdirs", "direction_norm"):
            np.testing.assert_allclose(a, b, rtol=2e-7, atol=2e-7, err_msg=k)
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)


def test_lidar_source_matches_reference_recording(hip_lib):
    """N2, lidar half, against a recording of the reference's own datasets/base/lidar_source.py (tests/golden/make_golden.py::
    run_lidar_source_case): the cached per-timestep subset after a first candidate list, after a CHANGED list and for the same
    candidates given as a Tensor (:246-275), the ``get_train_rays`` batch assembled for the indices IT drew (:277-308; its torch.randint
    draw is part of the recording, our kernel draws its own), ``get_render_rays`` (:310-330), the timestamp registry -- key for key,
    bit for bit (gathers only)."""
    from emernerf_amd.lidar_source import LidarSource
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(HERE, "golden", "lidar_source.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    src = LidarSource(t("src/origins"), t("src/directions"), t("src/ranges"), t("src/timesteps"))
    src.register_normalized_timestamps(t("src/normalized_timestamps"))
    assert src.num_timesteps == int(z["num_timesteps"]) and int(src.find_closest_timestep(0.37)) == int(z["closest_0p37"])
    for tag in ("first", "changed", "tensor"):
        cand = z[tag + "/cand"]
        cand = torch.from_numpy(cand).to(dev) if tag == "tensor" else [int(c) for c in cand]
        before = src.cached_origins
        got = src.get_train_rays(0, candidate_indices=cand, lidar_idx=t(tag + "/lidar_idx"))
        if tag == "tensor":
            assert src.cached_origins is before, "a Tensor equal to the cached candidates must not rebuild the cache"
        np.testing.assert_array_equal(src.cached_indices.cpu().numpy(), z[tag + "/cached_indices"])
        for k in ("cached_origins", "cached_directions", "cached_ranges", "cached_normalized_timestamps"):
            np.testing.assert_array_equal(getattr(src, k).cpu().numpy(), z[tag + "/" + k], err_msg=f"{tag}/{k}")
        want_keys = {k.split("/")[-1] for k in z.files if k.startswith(tag + "/batch/")}
        assert set(got.keys()) == want_keys, (sorted(got.keys()), sorted(want_keys))
        for k in want_keys:
            a, b = got[k].cpu().numpy(), z[f"{tag}/batch/{k}"]
            assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
            np.testing.assert_array_equal(a, b, err_msg=f"{tag}/{k}")
    rr = src.get_render_rays(2)
    want_keys = {k.split("/", 1)[1] for k in z.files if k.startswith("render2/")}
    assert set(rr.keys()) == want_keys
    for k in want_keys:
        np.testing.assert_array_equal(rr[k].cpu().numpy(), z["render2/" + k], err_msg=k)
    # our own draw: indices inside the cached subset, every cached scan hit, fresh rays per call, batch == gather of the drawn indices
    idx = src.sample_uniform_rays(4096, candidate_indices=[0, 2, 5, 3])
    n_c = src.cached_origins.shape[0]
    assert int(idx.min()) >= 0 and int(idx.max()) < n_c and idx.dtype == torch.int64
    hit = torch.bincount(idx, minlength=n_c)
    assert int((hit == 0).sum()) == 0 and float(hit.float().std()) < 3.0 * (4096 / n_c) ** 0.5 + 2, "uniform over the cached points"
    assert not torch.equal(idx, src.sample_uniform_rays(4096, candidate_indices=[0, 2, 5, 3])), "the seed word advances per call"
    b1 = src.get_train_rays(256, candidate_indices=[0, 2, 5, 3])
    assert b1["lidar_ranges"].shape == (256, 1) and b1["lidar_normed_timestamps"].shape == (256,)
    allowed = torch.tensor([0.0, 2.0, 5.0, 3.0], device=dev) / 5.0
    assert bool(torch.isin(b1["lidar_normed_timestamps"], allowed).all()), "only rays of the candidate scans"
    from emernerf_amd import _lib
    fresh = LidarSource(t("src/origins"), t("src/directions"), t("src/ranges"), t("src/timesteps"), t("src/normalized_timestamps"))
    with pytest.raises(_lib.EmerError):
        fresh.get_train_rays(16)   # the reference indexes a cache that does not exist (TypeError); here: a clear error
    with pytest.raises(TypeError):
        fresh.get_train_rays(16, candidate_indices=torch.tensor([1, 2], device=dev))


def test_lidar_step_trains_on_the_lidar_source(hip_lib):
    """Trainer.lidar_step (train_emernerf.py:747-826) fed from LidarSource.get_train_rays the way the reference's loop feeds it
    (train_emernerf.py:749-752): the depth loss of a fixed validation batch goes down over a few steps."""
    from emernerf_amd.lidar_source import LidarSource
    from emernerf_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    src = LidarSource.synthetic(dev, num_timesteps=10, points_per_scan=2048, seed=4)
    train_steps = [t for t in range(10) if t % 5 != 4]          # hold out scans 4 and 9, as a test split would
    tr = Trainer(kind="dynamic", device=dev, num_samples=32, prop_samples=(32, 16), table_init=0.3, seed=2, num_iters=400)
    losses = []
    for _ in range(12):
        batch = src.get_train_rays(1024, candidate_indices=train_steps)
        assert set(batch) == {"lidar_origins", "lidar_viewdirs", "lidar_ranges", "lidar_normed_timestamps"}
        losses.append(float(tr.lidar_step(batch)["loss"]))
        tr.step_count += 1
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3]), losses
