"""CPU tests (no GPU): the oracle is pinned before anything trusts it.

* the C restatement agrees with an independent vectorised torch restatement, passes fp64 gradcheck and the
  structural invariants of SURVEY.md section 8c(iii);
* oracle/ref_path.py (the CPU restatement of the whole path, used as run-time checker and cpu_baseline on
  the GPU box) reproduces the golden vectors recorded from the REFERENCE's own Python.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden import make_golden as G

HERE = os.path.dirname(os.path.abspath(__file__))

SURVEY_TABLE = {  # (D, L, base, max, T, F) -> entries
    (3, 10, 16, 8192, 20, 4): 7639040, (4, 10, 32, 8192, 18, 4): 2621440, (4, 10, 16, 4096, 18, 4): 2424832,
    (3, 8, 16, 512, 20, 1): 4661184, (3, 8, 16, 2048, 20, 1): 5541888, (3, 16, 16, 2048, 19, 2): 6098120,
    (3, 4, 16, 2048, 19, 2): 1576960,
}


@pytest.mark.parametrize("args,entries", list(SURVEY_TABLE.items()))
def test_level_table_matches_survey(oracle, args, entries):
    m = oracle.grid_meta_from_encoder_args(*args)
    assert m.n_entries == entries
    assert (m.offset[1:] == np.cumsum(m.size)[:-1]).all() and (m.size % 8 == 0).all()
    assert (m.size <= 2 ** args[4]).all()
    for l in range(m.n_levels):  # dense iff res^D fits
        assert bool(m.hashed[l]) == (int(m.res[l]) ** args[0] > int(m.size[l]))


@pytest.mark.parametrize("args", [(3, 16, 16, 2048, 19, 2), (4, 10, 32, 8192, 18, 4), (3, 8, 16, 512, 20, 1), (2, 5, 8, 256, 12, 2)])
def test_c_oracle_vs_torch_restatement(oracle, args):
    m = oracle.grid_meta_from_encoder_args(*args)
    g = torch.Generator().manual_seed(0)
    N = 1500
    x = torch.rand(N, args[0], generator=g)
    x[0] = 0.0; x[1] = 1.0 - 2 ** -24; x[2, 0] = 0.9671
    p = torch.rand(m.n_params, generator=g) - 0.5
    xd, pd = x.double().requires_grad_(True), p.double().requires_grad_(True)
    out = oracle.hashgrid_torch(xd, pd, m)
    do = torch.randn(out.shape, generator=g, dtype=torch.double)
    out.backward(do)
    a = oracle.hashgrid_fwd(m, x, p)
    np.testing.assert_allclose(a, out.detach().numpy(), atol=5e-7)
    gp = oracle.hashgrid_bwd_params(m, x, do.float())
    np.testing.assert_allclose(gp, pd.grad.numpy(), atol=1e-6 * np.abs(gp).max())
    gx = oracle.hashgrid_bwd_input(m, x, p, do.float())
    np.testing.assert_allclose(gx, xd.grad.numpy(), atol=2e-6 * np.abs(gx).max())


def test_grid_gradcheck_fp64(oracle):
    m = oracle.grid_meta(3, 3, 2, 8, 4, 1.5)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(6, 3, generator=g, dtype=torch.double) * 0.9 + 0.05).requires_grad_(True)
    p = torch.rand(m.n_params, generator=g, dtype=torch.double).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b: oracle.hashgrid_torch(a, b, m, f32_cells=False), (x, p), eps=1e-7, atol=1e-5)


def test_dense_level_index_is_analytic(oracle):
    """On a dense level the table is the grid itself: encoding a lattice point returns its entry."""
    m = oracle.grid_meta(3, 1, 1, 19, 16, 1.0)  # one dense level, res 16, scale 15
    assert not m.hashed[0] and m.res[0] == 16
    p = torch.arange(m.n_params, dtype=torch.float32)
    ijk = torch.tensor([[3, 5, 7], [0, 0, 0], [14, 2, 9]])
    x = (ijk.float() - 0.5) / 15.0  # pos = 15*x + 0.5 = integer -> w = 0 -> corner (i,j,k) exactly
    x = x.clamp_min(0)
    out = oracle.hashgrid_fwd(m, x, p)[:, 0]
    want = ijk[:, 0] + 16 * ijk[:, 1] + 256 * ijk[:, 2]
    np.testing.assert_allclose(out[0], float(want[0]), atol=1e-3)
    np.testing.assert_allclose(out[2], float(want[2]), atol=1e-3)


def test_sampler_invariants(oracle):
    g = torch.Generator().manual_seed(2)
    R, m, n = 50, 65, 128
    w = torch.rand(R, m - 1, generator=g) ** 3
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(w, -1)], -1)
    cdf = cdf / cdf[:, -1:]
    vals = torch.sort(torch.rand(R, m, generator=g), -1).values
    for jit in (None, torch.rand(R, generator=g)):
        s = oracle.importance_sample(vals, cdf, n, jit)
        assert s.shape == (R, n + 1)
        assert (np.diff(s, axis=-1) >= 0).all(), "edges are sorted"
        assert (s >= vals.numpy()[:, :1] - 1e-6).all() and (s <= vals.numpy()[:, -1:] + 1e-6).all()
    # uniform CDF on [0,1]: centre-of-bin placement (SURVEY A.2 frozen spec)
    u = torch.tensor([[0.0, 1.0]]).repeat(3, 1)
    s = oracle.importance_sample(u, u, 4, None)
    np.testing.assert_allclose(s[0], (np.arange(5) + 0.5) / 5, rtol=1e-6)
    # empty / degenerate: a flat CDF returns interval midpoints instead of dividing by zero
    flat = torch.tensor([[0.3, 0.3, 0.3]])
    s = oracle.importance_sample(torch.tensor([[0.0, 0.5, 1.0]]), flat, 3, None)
    assert np.isfinite(s).all()


def _end_clamp_case():
    """A ray whose CDF saturates before its last edge (d < 1e-10 on the tail) and a stratified offset of 1 - 2^-24: (k + beta)
    rounds to n + 1 for k = n, so u_n == cdf_last and the upper-bound search runs off the end of the row."""
    m, n = 9, 128
    vals = torch.linspace(0.0, 1.0, m).repeat(3, 1).contiguous()
    cdf = torch.tensor([0.0, 0.1, 0.4, 0.9, 1.0, 1.0, 1.0, 1.0, 1.0]).repeat(3, 1).contiguous()
    cdf[1] = cdf[1] * 0.5 + 0.25               # first / last values other than 0 / 1
    cdf[2] = torch.linspace(0.0, 1.0, m)        # no saturated tail: the last edge is reached by interpolation
    jit = torch.full((3,), float(np.float32(1.0) - np.float32(2.0 ** -24)))
    return vals, cdf, n, jit


def test_sampler_end_clamp_returns_the_last_edge(oracle):
    """nerfacc's pdf.cu clamps the two bracketing indices separately (p0 = clamp(p - 1), p1 = clamp(p)): a u at or beyond the last
    CDF value brackets (m - 1, m - 1) and yields v[m - 1] -- not the midpoint of the last interval, which a joint clamp of p to
    [0, m - 2] gives on a saturated tail."""
    vals, cdf, n, jit = _end_clamp_case()
    s = oracle.importance_sample(vals, cdf, n, jit)
    assert (np.float32(n) + jit[0].numpy()) == np.float32(n + 1), "the case must reach u == cdf_last"
    np.testing.assert_array_equal(s[:, -1], vals[:, -1].numpy())
    assert (np.diff(s, axis=-1) >= 0).all()
    # one step before the end nothing changes: interior samples interpolate inside their bracket
    assert (s[:, :-1] <= vals[:, -1:].numpy()).all() and (s[0, :-1] < 0.5 + 1e-6).all()


def test_stot_matches_reference_lambdas(oracle):
    """nerfacc_prop_net.py:307-308 evaluated with torch, bit for bit."""
    s = torch.rand(1000, generator=torch.Generator().manual_seed(3))
    f = lambda x: torch.where(x < 200, x / 400, 1 - 1 / (2 * x / 200))  # noqa: E731
    inv = lambda x: torch.where(x < 0.5, x * 400, 200 / (2 - 2 * x))  # noqa: E731
    s_min, s_max = f(torch.tensor(0.1)), f(torch.tensor(1000.0))
    want = inv(s * s_max + (1 - s) * s_min)
    got = oracle.stot(s, 0.1, 1000.0, "uniform_lindisp")
    assert np.array_equal(got.view(np.uint32), want.numpy().view(np.uint32))


def test_volrend_brute_force(oracle):
    g = torch.Generator().manual_seed(4)
    R, S = 7, 33
    e = torch.sort(torch.rand(R, S + 1, generator=g) * 30, -1).values
    ts, te, sg = e[:, :-1].contiguous(), e[:, 1:].contiguous(), torch.rand(R, S, generator=g) * 3
    w, T, a = oracle.render_weights(ts, te, sg)
    for r in range(R):
        acc = 0.0
        for s in range(S):
            sdt = float(sg[r, s] * (te[r, s] - ts[r, s]))
            assert abs(T[r, s] - np.exp(-acc)) < 1e-5 and abs(a[r, s] - (1 - np.exp(-sdt))) < 1e-5
            acc += sdt
    assert (w.sum(-1) <= 1 + 1e-5).all() and (np.diff(T, axis=-1) <= 1e-7).all()
    v = torch.randn(R, S, 5, generator=g)
    np.testing.assert_allclose(oracle.accumulate(w, v), (torch.from_numpy(w)[..., None] * v).sum(1).numpy(), rtol=1e-5, atol=1e-6)


def test_contract_matches_reference_expression(oracle):
    g = torch.Generator().manual_seed(5)
    aabb = torch.tensor(G.AABB)
    pos = (torch.rand(4000, 3, generator=g) - 0.5) * 600
    from oracle import ref_path
    want = ref_path.contract_points(pos, aabb, True).numpy()
    got = oracle.contract(pos, aabb, True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "C contraction == torch expression, bit for bit"


# ------------------------------------------------------------------- ref_path pinned on the reference
def _grids(oracle, kind, grid="toy"):
    c = G.model_cfg(kind, grid=grid)
    grids = {"model/xyz_encoder": oracle.grid_meta_from_encoder_args(3, c.xyz_encoder.n_levels, c.xyz_encoder.base_resolution,
                                                                       c.xyz_encoder.max_resolution, c.xyz_encoder.log2_hashmap_size,
                                                                       c.xyz_encoder.n_features_per_level)}
    d = c.dynamic_xyz_encoder
    grids["model/dynamic_xyz_encoder"] = oracle.grid_meta_from_encoder_args(4, d.n_levels, d.base_resolution, d.max_resolution,
                                                                            d.log2_hashmap_size, d.n_features_per_level)
    grids["model/flow_xyz_encoder"] = oracle.grid_meta_from_encoder_args(4, 10, 16, 4096, 18, 4)  # radiance_field.py:916-923
    for i, kw in enumerate(G.prop_kw(grid)):
        grids[f"prop{i}/xyz_encoder"] = oracle.grid_meta_from_encoder_args(3, kw["n_levels"], 16, kw["max_resolution"],
                                                                           kw["log2_hashmap_size"], kw["n_features_per_level"])
    return grids


@pytest.mark.parametrize("case", ["static_train", "dynamic_train", "flow_train", "flow_lidar_train", "feature_train"] + list(G.SHIPPED_CASES))
def test_ref_path_reproduces_reference_goldens(oracle, case):
    """(the ``*_shipped_*`` / ``*_encdefaults_*`` cases: the reference's Python recorded at the SHIPPED grid hyper-parameters --
    configs/default_config.yaml:51-77, encodings.py:110-118 -- so the port's level tables of 2^18 .. 2^20 entries meet the reference)"""
    from oracle.ref_path import RefPath, prop_loss
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    gold = {k: z[k] for k in z.files}
    kw = {**G.CASES, **G.SHIPPED_CASES}[case]
    grids = _grids(oracle, kw["kind"], kw.get("grid", "toy"))
    seed = int(gold["table_seed"])
    states = {"model/": {}, "prop0/": {}, "prop1/": {}}
    for k in gold:
        for pre in states:
            if k.startswith("state/" + pre):
                states[pre][k[len("state/" + pre):]] = torch.from_numpy(gold[k])
    for pre, st in states.items():
        for enc in ("xyz_encoder", "dynamic_xyz_encoder", "flow_xyz_encoder"):
            gk = pre + enc
            needed = gk in grids and (pre != "model/" or enc == "xyz_encoder" or (enc == "dynamic_xyz_encoder" and kw["kind"] != "static")
                                      or (enc == "flow_xyz_encoder" and kw["kind"] in ("flow", "feature")))
            if needed:
                name = enc + ".tcnn_encoding.params"
                st[name] = G.table_values(pre + name, grids[gk].n_params, seed)
    ref = RefPath(states["model/"], [states["prop0/"], states["prop1/"]], grids, G.AABB, time_diff=0.1,
                  cam_embedding=kw["kind"] == "feature")
    prefix = "lidar_" if kw.get("lidar") else ""
    data = {k[len("data/"):]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("data/")}
    jit = [torch.from_numpy(gold[f"jitter/{i}"]) for i in range(sum(k.startswith("jitter/") for k in gold))]
    noises = iter([torch.from_numpy(gold[f"noise/{i}"]) for i in range(sum(k.startswith("noise/") for k in gold))])
    res = ref.render_rays(data, kw["num_samples"], list(kw["prop_samples"]), jitters=jit,
                          noise_fn=lambda like: next(noises).reshape(*like.shape[:-1], 1), requires_grad=True, prefix=prefix)
    for k in [k for k in gold if k.startswith("out/")]:
        name = k[len("out/"):]
        np.testing.assert_allclose(res[name].detach().numpy(), gold[k], rtol=2e-5, atol=2e-6, err_msg=k)
    for k in [k for k in gold if k.startswith("extras/")]:
        np.testing.assert_allclose(res["extras"][k[len("extras/"):]].detach().numpy(), gold[k], rtol=2e-5, atol=2e-6, err_msg=k)
    pl = prop_loss(ref.cache, res["extras"]["trans"], 1024)
    np.testing.assert_allclose(float(pl), float(gold["prop_loss"]), rtol=1e-5)
    loss = G.golden_loss(res, data, prefix)
    np.testing.assert_allclose(float(loss), float(gold["loss"]), rtol=1e-5)
    loss.backward()
    for k in ("base_mlp.0.weight", "rgb_head.layers.1.weight", "appearance_embedding.weight", "dino_head.4.weight", "learnable_pe_map",
              "pe_head.0.weight"):
        if "grad/" + k in gold:
            np.testing.assert_allclose(ref.t["model/" + k].grad.numpy(), gold["grad/" + k], rtol=1e-4,
                                       atol=1e-6 * np.abs(gold["grad/" + k]).max(), err_msg=k)
