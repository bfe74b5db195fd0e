"""the reference's OWN Python on the four import swaps (CPU, build container only).

The reference checkout is read-only and absent from the GPU box, so the swaps are installed the way Python resolves them:
``sys.modules`` entries for ``third_party.tcnn_modules`` -> ``emernerf_amd.tcnn_modules`` and ``nerfacc`` (+ ``nerfacc.data_specs``,
``.estimators.base``, ``.pdf``, ``.volrend``) -> ``emernerf_amd.nerfacc_compat`` -- exactly the names the four edited import lines
(radiance_fields/encodings.py:9, radiance_fields/render_utils.py:4-8, third_party/nerfacc_prop_net.py:11-14, loss/base.py:7) would
bind.  Then the reference's ``build_radiance_field_from_cfg`` / ``build_density_field`` / ``PropNetEstimator`` are constructed
unmodified and compared with ``emernerf_amd.radiance_field``: same ``state_dict`` keys, shapes and (same seed) values.  No kernel
runs: this is the import + construction half; the binding points' arithmetic is tests/test_a_metric_shape_gpu.py.
Runs in a subprocess so that the reference's top-level packages do not leak into this session's ``sys.modules``.
"""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, types
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, %(ref)r)
    import torch
    import emernerf_amd.tcnn_modules as amd_tcnn
    import emernerf_amd.nerfacc_compat as NC
    # --- the four swaps, as sys.modules bindings
    import third_party                                   # the reference's (empty-__init__) package
    sys.modules["third_party.tcnn_modules"] = amd_tcnn   # encodings.py:9
    third_party.tcnn_modules = amd_tcnn
    nerfacc = types.ModuleType("nerfacc")
    for name in ("accumulate_along_rays", "render_transmittance_from_density", "render_weight_from_density"):
        setattr(nerfacc, name, getattr(NC, name))        # render_utils.py:4-8, loss/base.py:7
    def sub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    sys.modules["nerfacc"] = nerfacc
    nerfacc.data_specs = sub("nerfacc.data_specs", RayIntervals=NC.RayIntervals)                       # nerfacc_prop_net.py:11
    nerfacc.estimators = sub("nerfacc.estimators")
    nerfacc.estimators.base = sub("nerfacc.estimators.base", AbstractEstimator=NC.AbstractEstimator)   # :12
    nerfacc.pdf = sub("nerfacc.pdf", importance_sampling=NC.importance_sampling, searchsorted=NC.searchsorted)   # :13
    nerfacc.volrend = sub("nerfacc.volrend", render_transmittance_from_density=NC.render_transmittance_from_density)  # :14
    # not a swap: omegaconf is simply absent from this image (type annotations only on this path)
    sub("omegaconf", OmegaConf=type("OmegaConf", (), {}))

    from radiance_fields.radiance_field import build_radiance_field_from_cfg as ref_build, build_density_field as ref_density
    from radiance_fields import render_utils as ref_render_utils          # imports the three nerfacc names
    from third_party.nerfacc_prop_net import PropNetEstimator as RefEstimator, get_proposal_requires_grad_fn
    import loss.base as ref_loss                                          # imports accumulate_along_rays
    import radiance_fields.encodings as ref_enc
    assert ref_enc.tcnn is amd_tcnn, "encodings.py must have bound the swapped tcnn module"
    assert ref_render_utils.accumulate_along_rays is NC.accumulate_along_rays and ref_loss.accumulate_along_rays is NC.accumulate_along_rays

    from emernerf_amd.radiance_field import build_radiance_field_from_cfg as amd_build, build_density_field as amd_density
    from emernerf_amd.trainer import AABB, PROP_KW, model_config
    for kind in ("static", "dynamic", "flow", "feature"):
        cfg = model_config(kind, num_cams=3 if kind == "feature" else 1)
        torch.manual_seed(0); ref = ref_build(cfg, verbose=False)
        torch.manual_seed(0); amd = amd_build(cfg, verbose=False)
        rs, as_ = ref.state_dict(), amd.state_dict()
        assert list(rs.keys()) == list(as_.keys()), (kind, sorted(set(rs) ^ set(as_)))
        for k in rs:
            assert rs[k].shape == as_[k].shape and rs[k].dtype == as_[k].dtype, (kind, k)
            assert torch.equal(rs[k], as_[k]), (kind, k, "same seed, same construction order -> same initial values")
        enc = ref.xyz_encoder.tcnn_encoding
        assert type(enc) is amd_tcnn.Encoding and enc.params.numel() == enc.desc.n_entries * enc.desc.n_features
        print(kind, len(rs), "state_dict entries match")
    for kw in PROP_KW:
        torch.manual_seed(1); r = ref_density(aabb=AABB, unbounded=True, **kw)
        torch.manual_seed(1); a = amd_density(aabb=AABB, unbounded=True, **kw)
        assert list(r.state_dict()) == list(a.state_dict())
        assert all(torch.equal(r.state_dict()[k], a.state_dict()[k]) for k in r.state_dict())
    est = RefEstimator(None, None)
    assert isinstance(est, NC.AbstractEstimator) and est.device.type == "cpu"
    fn = get_proposal_requires_grad_fn()
    from emernerf_amd.prop_net import get_proposal_requires_grad_fn as amd_fn
    g = amd_fn()
    assert [fn(s) for s in range(0, 3000, 7)] == [g(s) for s in range(0, 3000, 7)], "proposal schedule"
    print("SWAPS_OK")
''')


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_reference_python_constructs_over_the_four_import_swaps(hip_lib):
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "ref": REF}], capture_output=True, text=True, timeout=600,
                       cwd=REF, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0 and "SWAPS_OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
