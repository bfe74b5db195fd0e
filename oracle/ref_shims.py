"""ORACLE -- test infrastructure.  Import shims that let /root/reference's own Python run UNMODIFIED on CPU.

The reference needs three packages that are not installed here: the vendored
``third_party.tcnn_modules`` refuses to import without CUDA, ``nerfacc`` and ``omegaconf`` are absent.
``install()`` registers stand-ins in ``sys.modules`` built on the CPU oracle (oracle/oracle.py), then
the reference's ``radiance_fields``, ``third_party.nerfacc_prop_net`` and ``loss`` import and run as
written.  Used only by tests/golden/make_golden.py (in the build container, where /root/reference
exists) to pin the Python layer of the path; never imported by the product.

Randomness the reference draws internally is captured so a second implementation can replay it:
  * ``JITTER_LOG``: every per-ray U(0,1) tensor ``importance_sampling(stratified=True)`` used;
  * temporal-aggregation noise: make_golden patches ``torch.rand_like`` around the call.
"""
from __future__ import annotations

import sys
import types
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor

from . import oracle as O

REFERENCE_ROOT = "/root/reference"
JITTER_LOG: List[Tensor] = []


# ------------------------------------------------------------------ third_party.tcnn_modules
class Encoding(torch.nn.Module):
    """Stand-in for tcnn.Encoding (third_party/tcnn_modules.py:375-423) for otype HashGrid."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        if encoding_config["otype"] != "HashGrid":
            raise NotImplementedError(encoding_config["otype"])
        self.n_input_dims = n_input_dims
        self.encoding_config = encoding_config
        self.meta = O.grid_meta(
            n_input_dims, encoding_config["n_levels"], encoding_config["n_features_per_level"],
            encoding_config["log2_hashmap_size"], encoding_config["base_resolution"],
            encoding_config["per_level_scale"])
        self.n_output_dims = self.meta.n_output_dims
        g = torch.Generator().manual_seed(seed)
        # tcnn initial_params: U(-1e-4, 1e-4) (the pcg32 stream is not reproducible)
        init = (torch.rand(self.meta.n_params, generator=g) * 2 - 1) * 1e-4
        self.params = torch.nn.Parameter(init)
        self.dtype = dtype or torch.float32

    def forward(self, x: Tensor) -> Tensor:
        return O.hashgrid(x, self.params, self.meta)


def _tcnn_module():
    m = types.ModuleType("third_party.tcnn_modules")
    m.Encoding = Encoding
    return m


# ----------------------------------------------------------------------------------- nerfacc
@dataclass
class RayIntervals:
    """nerfacc.data_specs.RayIntervals (batched mode: vals [R, n_edges])."""
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None
    is_left: Optional[Tensor] = None
    is_right: Optional[Tensor] = None

    @property
    def device(self):
        return self.vals.device


class AbstractEstimator(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("_dummy", torch.empty(0), persistent=False)

    @property
    def device(self):
        return self._dummy.device


def render_transmittance_from_density(t_starts, t_ends, sigmas, **_):
    sdt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sdt)
    cum = torch.cumsum(sdt, dim=-1)
    excl = torch.cat([torch.zeros_like(cum[..., :1]), cum[..., :-1]], dim=-1)
    return torch.exp(-excl), alphas


def render_weight_from_density(t_starts, t_ends, sigmas, **_):
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas)
    return trans * alphas, trans, alphas


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    if values is None:
        return weights.sum(dim=-1, keepdim=True)
    return (weights[..., None] * values).sum(dim=-2)


def importance_sampling(intervals: RayIntervals, cdfs: Tensor, n_intervals_per_ray: int, stratified: bool = False):
    jitter = None
    if stratified:
        jitter = torch.rand(cdfs.shape[0])
        JITTER_LOG.append(jitter.clone())
    out = O.importance_sample(intervals.vals.detach(), cdfs.detach(), n_intervals_per_ray, jitter)
    return RayIntervals(vals=torch.from_numpy(out)), None


def searchsorted(sorted_sequence: RayIntervals, values: RayIntervals):
    ids_right = torch.searchsorted(sorted_sequence.vals.contiguous(), values.vals.contiguous(), right=True)
    ids_left = (ids_right - 1).clamp(0, sorted_sequence.vals.shape[-1] - 1)
    ids_right = ids_right.clamp(0, sorted_sequence.vals.shape[-1] - 1)
    return ids_left, ids_right


def _nerfacc_modules():
    root = types.ModuleType("nerfacc")
    root.accumulate_along_rays = accumulate_along_rays
    root.render_transmittance_from_density = render_transmittance_from_density
    root.render_weight_from_density = render_weight_from_density
    ds = types.ModuleType("nerfacc.data_specs"); ds.RayIntervals = RayIntervals
    est = types.ModuleType("nerfacc.estimators"); base = types.ModuleType("nerfacc.estimators.base")
    base.AbstractEstimator = AbstractEstimator
    pdf = types.ModuleType("nerfacc.pdf"); pdf.importance_sampling = importance_sampling; pdf.searchsorted = searchsorted
    vol = types.ModuleType("nerfacc.volrend"); vol.render_transmittance_from_density = render_transmittance_from_density
    vol.render_weight_from_density = render_weight_from_density; vol.accumulate_along_rays = accumulate_along_rays
    root.data_specs, root.estimators, root.pdf, root.volrend = ds, est, pdf, vol
    est.base = base
    return {"nerfacc": root, "nerfacc.data_specs": ds, "nerfacc.estimators": est, "nerfacc.estimators.base": base,
            "nerfacc.pdf": pdf, "nerfacc.volrend": vol}


def install(reference_root: str = REFERENCE_ROOT):
    """Register the shims and put the reference on sys.path.  Idempotent."""
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    for name, mod in _nerfacc_modules().items():
        sys.modules[name] = mod
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules["omegaconf"] = oc
    import third_party  # the reference's (empty) package
    tm = _tcnn_module()
    sys.modules["third_party.tcnn_modules"] = tm
    third_party.tcnn_modules = tm  # encodings.py:9 does `import third_party.tcnn_modules as tcnn`
    return tm


def ns(**kw):
    """Nested SimpleNamespace config (render_utils.py reads cfg by attribute only)."""
    return types.SimpleNamespace(**{k: ns(**v) if isinstance(v, dict) else v for k, v in kw.items()})
