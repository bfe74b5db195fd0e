"""The seven nerfacc symbols EmerNeRF imports, on the HIP kernels (dense batched (R,S) mode only).

Reference import sites: radiance_fields/render_utils.py:4-8, loss/base.py:7,
third_party/nerfacc_prop_net.py:11-14.  A reference checkout can point those imports at this module
(see INTEGRATION.md).  Packed / ``ray_indices`` mode is never used by EmerNeRF and raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops


@dataclass
class RayIntervals:
    """nerfacc.data_specs.RayIntervals; batched mode: ``vals`` is [n_rays, n_edges]."""
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None
    is_left: Optional[Tensor] = None
    is_right: Optional[Tensor] = None

    @property
    def device(self) -> torch.device:
        return self.vals.device


@dataclass
class RaySamples:
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None

    @property
    def device(self) -> torch.device:
        return self.vals.device


class AbstractEstimator(torch.nn.Module):
    """nerfacc.estimators.base.AbstractEstimator: exposes ``.device`` through an empty buffer."""

    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("_dummy", torch.empty(0), persistent=False)

    @property
    def device(self) -> torch.device:
        return self._dummy.device

    def sampling(self, *args, **kwargs):
        raise NotImplementedError

    def update_every_n_steps(self, *args, **kwargs) -> None:
        raise NotImplementedError


def _dense_only(**kw):
    for k, v in kw.items():
        if v is not None:
            raise NotImplementedError(f"nerfacc packed mode ({k}=...) is not used by EmerNeRF and not implemented")


def render_transmittance_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info=None,
                                      ray_indices=None, n_rays=None, prefix_trans=None) -> Tuple[Tensor, Tensor]:
    """-> (trans, alphas), both (R,S).  Call sites: render_utils.py:73, nerfacc_prop_net.py:165."""
    _dense_only(packed_info=packed_info, ray_indices=ray_indices, prefix_trans=prefix_trans)
    _, trans, alphas, _, _ = ops.render_weights(t_starts, t_ends, sigmas)
    return trans, alphas


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info=None, ray_indices=None,
                               n_rays=None, prefix_trans=None) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (weights, trans, alphas).  Call site: render_utils.py:35."""
    _dense_only(packed_info=packed_info, ray_indices=ray_indices, prefix_trans=prefix_trans)
    w, trans, alphas, _, _ = ops.render_weights(t_starts, t_ends, sigmas)
    return w, trans, alphas


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices=None, n_rays=None) -> Tensor:
    """sum_s w[..., s, None] * values[..., s, :] -> (R, C) (or (R, 1) when values is None)."""
    _dense_only(ray_indices=ray_indices)
    return ops.accumulate_along_rays(weights, values)


def importance_sampling(intervals: RayIntervals, cdfs: Tensor, n_intervals_per_ray: int, stratified: bool = False,
                        jitter: Optional[Tensor] = None) -> Tuple[RayIntervals, RaySamples]:
    """nerfacc.pdf.importance_sampling (batched).  ``jitter``: the per-ray U(0,1) used when stratified
    (drawn with torch.rand when not supplied).  Frozen spec: SURVEY.md Appendix A.2."""
    _dense_only(packed_info=intervals.packed_info)
    if stratified and jitter is None:
        jitter = torch.rand(cdfs.shape[0], device=cdfs.device)
    s, _ = ops.importance_sample(intervals.vals, cdfs, n_intervals_per_ray, jitter if stratified else None)
    return RayIntervals(vals=s), RaySamples(vals=(s[..., :-1] + s[..., 1:]) * 0.5)


def searchsorted(sorted_sequence: RayIntervals, values: RayIntervals) -> Tuple[Tensor, Tensor]:
    """nerfacc.pdf.searchsorted (batched): bracketing indices of each value among the sorted edges.
    Only reached through _pdf_loss when enable_anti_aliasing_level_loss is False (non-default); index
    search over <= 129 edges per ray, served by torch.searchsorted."""
    _dense_only(packed_info=sorted_sequence.packed_info)
    n = sorted_sequence.vals.shape[-1]
    ids_right = torch.searchsorted(sorted_sequence.vals.contiguous(), values.vals.contiguous(), right=True)
    ids_left = (ids_right - 1).clamp(0, n - 1)
    return ids_left, ids_right.clamp(0, n - 1)
