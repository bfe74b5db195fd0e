"""Proposal-network estimator with the reference's interface on the HIP kernels.

Drop-in for third_party/nerfacc_prop_net.py: ``PropNetEstimator`` (``sampling`` / ``compute_loss`` /
``update_every_n_steps``; attributes ``optimizer``, ``scheduler``, ``prop_cache`` are touched by the
reference's driver and checkpoint code), ``get_proposal_requires_grad_fn``, ``_transform_stot``.

``sampling`` runs entirely on hand-written HIP: inverse-CDF resampling fused with the s->t transform
(``emer_importance_sample``), ray-point generation + contraction (``emer_ray_points``), proposal
density field (hash grid + MFMA linears) and the transmittance scan (``emer_render_weights_fwd``).
``compute_loss`` (zip-NeRF anti-aliased interlevel loss, reference :181-238) is SURVEY section 8f row N4
("next"): it stays in torch, restated with ``searchsorted`` instead of the reference's O(R*S*m) masks.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .nerfacc_compat import AbstractEstimator, RayIntervals, importance_sampling, searchsorted

logger = logging.getLogger()


def blur_stepfun(x, y, r):
    """nerfacc_prop_net.py:22-34 (zipnerf stepfun blur)."""
    xr, xr_idx = torch.sort(torch.cat([x - r, x + r], dim=-1))
    y1 = (torch.cat([y, torch.zeros_like(y[..., :1])], dim=-1) - torch.cat([torch.zeros_like(y[..., :1]), y], dim=-1)) / (2 * r)
    y2 = torch.cat([y1, -y1], dim=-1).take_along_dim(xr_idx[..., :-1], dim=-1)
    yr = torch.cumsum((xr[..., 1:] - xr[..., :-1]) * torch.cumsum(y2, dim=-1), dim=-1).clamp_min(0)
    yr = torch.cat([torch.zeros_like(yr[..., :1]), yr], dim=-1)
    return xr, yr


def sorted_interp_quad(x, xp, fpdf, fcdf):
    """nerfacc_prop_net.py:37-60, restated with searchsorted.

    The reference builds ``mask = x[..., None, :] >= xp[..., :, None]`` ([R, m, n] booleans plus four
    fp32 temporaries of that shape) and takes masked max / min; with ``xp`` sorted that selects the last
    knot <= x and the first knot > x, i.e. ``k = searchsorted(xp, x, right=True)``: i0 = max(k - 1, 0)
    (falling back to knot 0 when no knot is <= x) and i1 = min(k, m - 1).
    """
    m = xp.shape[-1]
    k = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)
    i0 = (k - 1).clamp(0, m - 1)
    i1 = k.clamp(0, m - 1)
    fcdf0, fcdf1 = fcdf.gather(-1, i0), fcdf.gather(-1, i1)
    fpdf0, fpdf1 = fpdf.gather(-1, i0), fpdf.gather(-1, i1)
    xp0, xp1 = xp.gather(-1, i0), xp.gather(-1, i1)
    offset = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fcdf0 + (x - xp0) * (fpdf0 + fpdf1 * offset + fpdf0 * (1 - offset)) / 2


class PropNetEstimator(AbstractEstimator):
    """nerfacc_prop_net.py:63-277."""

    def __init__(self, optimizer: Optional[torch.optim.Optimizer] = None, scheduler=None,
                 enable_anti_aliasing_loss: Optional[bool] = True,
                 anti_aliasing_pulse_width: Optional[List[float]] = [0.03, 0.003]) -> None:
        super().__init__()
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.prop_cache: List = []
        self.enable_anti_aliasing_loss = enable_anti_aliasing_loss
        self.pulse_width = anti_aliasing_pulse_width
        # Stratified jitter: one U(0,1) per ray per resampling round.  Upstream draws it inside the CUDA
        # kernel from torch's Philox state (irreproducible); here it is an explicit tensor so the sampler
        # stays bit-exact testable.  Tests replace this hook to replay the oracle's draws.
        self.jitter_fn: Callable[[int, torch.device], Tensor] = lambda n, dev: torch.rand(n, device=dev)

    @torch.no_grad()
    def sampling(self, prop_sigma_fns: List[Callable], prop_samples: List[int], num_samples: int, n_rays: int,
                 near_plane: float, far_plane: float, sampling_type: str = "uniform_lindisp",
                 stratified: bool = False, requires_grad: bool = False) -> Tuple[Tensor, Tensor]:
        """:89-179.  Returns (t_starts, t_ends), both (n_rays, num_samples)."""
        assert len(prop_sigma_fns) == len(prop_samples), \
            "The number of proposal networks and the number of samples should be the same."
        dev = self.device
        cdfs = torch.cat([torch.zeros((n_rays, 1), device=dev), torch.ones((n_rays, 1), device=dev)], dim=-1)
        s_vals = cdfs
        stot = (float(near_plane), float(far_plane), sampling_type)
        for i, (level_fn, level_samples) in enumerate(zip(prop_sigma_fns, prop_samples)):
            jitter = self.jitter_fn(n_rays, dev) if stratified else None
            s_vals, t_starts, t_ends = ops.importance_sample(s_vals, cdfs, level_samples, jitter, stot=stot, intervals=True)
            with torch.set_grad_enabled(requires_grad):
                sigmas = level_fn(t_starts, t_ends)["density"].squeeze(-1)
                assert sigmas.shape == t_starts.shape
                _, _, _, cdfs, _ = ops.render_weights(t_starts, t_ends, sigmas)  # cdfs = 1 - [T, 0]
                if requires_grad:
                    self.prop_cache.append((RayIntervals(vals=s_vals), cdfs, i))
            cdfs = cdfs.detach() if not requires_grad else cdfs
        jitter = self.jitter_fn(n_rays, dev) if stratified else None
        s_vals, t_starts, t_ends = ops.importance_sample(s_vals, cdfs.detach(), num_samples, jitter, stot=stot, intervals=True)
        if requires_grad:
            self.prop_cache.append((RayIntervals(vals=s_vals), None, None))
        return t_starts, t_ends

    @torch.enable_grad()
    def compute_loss(self, trans: Tensor, loss_scaler: float = 1.0) -> Tensor:
        """:181-238."""
        if len(self.prop_cache) == 0:
            return torch.zeros((), device=self.device)
        intervals, _, _ = self.prop_cache.pop()
        cdfs = 1.0 - torch.cat([trans, torch.zeros_like(trans[..., :1])], dim=-1)
        cdfs = cdfs.detach()
        loss = 0.0
        if self.enable_anti_aliasing_loss:
            w_normalize = (cdfs[..., 1:] - cdfs[..., :-1]) / (intervals.vals[..., 1:] - intervals.vals[..., :-1])
            c1, w1 = blur_stepfun(intervals.vals, w_normalize, self.pulse_width[0])
            c2, w2 = blur_stepfun(intervals.vals, w_normalize, self.pulse_width[1])
            area1 = 0.5 * (w1[..., 1:] + w1[..., :-1]) * (c1[..., 1:] - c1[..., :-1])
            area2 = 0.5 * (w2[..., 1:] + w2[..., :-1]) * (c2[..., 1:] - c2[..., :-1])
            cdfs1 = torch.cat([torch.zeros_like(area1[..., :1]), torch.cumsum(area1, dim=-1)], dim=-1)
            cdfs2 = torch.cat([torch.zeros_like(area2[..., :1]), torch.cumsum(area2, dim=-1)], dim=-1)
            cs, ws, _cdfs = [c1, c2], [w1, w2], [cdfs1, cdfs2]
            while self.prop_cache:
                prop_intervals, prop_cdfs, prop_id = self.prop_cache.pop()
                wp = prop_cdfs[..., 1:] - prop_cdfs[..., :-1]
                cdf_interp = sorted_interp_quad(prop_intervals.vals, cs[prop_id], ws[prop_id], _cdfs[prop_id])
                w_s = torch.diff(cdf_interp, dim=-1)
                loss += ((w_s - wp).clamp_min(0) ** 2 / (wp + 1e-5)).mean()
        else:
            while self.prop_cache:
                prop_intervals, prop_cdfs, _ = self.prop_cache.pop()
                loss += _pdf_loss(intervals, cdfs, prop_intervals, prop_cdfs).mean()
        return loss * loss_scaler

    @torch.enable_grad()
    def update_every_n_steps(self, trans: Tensor, requires_grad: bool = False, loss_scaler: float = 1.0) -> float:
        """:240-262."""
        if requires_grad:
            return self._update(trans=trans, loss_scaler=loss_scaler)
        if self.scheduler is not None:
            self.scheduler.step()
        return 0.0

    @torch.enable_grad()
    def _update(self, trans: Tensor, loss_scaler: float = 1.0) -> float:
        """:264-277."""
        assert len(self.prop_cache) > 0
        assert self.optimizer is not None, "No optimizer is provided."
        loss = self.compute_loss(trans, loss_scaler)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.item()


def get_proposal_requires_grad_fn(target: float = 5.0, num_steps: int = 1000) -> Callable:
    """nerfacc_prop_net.py:280-296."""
    schedule = lambda s: min(s / num_steps, 1.0) * target  # noqa: E731
    steps_since_last_grad = 0

    def proposal_requires_grad_fn(step: int) -> bool:
        nonlocal steps_since_last_grad
        target_steps_since_last_grad = schedule(step)
        requires_grad = steps_since_last_grad > target_steps_since_last_grad
        if requires_grad:
            steps_since_last_grad = 0
        steps_since_last_grad += 1
        return requires_grad

    return proposal_requires_grad_fn


def _transform_stot(transform_type: str, s_vals: Tensor, t_min, t_max) -> Tensor:
    """nerfacc_prop_net.py:317-339 for scalar near/far planes (what render_rays passes)."""
    if isinstance(t_min, Tensor) and t_min.dim() > 0 or isinstance(t_max, Tensor) and t_max.dim() > 0:
        raise NotImplementedError("per-ray near/far planes are not used by EmerNeRF (render_utils.py:363-364)")
    return ops.stot(s_vals, float(t_min), float(t_max), transform_type)


def _pdf_loss(segments_query: RayIntervals, cdfs_query: Tensor, segments_key: RayIntervals, cdfs_key: Tensor,
              eps: float = 1e-7) -> Tensor:
    """nerfacc_prop_net.py:342-362 (batched branch; only used when enable_anti_aliasing_level_loss is False)."""
    ids_left, ids_right = searchsorted(segments_key, segments_query)
    w = cdfs_query[..., 1:] - cdfs_query[..., :-1]
    ids_left = ids_left[..., :-1]
    ids_right = ids_right[..., 1:]
    w_outer = cdfs_key.gather(-1, ids_right) - cdfs_key.gather(-1, ids_left)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + eps)
