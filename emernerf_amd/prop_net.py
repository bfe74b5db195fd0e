"""Proposal-network estimator with the reference's interface, entirely on the HIP kernels.

Drop-in for third_party/nerfacc_prop_net.py: ``PropNetEstimator`` (``sampling`` / ``compute_loss`` /
``update_every_n_steps``; the attributes ``optimizer``, ``scheduler``, ``prop_cache`` are touched by the reference's
driver and checkpoint code), ``get_proposal_requires_grad_fn``, ``_transform_stot``.

  * ``sampling``: inverse-CDF resampling fused with the s->t transform (``emer_importance_sample``), ray points +
    contraction (``emer_ray_points``), the proposal density field (hash grid + fused MLP) and the transmittance scan
    (``emer_render_weights_fwd``).
  * ``compute_loss`` (reference :181-238): ONE launch per cached proposal level (``emer_prop_loss``: merge of the two
    shifted edge lists instead of a sort, wave scans, binary-search interpolation, hinge loss and its gradient) -- the
    reference's helper functions blur_stepfun / sorted_interp_quad / _pdf_loss (:22-60, :342-362) have no Python
    counterpart here, they are stages of that kernel (csrc/proploss.hip).
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops
from .nerfacc_compat import AbstractEstimator, RayIntervals

logger = logging.getLogger()


class PropNetEstimator(AbstractEstimator):
    """nerfacc_prop_net.py:63-277 (same constructor arguments and public attributes)."""

    def __init__(self, optimizer: Optional[torch.optim.Optimizer] = None, scheduler=None,
                 enable_anti_aliasing_loss: Optional[bool] = True,
                 anti_aliasing_pulse_width: Optional[List[float]] = [0.03, 0.003]) -> None:
        super().__init__()
        self.optimizer, self.scheduler = optimizer, scheduler
        self.prop_cache: List = []
        self.enable_anti_aliasing_loss = enable_anti_aliasing_loss
        self.pulse_width = anti_aliasing_pulse_width
        # Stratified jitter: one U(0,1) per ray per resampling round.  Upstream draws it inside the CUDA kernel from
        # torch's Philox state (irreproducible); here it is an explicit tensor, so the sampler stays bit-exact testable.
        # Tests replace this hook to replay the oracle's draws.  None (default): ONE torch.rand call per sampling() for all of
        # its resampling rounds.
        self.jitter_fn: Optional[Callable[[int, torch.device], Tensor]] = None
        # [n_rays, 2] edges (0, 1) of the trivial level-0 histogram, one persistent constant per (n_rays, device): a captured
        # hipGraph has the tensor's address baked into its first importance_sample launch, so an entry is NEVER freed or
        # replaced (an eval chunk or a lidar step with another ray count adds an entry; 8 bytes per ray)
        self._unit = {}

    # ------------------------------------------------------------------------------------------ sampling (:89-179)
    @torch.no_grad()
    def sampling(self, prop_sigma_fns: List[Callable], prop_samples: List[int], num_samples: int, n_rays: int,
                 near_plane: float, far_plane: float, sampling_type: str = "uniform_lindisp",
                 stratified: bool = False, requires_grad: bool = False, ray_geometry=None) -> Tuple[Tensor, Tensor]:
        """Returns (t_starts, t_ends), both (n_rays, num_samples).

        ``ray_geometry`` [r5, optional; not in the reference's signature]: (origins [R,3], dirs [R,3], [(aabb, unbounded, want_positions)
        per proposal level ... and for the final samples]).  The sampler's launch then also computes the sample points of the intervals it
        produces (``ops.importance_sample(points=...)``) and hands them to the consumer as ``t_starts._emer_points = (normed, positions,
        aabb, unbounded)``, so ``sigma_fn`` / ``query_fn`` need no ``ray_points`` launch of their own."""
        assert len(prop_sigma_fns) == len(prop_samples), \
            "The number of proposal networks and the number of samples should be the same."
        dev = self.device
        planes = (float(near_plane), float(far_plane), sampling_type)
        # level 0 resamples the trivial histogram on [0, 1]
        key = (int(n_rays), str(dev))
        unit = self._unit.get(key)
        if unit is None:
            if torch.cuda.is_available() and dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                # first sight of this ray count inside a capture: allocate from the graph's pool (kept alive by the graph)
                unit = torch.arange(2, device=dev, dtype=torch.float32).expand(n_rays, 2).contiguous()
            else:
                unit = self._unit[key] = torch.arange(2, device=dev, dtype=torch.float32).expand(n_rays, 2).contiguous()
        edges = cdfs = unit  # read-only constant: no launch per call
        jitters = None
        if stratified and self.jitter_fn is None:
            jitters = iter(torch.rand((len(prop_samples) + 1, n_rays), device=dev).unbind(0))
        draw = (lambda: next(jitters)) if jitters is not None else (lambda: self.jitter_fn(n_rays, dev))
        def sample(edges_in, cdfs_in, n_out, u, slot):
            geo = None
            if ray_geometry is not None and ops.SAMPLE_POINTS and 2 * edges_in.shape[-1] + n_out + 1 <= ops.sample_points_capacity():
                geo = ray_geometry[2][slot]
            if geo is None:
                return ops.importance_sample(edges_in, cdfs_in, n_out, u, stot=planes, intervals=True)
            aabb, unbounded, want_pos = geo
            e, a, b, normed, pos = ops.importance_sample(edges_in, cdfs_in, n_out, u, stot=planes, intervals=True,
                                                         points=(ray_geometry[0], ray_geometry[1], aabb, unbounded, want_pos))
            a._emer_points = (normed, pos, aabb, bool(unbounded))
            return e, a, b

        for level, (sigma_fn, n_level) in enumerate(zip(prop_sigma_fns, prop_samples)):
            u = draw() if stratified else None
            edges, t0, t1 = sample(edges, cdfs, n_level, u, level)
            with torch.set_grad_enabled(requires_grad):
                sigma = sigma_fn(t0, t1)["density"].squeeze(-1)
                assert sigma.shape == t0.shape
                cdfs = ops.render_weights(t0, t1, sigma)[3]  # 1 - [T, 0]  (:165-168)
            if requires_grad:
                self.prop_cache.append((RayIntervals(vals=edges), cdfs, level))
            else:
                cdfs = cdfs.detach()
        u = draw() if stratified else None
        edges, t0, t1 = sample(edges, cdfs.detach(), num_samples, u, len(prop_samples))
        if requires_grad:
            self.prop_cache.append((RayIntervals(vals=edges), None, None))
        return t0, t1

    # ------------------------------------------------------------------------------------- supervision (:181-238)
    @torch.enable_grad()
    def compute_loss(self, trans: Tensor, loss_scaler: float = 1.0) -> Tensor:
        """Interlevel loss of every cached proposal level against the final samples' histogram
        (cdf = 1 - [trans, 0], detached), summed over levels, times ``loss_scaler``.  The cache is consumed."""
        if not self.prop_cache:
            return torch.zeros((), device=self.device)
        final, _, _ = self.prop_cache.pop()
        s_final = final.vals
        R, n = trans.shape
        anti = bool(self.enable_anti_aliasing_loss)
        trans_ng = trans.detach()  # the final histogram is a target, not a variable (:186-187)
        total = None
        while self.prop_cache:
            level_edges, level_cdfs, level = self.prop_cache.pop()
            m = level_cdfs.shape[-1] - 1
            # mean over (R, m) proposal intervals (anti-aliased, :226) or over (R, n) final intervals (_pdf_loss, :230)
            count = R * (m if anti else n)
            pulse = float(self.pulse_width[level]) if anti else 0.0
            term = ops.prop_level_loss(s_final, trans_ng, level_edges.vals, level_cdfs, pulse, anti, float(loss_scaler) / count)
            total = term if total is None else total + term
        return total if total is not None else torch.zeros((), device=self.device)

    # ------------------------------------------------------------------------------ reference driver hooks (:240-277)
    @torch.enable_grad()
    def update_every_n_steps(self, trans: Tensor, requires_grad: bool = False, loss_scaler: float = 1.0) -> float:
        """Train the proposal nets on the steps that cached their levels; otherwise only advance the LR schedule."""
        if not requires_grad:
            if self.scheduler is not None:
                self.scheduler.step()
            return 0.0
        return self._update(trans=trans, loss_scaler=loss_scaler)

    @torch.enable_grad()
    def _update(self, trans: Tensor, loss_scaler: float = 1.0) -> float:
        assert self.optimizer is not None, "No optimizer is provided."
        assert len(self.prop_cache) > 0
        loss = self.compute_loss(trans, loss_scaler)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.item()


class _ProposalGradSchedule:
    """nerfacc_prop_net.py:280-296: train the proposal nets whenever more steps have passed since their last update
    than ``target * min(step / num_steps, 1)`` -- every step at the start, one step in ``target + 1`` at steady state."""

    def __init__(self, target: float, num_steps: int) -> None:
        self.target, self.num_steps = target, num_steps
        self.since_last = 0

    def __call__(self, step: int) -> bool:
        wait = self.target * min(step / self.num_steps, 1.0)
        fire = self.since_last > wait
        self.since_last = 1 if fire else self.since_last + 1
        return fire


def get_proposal_requires_grad_fn(target: float = 5.0, num_steps: int = 1000) -> Callable[[int], bool]:
    return _ProposalGradSchedule(target, num_steps)


def _transform_stot(transform_type: str, s_vals: Tensor, t_min, t_max) -> Tensor:
    """nerfacc_prop_net.py:317-339 for scalar near/far planes (what render_rays passes)."""
    per_ray = [v for v in (t_min, t_max) if isinstance(v, Tensor) and v.dim() > 0]
    if per_ray:
        raise NotImplementedError("per-ray near/far planes are not used by EmerNeRF (render_utils.py:363-364)")
    if transform_type not in ops.STOT_TYPES:
        raise ValueError(f"Unknown transform_type: {transform_type}")
    return ops.stot(s_vals, float(t_min), float(t_max), transform_type)
