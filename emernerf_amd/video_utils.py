"""Evaluation render loop on the HIP kernels.

Mirrors the rendering half of the reference's radiance_fields/video_utils.py: ``render_pixels`` (:50-106) and ``render``
(:109-468) -- iterate over the images of a split, call ``render_rays`` in eval mode with ``return_decomposition`` on
16 384-ray chunks (cfg.render.render_chunk_size), and collect per-image outputs under the reference's key names (rgbs,
gt_rgbs, depths, opacities, static_* / dynamic_* decomposition, shadow_reduced_static_rgbs, flows, sky_masks ...).

What changed underneath: an image's rays come from ``PixelSource.get_render_rays`` (one gather kernel), all chunks of an
image are rendered before any result leaves the GPU, an image's results leave it as ONE asynchronous transfer that overlaps
the next image's rendering (the reference interleaves a blocking ``.cpu().numpy()`` per key with the rendering), and PSNR is
computed on the device.  Out of scope here, as in SURVEY section 2: video encoding, SSIM (scikit-image), DINO-feature PCA colouring.
"""
from __future__ import annotations

import logging
import time
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .prop_net import PropNetEstimator
from .radiance_field import DensityField, RadianceField
from .render_utils import render_rays

logger = logging.getLogger()

# results key -> (list name in the returned dict)  (video_utils.py:118-146)
_COLLECT = {"rgb": "rgbs", "static_rgb": "static_rgbs", "shadow_reduced_static_rgb": "shadow_reduced_static_rgbs",
            "shadow_only_static_rgb": "shadow_only_static_rgbs", "depth": "depths", "opacity": "opacities",
            "static_depth": "static_depths", "static_opacity": "static_opacities", "dynamic_depth": "dynamic_depths",
            "dynamic_opacity": "dynamic_opacities", "forward_flow": "forward_flows", "backward_flow": "backward_flows"}


def compute_psnr(prediction: Tensor, target: Tensor) -> float:
    """datasets/metrics.py:31-46."""
    return float(-10.0 * torch.log10(torch.nn.functional.mse_loss(prediction, target)))


def _pack_to_host(keep: Dict[str, Tensor], pinned: Dict[int, Tensor], slot: int):
    """Start the transfer of one image's results: (names, shapes, dtypes, host buffer, event)."""
    names = list(keep)
    parts = [keep[n].squeeze() for n in names]
    flat = torch.cat([p.reshape(-1).to(torch.float32) for p in parts]) if parts else None
    if flat is None:
        return names, [], [], None, None
    buf = pinned.get(slot)
    if buf is None or buf.numel() < flat.numel():
        buf = pinned[slot] = torch.empty((flat.numel(),), dtype=torch.float32, pin_memory=True)
    host = buf[:flat.numel()]
    host.copy_(flat, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return names, [tuple(p.shape) for p in parts], [p.dtype for p in parts], host, ev


def _unpack(item, out: Dict[str, list]) -> None:
    names, shapes, dtypes, host, ev = item
    if host is None:
        return
    ev.synchronize()
    arr, o = host.numpy(), 0
    for name, shape, dt in zip(names, shapes, dtypes):
        n = int(np.prod(shape)) if len(shape) else 1
        a = arr[o:o + n].reshape(shape).copy()  # the pinned buffer is reused two images later
        out[name].append(a if dt == torch.float32 else a.astype(torch.empty((), dtype=dt).numpy().dtype))
        o += n


def render_pixels(cfg, model: RadianceField, proposal_estimator: PropNetEstimator, dataset,
                  proposal_networks: Optional[List[DensityField]] = None, compute_metrics: bool = False,
                  vis_indices: Optional[List[int]] = None, return_decomposition: bool = True) -> Dict[str, list]:
    """video_utils.py:50-106.  ``dataset``: anything with ``__len__`` / ``__getitem__`` returning image-shaped ray dicts
    (``PixelSource`` here; the reference's SplitWrapper there)."""
    model.eval()
    for p in proposal_networks or []:
        p.eval()
    if proposal_estimator is not None:
        proposal_estimator.eval()

    def render_func(data_dict):
        return render_rays(radiance_field=model, proposal_estimator=proposal_estimator, proposal_networks=proposal_networks,
                           data_dict=data_dict, cfg=cfg, return_decomposition=return_decomposition)

    results = render(dataset, render_func, model=model, compute_metrics=compute_metrics, vis_indices=vis_indices)
    if compute_metrics:
        n = len(dataset) if vis_indices is None else len(vis_indices)
        logger.info(f"Eval over {n} images:\n\tPSNR: {results['psnr']:.4f}")
    return results


def render(dataset, render_func: Callable, model: Optional[RadianceField] = None, compute_metrics: bool = False,
           vis_indices: Optional[List[int]] = None) -> Dict[str, list]:
    """video_utils.py:109-468: the reference's result dictionary key for key -- the nine lists it always returns (possibly
    empty), the conditional ones (gt_rgbs, gt_sky_masks, shadow_*, forward_flows / backward_flows, median_depths), the
    scalars (psnr; ssim and the feature / masked metrics are -1: SSIM and the DINO-feature PCA colouring are out of scope,
    SURVEY section 2).  ``forward_flows`` / ``backward_flows`` hold the rendered flow (the reference stores its colour-wheel
    visualisation).  Checked against a recording of the reference's own loop: tests/golden/render_pixels_*.npz."""
    always = ["rgbs", "static_rgbs", "dynamic_rgbs", "depths", "opacities", "static_depths", "static_opacities", "dynamic_depths",
              "dynamic_opacities"]
    out: Dict[str, list] = {v: [] for v in _COLLECT.values()}
    out.update({"gt_rgbs": [], "dynamic_rgbs": [], "median_depths": [], "gt_sky_masks": []})
    psnrs: List[Tensor] = []
    pending: list = []
    pinned: Dict[int, Tensor] = {}
    n_rays, n_images, t0 = 0, 0, time.perf_counter()
    green = None
    with torch.no_grad():
        indices = vis_indices if vis_indices is not None else range(len(dataset))
        for i in indices:
            data = {k: (v.cuda(non_blocking=True) if isinstance(v, Tensor) and not v.is_cuda else v) for k, v in dataset[i].items()}
            res = render_func(data)
            n_rays += int(data["origins"].numel() // 3)
            keep: Dict[str, Tensor] = {}
            for k, name in _COLLECT.items():
                if k in res:
                    keep[name] = res[k]
            if "dynamic_rgb" in res:  # green-screen blend for visualisation (:169-177)
                if green is None:
                    green = torch.tensor([0.0, 177.0, 64.0], device=res["dynamic_rgb"].device) / 255.0
                keep["dynamic_rgbs"] = res["dynamic_rgb"] * res["dynamic_opacity"] + green * (1 - res["dynamic_opacity"])
            if "dynamic_depth" not in res and "median_depth" in res:
                keep["median_depths"] = res["median_depth"]
            if "pixels" in data:
                keep["gt_rgbs"] = data["pixels"]
            if "sky_masks" in data:
                keep["gt_sky_masks"] = data["sky_masks"]
            if compute_metrics and "pixels" in data:  # stays on the device until the loop is over (no sync per image)
                psnrs.append(-10.0 * torch.log10(torch.nn.functional.mse_loss(res["rgb"], data["pixels"])))
            # ONE device->host transfer per image: every kept tensor packed into a flat buffer, copied asynchronously into
            # pinned memory and unpacked while the NEXT image renders (the reference interleaves a blocking .cpu().numpy() per
            # key with the rendering; on a slow host that halves the loop's throughput)
            pending.append(_pack_to_host(keep, pinned, n_images & 1))  # two alternating pinned buffers
            n_images += 1
            if len(pending) > 1:
                _unpack(pending.pop(0), out)
        while pending:
            _unpack(pending.pop(0), out)
    torch.cuda.synchronize()
    out = {k: v for k, v in out.items() if len(v) > 0 or k in always}
    dt = time.perf_counter() - t0
    out["render_rays_per_s"] = n_rays / dt if dt > 0 else float("nan")
    out["psnr"] = (float(torch.stack(psnrs).mean()) if psnrs else -1.0) if compute_metrics else -1
    for k in ("ssim", "feat_psnr", "masked_psnr", "masked_ssim", "masked_feat_psnr"):
        out[k] = -1
    return out
