"""render_rays / rendering with the reference's interface on the HIP kernels.

Mirrors radiance_fields/render_utils.py: same signatures, same result keys and shapes (SURVEY.md
section 8b "verified output contract"), including the reference's quirks that callers depend on
(``extras`` from the last chunk only, :375-388; ``density`` popped into ``extras``, :384;
``shadow_ratio`` accumulated squared, :165-168).

What changed underneath:
  * sample points come from one fused kernel (``emer_ray_points``: o + d*(t0+t1)/2 -> contraction),
    not from broadcasting every per-ray key to (R,S) with ``repeat_interleave`` (:319-336); per-ray
    keys reach the field as stride-0 expanded views;
  * transmittance / alpha / weights / per-ray sums / median depth come from one wave-per-ray scan
    kernel, channel accumulation from ``emer_accumulate_*``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .nerfacc_compat import accumulate_along_rays
from .prop_net import PropNetEstimator
from .radiance_field import DensityField, RadianceField


def render_weights_opacity_depth_from_density(t_starts: Tensor, t_ends: Tensor, density: Tensor):
    """render_utils.py:19-45."""
    weights, _, _, _, stats = ops.render_weights(t_starts, t_ends, density)
    opacities, depths, _, _ = ops.ray_epilogue(stats)
    return weights, opacities, depths


def rendering(t_starts: Tensor, t_ends: Tensor, query_fn: Optional[Callable] = None,
              return_decomposition: bool = False) -> Dict[str, Tensor]:
    """render_utils.py:48-287."""
    results = query_fn(t_starts, t_ends)
    density = results["density"].squeeze(-1)
    if (ops.FUSE_COMPOSITE and density.is_cuda and density.dim() == 2 and "static_density" not in results and "static_rgb" not in results
            and "dino_feat" not in results and "static_dino_feat" not in results
            and ("rgb" not in results or results["rgb"].shape[-1] == 3)):
        # [r4] one density, one colour (or none: the lidar step's density-only render): scan, accumulation and per-ray epilogue in one
        # launch each way (:73-122,158-159,217-220) -- same values as the general path below
        rgb = results.get("rgb")
        weights, trans, t_vals, t_dist, opacities, depths, median_depth, rgb_out = ops.composite_rgb(
            t_starts, t_ends, density, rgb, results.get("rgb_sky") if rgb is not None else None)
        extras = {"weights": weights, "trans": trans, "t_vals": t_vals, "t_dist": t_dist}
        for k in ["forward_flow", "backward_flow", "forward_pred_backward_flow", "backward_pred_forward_flow"]:
            if k in results:
                extras[k] = results[k]
        results_dict = {"density": density, "depth": depths, "opacity": opacities, "median_depth": median_depth}
        if rgb_out is not None:
            results_dict["rgb"] = rgb_out
        results_dict["extras"] = extras
        return results_dict
    # one scan kernel: weights, transmittance, the per-ray sums behind opacity / depth / median depth (:102-122) and the
    # t_vals / t_dist extras (:84-85)
    weights, trans, _, _, stats, t_vals, t_dist = ops.render_weights(t_starts, t_ends, density, want_t=True)
    extras = {"weights": weights, "trans": trans, "t_vals": t_vals, "t_dist": t_dist}
    for k in ["forward_flow", "backward_flow", "forward_pred_backward_flow", "backward_pred_forward_flow"]:
        if k in results:
            extras[k] = results[k]
    results_dict = {"density": density}

    if "static_density" in results and "dynamic_density" in results:  # :125-155
        extras["static_density"] = results["static_density"]
        extras["dynamic_density"] = results["dynamic_density"]
        if return_decomposition:
            static_weights, static_opacities, static_depths = render_weights_opacity_depth_from_density(
                t_starts, t_ends, results["static_density"])
            results_dict["static_opacity"], results_dict["static_depth"] = static_opacities, static_depths
            dynamic_weights, dynamic_opacities, dynamic_depths = render_weights_opacity_depth_from_density(
                t_starts, t_ends, results["dynamic_density"])
            results_dict["dynamic_opacity"], results_dict["dynamic_depth"] = dynamic_opacities, dynamic_depths

    acc_rgb = None
    if "rgb" in results:  # :158-159
        acc_rgb = accumulate_along_rays(weights, values=results["rgb"])
    elif "static_rgb" in results and "dynamic_rgb" in results:  # :160-214
        # ratios, shadowed blend and both accumulations in one kernel each way (:131-136,165-175)
        shadow_ratio = results.get("shadow_ratio", 0.0)
        acc_rgb, acc_shadow_sq = ops.blend_accumulate(weights, results["density"].reshape(weights.shape),
                                                      results["static_density"].reshape(weights.shape),
                                                      results["dynamic_density"].reshape(weights.shape), results["static_rgb"],
                                                      results["dynamic_rgb"], results.get("shadow_ratio"))
        if acc_shadow_sq is not None:
            results_dict["shadow_ratio"] = acc_shadow_sq
        if return_decomposition:
            results_dict["static_rgb"] = accumulate_along_rays(static_weights, values=results["static_rgb"])
            if "shadow_ratio" in results:
                results_dict["shadow_reduced_static_rgb"] = accumulate_along_rays(
                    static_weights, values=results["static_rgb"] * (1 - shadow_ratio))
                shadow_only_static_rgb = accumulate_along_rays(static_weights, values=results["static_rgb"] * shadow_ratio)
                acc_shadow = accumulate_along_rays(weights, values=shadow_ratio)
                results_dict["shadow_only_static_rgb"] = shadow_only_static_rgb + (1 - acc_shadow)
                results_dict["shadow"] = accumulate_along_rays(weights, values=shadow_ratio)
            results_dict["dynamic_rgb"] = accumulate_along_rays(dynamic_weights, values=results["dynamic_rgb"])
            if "forward_flow" in results:
                results_dict["forward_flow"] = accumulate_along_rays(dynamic_weights, values=results["forward_flow"])
                results_dict["backward_flow"] = accumulate_along_rays(dynamic_weights, values=results["backward_flow"])

    # geometry (:102-122) and the sky composite (:217-220) in one per-ray kernel:
    # opacity = clamp(sum w, 1e-6, 1), depth = sum(w t) / opacity, rgb = acc_rgb + rgb_sky * (1 - opacity)
    opacities, depths, median_depth, rgb_out = ops.ray_epilogue(stats, acc_rgb, results.get("rgb_sky") if acc_rgb is not None else None)
    results_dict.update({"depth": depths, "opacity": opacities, "median_depth": median_depth})
    if rgb_out is not None:
        results_dict["rgb"] = rgb_out
    if "rgb_sky" in results and "static_rgb" in results_dict:  # :221-226
        results_dict["static_rgb"] = results_dict["static_rgb"] + results["rgb_sky"] * (1.0 - results_dict["static_opacity"])

    def _finish_dino():
        if "dino_sky_feat" in results:
            results_dict["dino_feat"] = results_dict["dino_feat"] + results["dino_sky_feat"] * (1.0 - results_dict["opacity"])
        if "dino_pe" in results:
            results_dict["dino_pe_free"] = results_dict["dino_feat"].clone()
            results_dict["dino_pe"] = results["dino_pe"]
            results_dict["dino_feat"] = results_dict["dino_feat"] + results["dino_pe"]

    if "dino_feat" in results:  # :229-246
        results_dict["dino_feat"] = accumulate_along_rays(weights, values=results["dino_feat"])
        _finish_dino()
    elif "static_dino_feat" in results and "dynamic_dino_feat" in results:  # :247-282
        if weights.is_cuda and results["static_dino_feat"].dim() == 3:
            # [r4] ratios, blend and accumulation of the C-channel features in one launch each way (:131-136,247-252)
            results_dict["dino_feat"] = ops.blend_accumulate_wide(weights, results["density"], results["static_density"],
                                                                  results["dynamic_density"], results["static_dino_feat"],
                                                                  results["dynamic_dino_feat"])
        else:
            static_ratio = results["static_density"] / (results["density"] + 1e-6)
            dynamic_ratio = results["dynamic_density"] / (results["density"] + 1e-6)
            dino_feat = static_ratio[..., None] * results["static_dino_feat"] + dynamic_ratio[..., None] * results["dynamic_dino_feat"]
            results_dict["dino_feat"] = accumulate_along_rays(weights, values=dino_feat)
        _finish_dino()
        if return_decomposition:
            results_dict["static_dino"] = accumulate_along_rays(static_weights, values=results["static_dino_feat"])
            results_dict["dynamic_dino"] = accumulate_along_rays(dynamic_weights, values=results["dynamic_dino_feat"])
            if "dino_sky_feat" in results:
                results_dict["static_dino"] = results_dict["static_dino"] + results["dino_sky_feat"] * (1.0 - results_dict["opacity"])

    results_dict["extras"] = extras
    return results_dict


def render_rays(radiance_field: RadianceField = None, proposal_estimator: PropNetEstimator = None,
                proposal_networks: Optional[List[DensityField]] = None, data_dict: Dict[str, Tensor] = None,
                cfg=None, proposal_requires_grad: bool = False, return_decomposition: bool = False,
                prefix="") -> Dict[str, Tensor]:
    """render_utils.py:290-389.  ``cfg`` is read by attribute only (cfg.nerf.sampling.num_samples,
    cfg.nerf.propnet.{num_samples_per_prop,near_plane,far_plane,sampling_type}, cfg.render.render_chunk_size)."""
    rays_shape = data_dict[prefix + "origins"].shape
    if len(rays_shape) == 3:
        height, width, _ = rays_shape
        num_rays = height * width
        reshaped = {k: v.reshape(num_rays, -1).squeeze() for k, v in data_dict.items()}
    else:
        num_rays, _ = rays_shape
        reshaped = data_dict.copy()

    def _per_sample(chunk: Dict[str, Tensor], n_samples: int, keys) -> Dict[str, Tensor]:
        # stride-0 views instead of repeat_interleave copies (:319-323,332-336)
        return {k: chunk[k][..., None].expand(*chunk[k].shape, n_samples) for k in keys if k in chunk}

    def prop_sigma_fn(t_starts, t_ends, proposal_network: DensityField):
        pre = getattr(t_starts, "_emer_points", None)   # [r5] computed by the sampler's launch for exactly this network's box
        if pre is not None and pre[2] is proposal_network.aabb and pre[3] == bool(proposal_network.unbounded):
            normed = pre[0]
        else:
            normed, _ = ops.ray_points(chunk[prefix + "origins"], chunk[prefix + "viewdirs"], t_starts, t_ends,
                                       proposal_network.aabb, proposal_network.unbounded)
        return {"density": proposal_network.density_from_normed(normed)}

    def query_fn(t_starts, t_ends):
        S = t_starts.shape[-1]
        t_dirs = chunk[prefix + "viewdirs"][..., None, :].expand(-1, S, -1)
        sub_dict = _per_sample(chunk, S, [k for k in chunk if k not in (prefix + "viewdirs", prefix + "origins", "pixel_coords")
                                          and chunk[k].dim() == 1])
        sub_dict["t_starts"], sub_dict["t_ends"] = t_starts, t_ends
        if "pixel_coords" in chunk:
            sub_dict["pixel_coords"] = chunk["pixel_coords"]
        # o + d (t0 + t1) / 2 and the contraction in one kernel (sample positions never carry a gradient,
        # nerfacc_prop_net.py:89); the un-contracted positions are only needed by the flow warp (:553-620)
        want_pos = radiance_field.flow_xyz_encoder is not None
        pre = getattr(t_starts, "_emer_points", None)   # [r5] computed by the sampler's launch (same expressions, bitwise)
        if pre is not None and pre[2] is radiance_field.aabb and pre[3] == bool(radiance_field.unbounded) and (pre[1] is not None or not want_pos):
            normed, positions = pre[0], pre[1]
        else:
            normed, positions = ops.ray_points(chunk[prefix + "origins"], chunk[prefix + "viewdirs"], t_starts, t_ends,
                                               radiance_field.aabb, radiance_field.unbounded, want_positions=want_pos)
        results_dict = radiance_field(positions, t_dirs, sub_dict, return_density_only=(prefix == "lidar_"),
                                      normed_positions=normed, hash_encodings=False)  # rendering() reads none of them
        results_dict["density"] = results_dict["density"].squeeze(-1)
        return results_dict

    results = []
    chunk_size = 2 ** 24 if radiance_field.training else cfg.render.render_chunk_size
    for i in range(0, num_rays, chunk_size):
        chunk = {k: v[i:i + chunk_size] for k, v in reshaped.items()}
        assert proposal_networks is not None, "proposal_networks is required."
        t_starts, t_ends = proposal_estimator.sampling(
            # The reference builds these closures with a late-binding lambda (render_utils.py:356-358:
            # `lambda *args: prop_sigma_fn(*args, p) for p in proposal_networks`), so EVERY level queries
            # the LAST proposal network and the earlier ones are never evaluated or trained.  Reproduced
            # on purpose: results and checkpoints must match the reference.
            prop_sigma_fns=[lambda *args: prop_sigma_fn(*args, proposal_networks[-1]) for _ in proposal_networks],
            num_samples=cfg.nerf.sampling.num_samples,
            prop_samples=cfg.nerf.propnet.num_samples_per_prop,
            n_rays=chunk[prefix + "origins"].shape[0],
            near_plane=cfg.nerf.propnet.near_plane,
            far_plane=cfg.nerf.propnet.far_plane,
            sampling_type=cfg.nerf.propnet.sampling_type,
            stratified=radiance_field.training,
            requires_grad=proposal_requires_grad,
            # [r5] the sampler's launch also computes the sample points of its intervals: every proposal level queries the LAST proposal
            # network (the closures above), the final samples go to the radiance field
            ray_geometry=(chunk[prefix + "origins"], chunk[prefix + "viewdirs"],
                          [(proposal_networks[-1].aabb, proposal_networks[-1].unbounded, False)] * len(proposal_networks)
                          + [(radiance_field.aabb, radiance_field.unbounded, radiance_field.flow_xyz_encoder is not None)]),
        )
        chunk_results = rendering(t_starts, t_ends, query_fn=query_fn, return_decomposition=return_decomposition)
        extras = chunk_results.pop("extras")
        results.append(chunk_results)
    # (a training batch is one chunk: no concatenation copies of [R,S] tensors)
    render_results = dict(results[0]) if len(results) == 1 else {k: torch.cat([r[k] for r in results], 0) for k in results[0]}
    extras["density"] = render_results.pop("density")
    for k, v in render_results.items():
        render_results[k] = v.reshape(list(rays_shape[:-1]) + list(v.shape[1:]))
    render_results["extras"] = extras
    return render_results
