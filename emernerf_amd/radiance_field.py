"""RadianceField / DensityField with the reference's interface on the HIP kernels.

Mirrors radiance_fields/radiance_field.py (and radiance_fields/mlp.py) of the reference: same
constructor arguments, same sub-module / parameter names (reference checkpoints load with
``load_state_dict``), same ``forward`` contract, same default initialisation order
(so ``torch.manual_seed`` reproduces the reference's nn.Linear weights).  The arithmetic runs in
``emernerf_amd.ops`` (hand-written HIP): hash-grid encode/backward, contraction, fp32-MFMA linear layers
with fused ReLU / sigmoid / density epilogues.  torch itself only supplies parameter containers,
``torch.cat`` / indexing glue, the appearance-embedding gather and ``grid_sample`` for the (optional)
learnable PE map.
"""
from __future__ import annotations

import os

import logging
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib, fused, ops
from .encodings import HashEncoder, SinusoidalEncoder, build_xyz_encoder_from_cfg

logger = logging.getLogger()


def _run_sequential(seq: nn.Sequential, x: Tensor, density_from_col0: bool = False):
    """Evaluate nn.Sequential(Linear, ReLU, ..., Linear[, Sigmoid]) with fused-activation HIP linears."""
    mods = list(seq)
    lins = [m for m in mods if isinstance(m, nn.Linear)]
    # plain Linear-ReLU-...-Linear[-Sigmoid] stacks on row-major input: one fused chain launch each way
    body = mods[:-1] if isinstance(mods[-1], nn.Sigmoid) else mods
    pattern_ok = len(body) % 2 == 1 and all(isinstance(m, nn.Linear if j % 2 == 0 else nn.ReLU) for j, m in enumerate(body))
    if (pattern_ok and not density_from_col0 and x.is_cuda and all(l.bias is not None for l in lins)
            and fused.seq_mlp_supported([l.weight for l in lins])):
        lead = x.shape[:-1]
        y = fused.seq_mlp(x.reshape(-1, x.shape[-1]), [l.weight for l in lins], [l.bias for l in lins],
                          _lib.ACT_SIGMOID if isinstance(mods[-1], nn.Sigmoid) else _lib.ACT_NONE)
        return y.view(*lead, -1)
    i, density = 0, None
    while i < len(mods):
        lin = mods[i]
        assert isinstance(lin, nn.Linear), f"unexpected module {type(lin)} in head"
        act, step = None, 1
        if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
            act, step = "relu", 2
        elif i + 1 < len(mods) and isinstance(mods[i + 1], nn.Sigmoid):
            act, step = "sigmoid", 2
        last = i + step >= len(mods)
        if last and density_from_col0:
            assert act is None
            x, density = ops.linear_with_density(x, lin.weight, lin.bias)
        else:
            x = ops.linear(x, lin.weight, lin.bias, act)
        i += step
    return (x, density) if density_from_col0 else x


class _EmbedFn(torch.autograd.Function):
    """nn.Embedding lookup whose backward is ONE index_add (atomic scatter) instead of torch's sort + segmented-reduce
    pipeline (~20 launches for 8192 rays); same values up to fp32 summation order."""

    @staticmethod
    def forward(ctx, weight: Tensor, idx: Tensor):
        ctx.save_for_backward(idx)
        ctx.shape = weight.shape
        return weight.index_select(0, idx.reshape(-1)).view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g: Tensor):
        (idx,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, device=g.device, dtype=g.dtype)
        dw.index_add_(0, idx.reshape(-1), g.reshape(-1, ctx.shape[1]))
        return dw, None


_EMBED_CACHE: Dict[tuple, Tensor] = {}


def _embed_per_ray(weight: Tensor, idx: Tensor) -> Tensor:
    """The rgb head and the sky head of one forward pass look up the SAME per-ray indices (radiance_field.py:637-643,668-674):
    the second lookup reuses the first (one gather, one scatter in the backward; autograd sums the two consumers' gradients).
    The cache holds one entry and is keyed by the index storage, its version and the weight's, so it can never serve a stale
    lookup; RadianceField.forward clears it on entry."""
    key = (weight.data_ptr(), weight._version, idx.data_ptr(), idx._version, tuple(idx.shape), idx.stride(0), torch.is_grad_enabled())
    hit = _EMBED_CACHE.get(key)
    if hit is None:
        _EMBED_CACHE.clear()
        hit = _EMBED_CACHE[key] = _EmbedFn.apply(weight, idx)
    return hit


class MLP(nn.Module):
    """radiance_fields/mlp.py:7-46 (skip-connection MLP of the rgb / sky heads)."""

    def __init__(self, in_dims: int, out_dims: int, num_layers: int = 3, hidden_dims: Optional[int] = 256,
                 skip_connections: Optional[Tuple[int]] = [0]) -> None:
        super().__init__()
        self.in_dims, self.hidden_dims, self.n_output_dims = in_dims, hidden_dims, out_dims
        self.num_layers, self.skip_connections = num_layers, skip_connections
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dims, out_dims))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    layers.append(nn.Linear(in_dims, hidden_dims))
                elif i in skip_connections:
                    layers.append(nn.Linear(in_dims + hidden_dims, hidden_dims))
                else:
                    layers.append(nn.Linear(hidden_dims, hidden_dims))
            layers.append(nn.Linear(hidden_dims, out_dims))
        self.layers = nn.ModuleList(layers)

    def forward(self, x: Tensor, final_act: Optional[str] = None) -> Tensor:
        inp = x
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            if i in self.skip_connections:
                x = torch.cat([x, inp], -1)
            x = ops.linear(x, layer.weight, layer.bias, "relu" if i < n - 1 else final_act)
        return x


FUSE_FLOW_WARP = True  # ... and build its warped xyzt query points (and their gradient) in one launch each way (ops.flow_warp) [r4]
BATCH_XYZT = True  # flow configs: evaluate each xyzt table once per dependency level (RadianceField._flow_branch_batched)
FUSE_FIELD = os.environ.get("EMER_FUSE_FIELD", "1") != "0"   # neck + rgb head of the static model as one forward launch (fused.RgbRider); 0: two launches


class RadianceField(nn.Module):
    """radiance_fields/radiance_field.py:20-785."""

    def __init__(
        self,
        xyz_encoder: HashEncoder,
        dynamic_xyz_encoder: Optional[HashEncoder] = None,
        flow_xyz_encoder: Optional[HashEncoder] = None,
        aabb: Union[Tensor, List[float]] = [-1, -1, -1, 1, 1, 1],
        num_dims: int = 3,
        density_activation: Optional[Callable] = None,
        unbounded: bool = True,
        geometry_feature_dim: int = 15,
        base_mlp_layer_width: int = 64,
        head_mlp_layer_width: int = 64,
        enable_cam_embedding: bool = False,
        enable_img_embedding: bool = False,
        num_cams: int = 3,
        appearance_embedding_dim: int = 16,
        semantic_feature_dim: int = 64,
        feature_mlp_layer_width: int = 256,
        feature_embedding_dim: int = 768,
        enable_sky_head: bool = False,
        enable_shadow_head: bool = False,
        enable_feature_head: bool = False,
        num_train_timesteps: int = 0,
        interpolate_xyz_encoding: bool = False,
        enable_learnable_pe: bool = True,
        enable_temporal_interpolation: bool = False,
    ) -> None:
        super().__init__()
        if density_activation is not None:
            raise NotImplementedError("only the reference's default density activation trunc_exp(x - 1) is fused")
        if not isinstance(aabb, Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb)
        self.unbounded, self.num_cams, self.num_dims = unbounded, num_cams, num_dims
        self.enable_cam_embedding, self.enable_img_embedding = enable_cam_embedding, enable_img_embedding
        self.appearance_embedding_dim = appearance_embedding_dim
        self.geometry_feature_dim = geometry_feature_dim
        if not enable_feature_head:
            semantic_feature_dim = 0
        self.semantic_feature_dim = semantic_feature_dim

        # ---- static field (radiance_field.py:72-80)
        self.xyz_encoder = xyz_encoder
        self.base_mlp = nn.Sequential(
            nn.Linear(self.xyz_encoder.n_output_dims, base_mlp_layer_width), nn.ReLU(),
            nn.Linear(base_mlp_layer_width, geometry_feature_dim + semantic_feature_dim))
        # ---- dynamic field (:82-96)
        self.interpolate_xyz_encoding = interpolate_xyz_encoding
        self.dynamic_xyz_encoder = dynamic_xyz_encoder
        self.enable_temporal_interpolation = enable_temporal_interpolation
        if self.dynamic_xyz_encoder is not None:
            self.register_buffer("training_timesteps", torch.zeros(num_train_timesteps))
            self.dynamic_base_mlp = nn.Sequential(
                nn.Linear(self.dynamic_xyz_encoder.n_output_dims, base_mlp_layer_width), nn.ReLU(),
                nn.Linear(base_mlp_layer_width, geometry_feature_dim + semantic_feature_dim))
        # ---- flow field (:98-111)
        self.flow_xyz_encoder = flow_xyz_encoder
        if self.flow_xyz_encoder is not None:
            self.flow_mlp = nn.Sequential(
                nn.Linear(self.flow_xyz_encoder.n_output_dims, base_mlp_layer_width), nn.ReLU(),
                nn.Linear(base_mlp_layer_width, base_mlp_layer_width), nn.ReLU(),
                nn.Linear(base_mlp_layer_width, 6))
        # ---- appearance embedding (:113-123)
        if self.enable_cam_embedding:
            self.appearance_embedding = nn.Embedding(num_cams, appearance_embedding_dim)
        elif self.enable_img_embedding:
            self.appearance_embedding = nn.Embedding(num_train_timesteps * num_cams, appearance_embedding_dim)
        else:
            self.appearance_embedding = None
        self.direction_encoding = SinusoidalEncoder(n_input_dims=3, min_deg=0, max_deg=4)
        emb = appearance_embedding_dim if (enable_cam_embedding or enable_img_embedding) else 0
        # ---- colour head (:130-143)
        self.rgb_head = MLP(in_dims=geometry_feature_dim + self.direction_encoding.n_output_dims + emb, out_dims=3,
                            num_layers=3, hidden_dims=head_mlp_layer_width, skip_connections=[1])
        # ---- shadow head (:145-153)
        self.enable_shadow_head = enable_shadow_head
        if enable_shadow_head:
            self.shadow_head = nn.Sequential(nn.Linear(geometry_feature_dim, base_mlp_layer_width), nn.ReLU(),
                                             nn.Linear(base_mlp_layer_width, 1), nn.Sigmoid())
        # ---- sky heads (:155-187)
        self.enable_sky_head = enable_sky_head
        if enable_sky_head:
            self.sky_head = MLP(in_dims=self.direction_encoding.n_output_dims + emb, out_dims=3, num_layers=3,
                                hidden_dims=head_mlp_layer_width, skip_connections=[1])
            if enable_feature_head:
                self.dino_sky_head = nn.Sequential(
                    nn.Linear(self.direction_encoding.n_output_dims + emb, feature_mlp_layer_width), nn.ReLU(),
                    nn.Linear(feature_mlp_layer_width, feature_mlp_layer_width), nn.ReLU(),
                    nn.Linear(feature_mlp_layer_width, feature_embedding_dim))
        # ---- feature head (:189-217)
        self.enable_feature_head = enable_feature_head
        if enable_feature_head:
            self.dino_head = nn.Sequential(
                nn.Linear(semantic_feature_dim, feature_mlp_layer_width), nn.ReLU(),
                nn.Linear(feature_mlp_layer_width, feature_mlp_layer_width), nn.ReLU(),
                nn.Linear(feature_mlp_layer_width, feature_embedding_dim))
            self.register_buffer("feats_reduction_mat", torch.zeros(feature_embedding_dim, 3))
            self.register_buffer("feat_color_min", torch.zeros(3, dtype=torch.float32))
            self.register_buffer("feat_color_max", torch.ones(3, dtype=torch.float32))
            self.enable_learnable_pe = enable_learnable_pe
            if enable_learnable_pe:
                self.learnable_pe_map = nn.Parameter(0.05 * torch.randn(1, feature_embedding_dim // 2, 80, 120), requires_grad=True)
                self.pe_head = nn.Sequential(nn.Linear(feature_embedding_dim // 2, feature_embedding_dim))
        self.time_diff = 0
        self._time_diff_src, self._time_diff_f = None, 0.0

    # ------------------------------------------------------------------ bookkeeping (:219-276)
    def register_normalized_training_timesteps(self, normalized_timesteps: Tensor, time_diff: float = None) -> None:
        if self.dynamic_xyz_encoder is not None:
            self.training_timesteps.copy_(normalized_timesteps)
            self.training_timesteps = self.training_timesteps.to(self.device)
            if time_diff is not None:
                self.time_diff = time_diff
            elif len(self.training_timesteps) > 1:
                self.time_diff = self.training_timesteps[1] - self.training_timesteps[0]
            else:
                self.time_diff = 0
            # ``time_diff`` stays what the reference makes it (a 0-dim tensor on the default route); the one-launch flow warp takes the
            # same fp32 value as a kernel argument, read back ONCE here -- outside any graph capture -- so that the default registration
            # takes the fused path too
            self._time_diff_src, self._time_diff_f = self.time_diff, float(self.time_diff)

    def set_aabb(self, aabb: Union[Tensor, List[float]]) -> None:
        if not isinstance(aabb, Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        logger.info(f"Set aabb from {self.aabb} to {aabb}")
        self.aabb.copy_(aabb)
        self.aabb = self.aabb.to(self.device)

    def register_feats_reduction_mat(self, feats_reduction_mat: Tensor, feat_color_min: Tensor, feat_color_max: Tensor) -> None:
        self.feats_reduction_mat.copy_(feats_reduction_mat)
        self.feat_color_min.copy_(feat_color_min)
        self.feat_color_max.copy_(feat_color_max)

    @property
    def device(self) -> torch.device:
        return self.aabb.device

    # ------------------------------------------------------------------ field pieces
    def contract_points(self, positions: Tensor) -> Tensor:
        """:278-300 (contraction + selector zeroing), one fused kernel, differentiable."""
        return ops.contract_points(positions, self.aabb, self.unbounded)

    def _base(self, encoder: HashEncoder, mlp: nn.Sequential, x: Tensor):
        """grid -> Linear-ReLU-Linear (+ density of feature 0): one level-major grid kernel + one fused chain."""
        enc_lm = encoder.tcnn_encoding.forward_level_major(x)
        return fused.base_mlp(enc_lm, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)

    def _base_split(self, encoder: HashEncoder, mlp: nn.Sequential, x: Tensor, rider=None):
        """(geo feats, semantic feats or None, density) of the neck.  With the shipped widths (hidden 64, geometry and
        semantic features 64 each) this is the register-resident kernel, which writes the two halves as separate
        [N, 64] tensors -- the split of :400 costs nothing and an unused semantic half gets no backward work."""
        enc = encoder.tcnn_encoding
        n_out = mlp[2].out_features
        if (self.geometry_feature_dim == 64 and n_out in (64, 128) and n_out == 64 + self.semantic_feature_dim
                and fused.neck_supported(enc.desc.n_levels, enc.desc.n_features, mlp[0].out_features, n_out)):
            enc_lm = enc.forward_level_major(x)
            # the rider is built AFTER the encode: its per-ray input node must be younger than the grid node, so that autograd
            # runs the embedding gradient BEFORE the table backward (the data-parallel early bucket relies on the static
            # table's backward being the last writer of the step, trainer.py)
            rider = rider() if callable(rider) else rider
            return fused.neck(enc_lm, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, rider=rider)
        feats, density = self._base(encoder, mlp, x)
        geo, sem = torch.split(feats, [self.geometry_feature_dim, self.semantic_feature_dim], dim=-1)
        return geo, sem, density

    def _static_from_normed(self, normed_positions: Tensor):
        feats, density = self._base(self.xyz_encoder, self.base_mlp, normed_positions.reshape(-1, self.num_dims))
        lead = normed_positions.shape[:-1]
        return feats.view(*lead, -1), density.view(*lead)

    def _rgb_rider(self, directions: Optional[Tensor], data_dict):
        """The colour query of this forward pass as a rider of the static neck's launch (fused.RgbRider), or None when the
        query is not the per-ray, 64-wide, skip-at-1 head the fused kernel covers."""
        head = self.rgb_head
        if not (FUSE_FIELD and directions is not None and directions.dim() == 3 and self._per_ray(directions)
                and len(head.layers) == 3 and list(head.skip_connections) == [1] and head.hidden_dims == 64):
            return None
        both = self._ray_inputs(directions, data_dict)
        lw = [p for l in head.layers for p in (l.weight, l.bias)]
        if both is None or any(p is None for p in lw):
            return None
        return fused.RgbRider(both[0], directions.shape[1], lw, torch.is_grad_enabled())

    def _static_split(self, normed_positions: Tensor, rider=None):
        geo, sem, density = self._base_split(self.xyz_encoder, self.base_mlp, normed_positions.reshape(-1, self.num_dims), rider)
        lead = normed_positions.shape[:-1]
        return geo.view(*lead, -1), (None if sem is None else sem.view(*lead, -1)), density.view(*lead)

    def forward_static_hash(self, positions: Tensor) -> Tuple[Tensor, Tensor]:
        """:302-318.  Returns (encoded_features, normed_positions)."""
        normed_positions = self.contract_points(positions)
        feats, _ = self._static_from_normed(normed_positions)
        return feats, normed_positions

    def _dynamic(self, normed_positions: Tensor, normed_timestamps: Tensor, want_density: bool, want_hash: bool = True):
        """want_hash=False: the row-major hash encodings (a "to be studied" output of the reference, :457-459,
        615-617) are not needed, so the fused level-major path is used."""
        if normed_timestamps.shape[-1] != 1:
            normed_timestamps = normed_timestamps.unsqueeze(-1)
        temporal_positions = torch.cat([normed_positions, normed_timestamps.to(normed_positions.dtype)], dim=-1)
        lead = temporal_positions.shape[:-1]
        if not want_hash:
            geo, sem, density = self._base_split(self.dynamic_xyz_encoder, self.dynamic_base_mlp,
                                                 temporal_positions.reshape(-1, self.num_dims + 1))
            feats = (geo.view(*lead, -1), None if sem is None else sem.view(*lead, -1))  # already split (see _base_split)
            return feats, None, (density.view(*lead) if want_density else None)
        enc_t = self.dynamic_xyz_encoder.tcnn_encoding
        mlp = self.dynamic_base_mlp
        n_out = mlp[2].out_features
        if (temporal_positions.is_cuda and self.geometry_feature_dim == 64 and n_out == 64 + self.semantic_feature_dim
                and fused.neck_supported(enc_t.desc.n_levels, enc_t.desc.n_features, mlp[0].out_features, n_out)):
            # level-major encoding -> register-resident neck; the row-major hash encodings the reference hands back
            # (:457-459, 615-617, "to be studied") are one transpose of the same tensor, not a second path through the MLP
            enc_lm = enc_t.forward_level_major(temporal_positions.reshape(-1, self.num_dims + 1))
            geo, sem, density = fused.neck(enc_lm, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)
            feats = geo if sem is None else torch.cat([geo, sem], dim=-1)
            enc = ops.lm_to_rm(enc_lm)
            return feats.view(*lead, -1), enc.view(*lead, -1), (density.view(*lead) if want_density else None)
        enc = self.dynamic_xyz_encoder(temporal_positions.reshape(-1, self.num_dims + 1))
        if want_density:
            feats, density = _run_sequential(self.dynamic_base_mlp, enc, density_from_col0=True)
            return feats.view(*lead, -1), enc.view(*lead, -1), density.view(*lead)
        feats = _run_sequential(self.dynamic_base_mlp, enc)
        return feats.view(*lead, -1), enc.view(*lead, -1), None

    def forward_dynamic_hash(self, normed_positions: Tensor, normed_timestamps: Tensor, return_hash_encodings: bool = False):
        """:320-357 (the ``if True:`` branch: no temporal interpolation)."""
        feats, enc, _ = self._dynamic(normed_positions, normed_timestamps, want_density=False)
        return (feats, enc) if return_hash_encodings else feats

    def forward_flow_hash(self, normed_positions: Tensor, normed_timestamps: Tensor) -> Tensor:
        """:359-389.  Evaluation with ``enable_temporal_interpolation``: the flow field BETWEEN two training timesteps is the flow
        MLP of the linearly interpolated xyzt encodings at the two nearest training timesteps (``temporal_interpolation``, :844-905,
        called with interpolate_xyz_encoding=True).  (The dynamic branch never interpolates: the reference's ``if True:``, :336-337.)"""
        if normed_timestamps.shape[-1] != 1:
            normed_timestamps = normed_timestamps.unsqueeze(-1)
        if self.enable_temporal_interpolation and not self.training:
            return self._flow_hash_interpolated(normed_positions, normed_timestamps)
        return self._flow_from_xyzt(torch.cat([normed_positions, normed_timestamps.to(normed_positions.dtype)], dim=-1))

    def _flow_hash_interpolated(self, normed_positions: Tensor, normed_timestamps: Tensor) -> Tensor:
        """temporal_interpolation (:844-905) for the flow encoder.  One timestamp per ray (the slice [:, 0(, 0)] the reference reads);
        when EVERY ray sits on a training timestep the plain evaluation is used (its ``torch.allclose`` test: one host read, eval only)."""
        slice_t = normed_timestamps[:, 0] if normed_timestamps.dim() == 2 else normed_timestamps[:, 0, 0]
        train_t = self.training_timesteps.to(slice_t)
        # find_topk_nearby_timesteps (nerf_utils.py:31-57): the two training timesteps closest to each query
        idx = torch.topk((train_t[None, :] - slice_t[:, None]).abs(), k=2, dim=1, largest=False).indices
        closest = train_t[idx]
        if torch.allclose(closest[:, 0], slice_t):
            return self._flow_from_xyzt(torch.cat([normed_positions, normed_timestamps.to(normed_positions.dtype)], dim=-1))
        assert normed_positions.dim() == 3, "temporal interpolation expects (rays, samples, 3) positions, as the reference does"
        S = normed_positions.shape[1]
        left, right = closest[:, 0], closest[:, 1]
        offset = ((slice_t - left) / (right - left))[:, None, None]
        enc = self.flow_xyz_encoder
        xl = torch.cat([normed_positions, left[:, None, None].expand(-1, S, 1)], dim=-1)
        xr = torch.cat([normed_positions, right[:, None, None].expand(-1, S, 1)], dim=-1)
        el = enc(xl.reshape(-1, self.num_dims + 1)).view(*xl.shape[:-1], -1)
        er = enc(xr.reshape(-1, self.num_dims + 1)).view(*xr.shape[:-1], -1)
        return _run_sequential(self.flow_mlp, el * (1 - offset) + er * offset)

    def _flow_from_xyzt(self, temporal_positions: Tensor) -> Tensor:
        enc_t = self.flow_xyz_encoder.tcnn_encoding
        lins = [m for m in self.flow_mlp if isinstance(m, nn.Linear)]
        if (temporal_positions.is_cuda and len(lins) == 3 and all(l.bias is not None for l in lins)
                and fused.rmlp_supported([l.weight for l in lins], enc_t.desc.n_levels * enc_t.desc.n_features, enc_t.desc.n_features)):
            # level-major xyzt encoding straight into the register-resident 3-layer MLP (no row-major copy either way)
            enc_lm = enc_t.forward_level_major(temporal_positions.reshape(-1, self.num_dims + 1))
            flow = fused.seq_mlp_lm(enc_lm, [l.weight for l in lins], [l.bias for l in lins])
        else:
            enc = self.flow_xyz_encoder(temporal_positions.reshape(-1, self.num_dims + 1))
            flow = _run_sequential(self.flow_mlp, enc)
        return flow.view(*temporal_positions.shape[:-1], 6)

    def _noise(self, like: Tensor) -> Tensor:
        """Temporal-aggregation noise (:567-570); overridable so tests can replay the reference's draw."""
        if self.training:
            return torch.rand_like(like)[..., 0:1]
        return torch.ones_like(like)[..., 0:1]

    def temporal_aggregation(self, positions: Tensor, normed_timestamps: Tensor, forward_flow: Tensor,
                             backward_flow: Tensor, dynamic_feats: Tensor) -> Dict[str, Tensor]:
        """:553-620 (Eq. 8 of the paper)."""
        if normed_timestamps.shape[-1] != 1:
            normed_timestamps = normed_timestamps.unsqueeze(-1)
        noise = self._noise(forward_flow)
        fwd_pos = self.contract_points(positions + forward_flow * noise)
        bwd_pos = self.contract_points(positions + backward_flow * noise)
        fwd_t = torch.clamp(normed_timestamps + self.time_diff * noise, 0, 1.0)
        bwd_t = torch.clamp(normed_timestamps - self.time_diff * noise, 0, 1.0)
        fwd_feats, fwd_enc = self.forward_dynamic_hash(fwd_pos, fwd_t, return_hash_encodings=True)
        bwd_feats, bwd_enc = self.forward_dynamic_hash(bwd_pos, bwd_t, return_hash_encodings=True)
        fwd_pred_flow = self.forward_flow_hash(fwd_pos, fwd_t)
        bwd_pred_flow = self.forward_flow_hash(bwd_pos, bwd_t)
        aggregated = (dynamic_feats + 0.5 * fwd_feats + 0.5 * bwd_feats) / 2.0
        return {
            "dynamic_feats": aggregated,
            "forward_pred_backward_flow": fwd_pred_flow[..., 3:],
            "backward_pred_forward_flow": bwd_pred_flow[..., :3],
            "forward_dynamic_hash_encodings": fwd_enc,
            "backward_dynamic_hash_encodings": bwd_enc,
        }

    def _flow_branch_batched(self, positions: Tensor, normed_positions: Tensor, normed_timestamps: Tensor,
                             want_hash: bool) -> Optional[Dict[str, Tensor]]:
        """The flow branch of ``forward`` (:434-459 + temporal_aggregation :553-620) with every xyzt table evaluated in as few
        launches as the data flow allows: the flow grid at the current positions (N samples) -> flow MLP -> warped positions ->
        the dynamic grid ONCE at [current | forward-warped | backward-warped] (3N samples, one encode, one neck, one owner-
        computes backward) and the flow grid ONCE at both warped sets (2N).  Six encodes / six table backwards / four input-
        gradient launches of the call-by-call order become 3 / 3 / 2, and each table's slices are scanned, accumulated and
        flushed once per step instead of three times (no 40 MB temporaries merged with add_).  Same arithmetic per sample.
        Returns None when a stack is not covered by the fused kernels (the caller then takes the call-by-call path)."""
        enc_d, enc_f = self.dynamic_xyz_encoder.tcnn_encoding, self.flow_xyz_encoder.tcnn_encoding
        mlp = self.dynamic_base_mlp
        n_out = mlp[2].out_features
        lins = [m for m in self.flow_mlp if isinstance(m, nn.Linear)]
        if not (normed_positions.is_cuda and positions is not None and self.geometry_feature_dim == 64
                and n_out == 64 + self.semantic_feature_dim
                and fused.neck_supported(enc_d.desc.n_levels, enc_d.desc.n_features, mlp[0].out_features, n_out)
                and len(lins) == 3 and all(l.bias is not None for l in lins)
                and fused.rmlp_supported([l.weight for l in lins], enc_f.desc.n_levels * enc_f.desc.n_features, enc_f.desc.n_features)):
            return None
        if normed_timestamps.shape[-1] != 1:
            normed_timestamps = normed_timestamps.unsqueeze(-1)
        lead = normed_positions.shape[:-1]
        D4 = self.num_dims + 1
        ts = normed_timestamps.to(normed_positions.dtype)
        x_cur = torch.cat([normed_positions, ts], dim=-1).reshape(-1, D4)
        N = x_cur.shape[0]
        fw, fb = [l.weight for l in lins], [l.bias for l in lins]
        # (1) flow at the current positions
        flow = fused.seq_mlp_lm(enc_f.forward_level_major(x_cur), fw, fb).view(*lead, 6)
        forward_flow, backward_flow = flow[..., :3], flow[..., 3:]
        # (2) warped positions and times (:567-580)
        noise = self._noise(forward_flow)
        if isinstance(self.time_diff, Tensor):
            if self.time_diff is not self._time_diff_src:   # assigned directly after the registration: one host read, once
                self._time_diff_src, self._time_diff_f = self.time_diff, float(self.time_diff)
            td = self._time_diff_f
        else:
            td = float(self.time_diff)
        fused_warp = (FUSE_FLOW_WARP and self.num_dims == 3 and not positions.requires_grad
                      and not x_cur.requires_grad and not noise.requires_grad and flow.is_contiguous())
        if fused_warp:
            # [r4] one launch for both warps, both clamps and the batch assembly (and one for their gradient w.r.t. the flow), instead
            # of 14 elementwise / cat launches and ~18 autograd twins; the flow table's 2N-row batch is a second output, so the two
            # tables' input gradients arrive as two tensors (no pad / copy / add)
            x3, x2 = ops.flow_warp(positions, normed_positions, ts, flow, noise, td, self.aabb, self.unbounded)
        else:
            fwd_pos = self.contract_points(positions + forward_flow * noise)
            bwd_pos = self.contract_points(positions + backward_flow * noise)
            fwd_t = torch.clamp(ts + self.time_diff * noise, 0, 1.0)
            bwd_t = torch.clamp(ts - self.time_diff * noise, 0, 1.0)
            x_fwd = torch.cat([fwd_pos, fwd_t.to(fwd_pos.dtype)], dim=-1).reshape(-1, D4)
            x_bwd = torch.cat([bwd_pos, bwd_t.to(bwd_pos.dtype)], dim=-1).reshape(-1, D4)
            x3, x2 = torch.cat([x_cur, x_fwd, x_bwd], dim=0), torch.cat([x_fwd, x_bwd], dim=0)
        # (3) dynamic table: one evaluation of 3N samples, one neck
        # the current positions carry no gradient in a training step (inputs of the model); a caller that asks for one gets it
        enc3 = enc_d.forward_level_major(x3, skip_dx_rows=0 if x_cur.requires_grad else N)
        geo3, sem3, _ = fused.neck(enc3, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)
        # temporal aggregation (:595-613) per feature half, straight from the 3N-row batch: one launch each way, no [., 128]
        # concatenation and no 3N-row cat in the backward
        agg_ok = (N * 64) % 4 == 0
        dyn_density = None
        if agg_ok:
            # [r4] the density (trunc_exp of the aggregated geometry features' column 0, :461) comes out of the same launch
            dyn, dyn_density = ops.aggregate3_density(geo3)
            dyn, dyn_density = dyn.view(*lead, -1), dyn_density.view(*lead)
            if sem3 is not None:
                dyn = (dyn, ops.aggregate3(sem3).view(*lead, -1))  # forward() takes the pair as (geometry, semantic) features
        else:
            feats3 = geo3 if sem3 is None else torch.cat([geo3, sem3], dim=-1)
            cur_f, fwd_f, bwd_f = (t.view(*lead, -1) for t in feats3.split(N, dim=0))
            dyn = (cur_f + 0.5 * fwd_f + 0.5 * bwd_f) / 2.0
        # (4) flow table at both warped sets: one evaluation of 2N samples
        flow2 = fused.seq_mlp_lm(enc_f.forward_level_major(x2), fw, fb)
        fwd_pred, bwd_pred = (t.view(*lead, 6) for t in flow2.split(N, dim=0))
        out = {"forward_flow": forward_flow, "backward_flow": backward_flow,
               "dynamic_feats": dyn, "_dynamic_density": dyn_density,
               "forward_pred_backward_flow": fwd_pred[..., 3:], "backward_pred_forward_flow": bwd_pred[..., :3]}
        # [r5] the cycle loss reads its four operands as column blocks of these two tensors (ops.reg_losses(flow_pair=...)).  The result
        # dictionary keeps the reference's keys exactly, so the pair rides along as an attribute of one of the four views
        out["forward_pred_backward_flow"]._emer_flow_pair = (flow, flow2)
        if want_hash:  # row-major copies of the encodings: part of forward()'s contract (:453-459, 615-617), consumed by nobody
            cur_h, fwd_h, bwd_h = (t.view(*lead, -1) for t in ops.lm_to_rm(enc3).split(N, dim=0))
            out.update({"forward_dynamic_hash_encodings": fwd_h, "backward_dynamic_hash_encodings": bwd_h,
                        "current_dynamic_hash_encodings": cur_h})
        return out

    @staticmethod
    def _per_ray(t: Tensor) -> bool:
        """True for a (R, S[, C]) tensor that is a stride-0 broadcast of per-ray values along S (what
        render_rays passes instead of the reference's repeat_interleave copies)."""
        return t.dim() >= 2 and t.shape[1] > 1 and t.stride(1) == 0

    def _embed(self, idx: Tensor) -> Tensor:
        if self._per_ray(idx):  # look up once per ray, broadcast along the samples (same values, 1/S of the work)
            return _embed_per_ray(self.appearance_embedding.weight, idx[:, 0])[:, None, :].expand(-1, idx.shape[1], -1)
        if idx.dim() == 1:  # the sky head's per-ray indices: the same storage the rgb head just looked up
            return _embed_per_ray(self.appearance_embedding.weight, idx)
        return _EmbedFn.apply(self.appearance_embedding.weight, idx)

    def _encode_dirs(self, directions: Tensor, remap: bool) -> Tensor:
        if self._per_ray(directions):
            enc = self.direction_encoding(directions[:, 0].contiguous(), remap=remap)
            return enc[:, None, :].expand(-1, directions.shape[1], -1)
        return self.direction_encoding(directions, remap=remap)

    def _appearance(self, directions: Tensor, data_dict: Optional[Dict[str, Tensor]]):
        if not (self.enable_cam_embedding or self.enable_img_embedding):
            return None
        data_dict = data_dict or {}
        if "cam_idx" in data_dict and self.enable_cam_embedding:
            return self._embed(data_dict["cam_idx"])
        if "img_idx" in data_dict and self.enable_img_embedding:
            return self._embed(data_dict["img_idx"])
        return torch.ones((*directions.shape[:-1], self.appearance_embedding_dim), device=directions.device) \
            * self.appearance_embedding.weight.mean(dim=0)

    def _ray_inputs(self, directions: Tensor, data_dict: Optional[Dict[str, Tensor]]):
        """(rgb-head rows, sky-head rows) = [direction PE | appearance embedding] per RAY from one launch, shared by the two
        heads of a forward pass (they look up the same indices, :637-643 and :668-674), or None when the inputs are not
        per-ray / there is no embedding (then the general path below runs).  The cache holds one entry and is keyed by the
        storages and versions involved, so it never serves stale rows; forward() clears it on entry."""
        if not (self.enable_cam_embedding or self.enable_img_embedding) or directions is None:
            return None
        data_dict = data_dict or {}
        key = "cam_idx" if ("cam_idx" in data_dict and self.enable_cam_embedding) else \
            ("img_idx" if ("img_idx" in data_dict and self.enable_img_embedding) else None)
        if key is None:
            return None
        idx = data_dict[key]
        idx = idx[:, 0] if self._per_ray(idx) else idx
        d = directions[:, 0] if (directions.dim() == 3 and self._per_ray(directions)) else directions
        enc = self.direction_encoding
        if not (idx.dim() == 1 and d.dim() == 2 and d.shape[0] == idx.shape[0] and idx.dtype == torch.int64 and d.is_cuda
                and d.dtype == torch.float32 and d.stride(1) == 1 and not d.requires_grad
                and enc.n_input_dims == 3 and enc.min_deg == 0 and enc.enable_identity):
            return None
        w = self.appearance_embedding.weight
        ck = (w.data_ptr(), w._version, idx.data_ptr(), idx._version, idx.stride(0), d.data_ptr(), d._version, d.stride(0),
              d.shape[0], torch.is_grad_enabled())
        hit = _EMBED_CACHE.get(ck)
        if hit is None:
            _EMBED_CACHE.clear()
            hit = _EMBED_CACHE[ck] = fused.ray_inputs(w, idx, d, enc.max_deg)
        return hit

    def _query_rgb_fused(self, directions, geo_feats, dynamic_geo_feats, data_dict):
        """Fused rgb head: per-ray [dir-PE | appearance emb] stays per ray, geo stays per sample, the whole
        3-layer skip MLP + sigmoid is one chain.  Applies when render_rays hands in stride-0 per-ray views."""
        head = self.rgb_head
        if not (directions.dim() == 3 and self._per_ray(directions) and len(head.layers) == 3
                and list(head.skip_connections) == [1] and head.hidden_dims % 4 == 0):
            return None
        R, S = directions.shape[:2]
        data_dict = data_dict or {}
        emb = None
        both = self._ray_inputs(directions, data_dict)
        if both is not None:
            hray = both[0]
        elif self.enable_cam_embedding or self.enable_img_embedding:
            key = "cam_idx" if ("cam_idx" in data_dict and self.enable_cam_embedding) else \
                ("img_idx" if ("img_idx" in data_dict and self.enable_img_embedding) else None)
            if key is None:
                emb = self.appearance_embedding.weight.mean(dim=0)[None, :].expand(R, -1)
            elif self._per_ray(data_dict[key]):
                emb = _embed_per_ray(self.appearance_embedding.weight, data_dict[key][:, 0])
            else:
                return None
        if both is None:
            pe = self.direction_encoding(directions[:, 0].contiguous(), remap=True)
            hray = pe if emb is None else torch.cat([pe, emb], dim=-1)
        lw = [p for l in head.layers for p in (l.weight, l.bias)]

        def run(geo):
            g2 = geo.reshape(R * S, geo.shape[-1])
            if g2.stride(-1) != 1:
                g2 = g2.contiguous()
            return fused.rgb_head(hray, g2, S, *lw).view(R, S, -1)

        results = {"rgb": run(geo_feats)}
        if self.dynamic_xyz_encoder is not None:
            assert dynamic_geo_feats is not None, "Dynamic geometry features are not provided."
            results["dynamic_rgb"] = run(dynamic_geo_feats)
        return results

    def query_rgb(self, directions: Tensor, geo_feats: Tensor, dynamic_geo_feats: Tensor = None,
                  data_dict: Dict[str, Tensor] = None) -> Dict[str, Tensor]:
        """:622-658."""
        fused_out = self._query_rgb_fused(directions, geo_feats, dynamic_geo_feats, data_dict)
        if fused_out is not None:
            return fused_out
        h = self._encode_dirs(directions, remap=True)  # (d + 1) / 2 folded into the kernel
        emb = self._appearance(directions, data_dict)
        if emb is not None:
            h = torch.cat([h, emb], dim=-1)
        results = {"rgb": self.rgb_head(torch.cat([h, geo_feats], dim=-1), final_act="sigmoid")}
        if self.dynamic_xyz_encoder is not None:
            assert dynamic_geo_feats is not None, "Dynamic geometry features are not provided."
            results["dynamic_rgb"] = self.rgb_head(torch.cat([h, dynamic_geo_feats], dim=-1), final_act="sigmoid")
        return results

    def query_sky(self, directions: Tensor, data_dict: Dict[str, Tensor] = None) -> Dict[str, Tensor]:
        """:660-686 (per ray).  Note: the reference does NOT remap directions for the sky head."""
        d = directions if directions.dim() == 2 else directions[:, 0]
        both = self._ray_inputs(d, data_dict)
        if both is not None:
            dd = both[1]
        else:
            dd = self.direction_encoding(d, remap=False)
            emb = self._appearance(directions, data_dict)
            if emb is not None:
                dd = torch.cat([dd, emb], dim=-1)
        head = self.sky_head
        if dd.dim() == 2 and len(head.layers) == 3 and list(head.skip_connections) == [1] and head.hidden_dims % 4 == 0:
            lw = [p for l in head.layers for p in (l.weight, l.bias)]
            results = {"rgb_sky": fused.skip_mlp3(dd, *lw)}  # one chain launch each way (per-ray head)
        else:
            results = {"rgb_sky": head(dd, final_act="sigmoid")}
        if self.enable_feature_head:
            results["dino_sky_feat"] = _run_sequential(self.dino_sky_head, dd)
        return results

    # ------------------------------------------------------------------ forward (:391-551)
    def forward(self, positions: Tensor, directions: Tensor = None, data_dict: Dict[str, Tensor] = {},
                return_density_only: bool = False, combine_static_dynamic: bool = False,
                query_feature_head: bool = True, query_pe_head: bool = True,
                normed_positions: Optional[Tensor] = None, hash_encodings: bool = True) -> Dict[str, Tensor]:
        """``normed_positions`` (extension): the contracted positions when the caller already has them (render_rays
        gets them from the ray-point kernel); ``positions`` may then be None unless the flow branch is on.
        ``hash_encodings`` (extension): False drops the three row-major ``*_dynamic_hash_encodings`` outputs of the flow
        branch ("to be studied" in the reference, consumed by nothing): render_rays never reads them."""
        results_dict = {}
        _EMBED_CACHE.clear()
        fused.clear_riders()
        if normed_positions is None:
            normed_positions = self.contract_points(positions)
        # the static colour query rides along in the neck's launch when it is the plain per-ray one (static model)
        rider = (lambda: self._rgb_rider(directions, data_dict)) if (not return_density_only and normed_positions.dim() == 3) else None
        geo_feats, semantic_feats, static_density = self._static_split(normed_positions, rider)

        has_timestamps = "normed_timestamps" in data_dict or "lidar_normed_timestamps" in data_dict
        dynamic = self.dynamic_xyz_encoder is not None and has_timestamps
        if dynamic:
            normed_timestamps = data_dict["normed_timestamps"] if "normed_timestamps" in data_dict \
                else data_dict["lidar_normed_timestamps"]
            use_flow = self.flow_xyz_encoder is not None
            # (eval-mode temporal interpolation of the flow field takes the call-by-call path: forward_flow_hash)
            interp = self.enable_temporal_interpolation and not self.training
            batched = self._flow_branch_batched(positions, normed_positions, normed_timestamps, hash_encodings) \
                if (use_flow and BATCH_XYZT and not interp) else None
            if batched is not None:
                dynamic_feats = batched["dynamic_feats"]
                fused_density = batched.pop("_dynamic_density")
                results_dict.update(batched)
                if isinstance(dynamic_feats, tuple):  # (geometry, semantic) halves kept apart: the reference's tensor on request only
                    if hash_encodings:
                        results_dict["dynamic_feats"] = torch.cat(dynamic_feats, dim=-1)
                    else:
                        del results_dict["dynamic_feats"]
                dynamic_density = fused_density if fused_density is not None else \
                    ops.trunc_exp_column(dynamic_feats[0] if isinstance(dynamic_feats, tuple) else dynamic_feats, 0)
            else:
                dynamic_feats, dynamic_hash_encodings, dynamic_density = self._dynamic(
                    normed_positions, normed_timestamps, want_density=not use_flow, want_hash=use_flow)
            if use_flow and batched is None:
                flow = self.forward_flow_hash(normed_positions, normed_timestamps)
                forward_flow, backward_flow = flow[..., :3], flow[..., 3:]
                results_dict["forward_flow"] = forward_flow
                results_dict["backward_flow"] = backward_flow
                agg = self.temporal_aggregation(positions, normed_timestamps, forward_flow, backward_flow, dynamic_feats)
                dynamic_feats = agg["dynamic_feats"]
                agg["current_dynamic_hash_encodings"] = dynamic_hash_encodings
                results_dict.update(agg)
                dynamic_density = ops.trunc_exp_column(dynamic_feats, 0)
            if isinstance(dynamic_feats, tuple):
                dynamic_geo_feats, dynamic_semantic_feats = dynamic_feats
            elif self.semantic_feature_dim == 0:
                # (torch.split(x, [64, 0]) is a view forward and a full-size cat backward -- 31 us per flow step for an empty half)
                dynamic_geo_feats, dynamic_semantic_feats = dynamic_feats, dynamic_feats[..., self.geometry_feature_dim:]
            else:
                dynamic_geo_feats, dynamic_semantic_feats = torch.split(
                    dynamic_feats, [self.geometry_feature_dim, self.semantic_feature_dim], dim=-1)
            density = static_density + dynamic_density
            results_dict.update({"density": density, "static_density": static_density, "dynamic_density": dynamic_density})
            if return_density_only:
                return results_dict
            if directions is not None:
                rgb_results = self.query_rgb(directions, geo_feats, dynamic_geo_feats, data_dict=data_dict)
                results_dict["dynamic_rgb"] = rgb_results["dynamic_rgb"]
                results_dict["static_rgb"] = rgb_results["rgb"]
                if combine_static_dynamic:
                    static_ratio = static_density / (density + 1e-6)
                    dynamic_ratio = dynamic_density / (density + 1e-6)
                    results_dict["rgb"] = static_ratio[..., None] * results_dict["static_rgb"] \
                        + dynamic_ratio[..., None] * results_dict["dynamic_rgb"]
            if self.enable_shadow_head:
                shadow_ratio = _run_sequential(self.shadow_head, dynamic_geo_feats)
                results_dict["shadow_ratio"] = shadow_ratio
                if combine_static_dynamic and "rgb" in results_dict:
                    results_dict["rgb"] = static_ratio[..., None] * results_dict["rgb"] * (1 - shadow_ratio) \
                        + dynamic_ratio[..., None] * results_dict["dynamic_rgb"]
        else:
            results_dict["density"] = static_density
            if return_density_only:
                return results_dict
            if directions is not None:
                results_dict["rgb"] = self.query_rgb(directions, geo_feats, data_dict=data_dict)["rgb"]

        if self.enable_feature_head and query_feature_head:
            if self.enable_learnable_pe and query_pe_head:
                pe = F.grid_sample(self.learnable_pe_map, data_dict["pixel_coords"].reshape(1, 1, -1, 2) * 2 - 1,
                                   align_corners=False, mode="bilinear").squeeze(2).squeeze(0).permute(1, 0)
                results_dict["dino_pe"] = _run_sequential(self.pe_head, pe.contiguous())
            dino_feats = _run_sequential(self.dino_head, semantic_feats)
            if dynamic:
                dynamic_dino_feats = _run_sequential(self.dino_head, dynamic_semantic_feats)
                results_dict["static_dino_feat"] = dino_feats
                results_dict["dynamic_dino_feat"] = dynamic_dino_feats
                if combine_static_dynamic:
                    static_ratio = static_density / (density + 1e-6)
                    dynamic_ratio = dynamic_density / (density + 1e-6)
                    results_dict["dino_feat"] = static_ratio[..., None] * dino_feats + dynamic_ratio[..., None] * dynamic_dino_feats
            else:
                results_dict["dino_feat"] = dino_feats

        # sky (the reference's gate looks for a key that never exists, so lidar rays skip sky only through
        # return_density_only; reproduced as is, :540-549)
        if self.enable_sky_head and "lidar_origin" not in data_dict and directions is not None:
            sky_dirs = directions[:, 0]
            reduced = {k: v[:, 0] for k, v in data_dict.items()}
            results_dict.update(self.query_sky(sky_dirs, data_dict=reduced))
        return results_dict

    def query_flow(self, positions: Tensor, normed_timestamps: Tensor, query_density: bool = True) -> Dict[str, Tensor]:
        """:688-713."""
        normed_positions = self.contract_points(positions)
        flow = self.forward_flow_hash(normed_positions, normed_timestamps)
        results = {"forward_flow": flow[..., :3], "backward_flow": flow[..., 3:]}
        if query_density:
            _, _, density = self._dynamic(normed_positions, normed_timestamps, want_density=True, want_hash=False)
            results["dynamic_density"] = density
        return results

    def query_attributes(self, positions: Tensor, normed_timestamps: Tensor = None, query_feature_head: bool = True):
        """:715-785."""
        data = {} if normed_timestamps is None else {"normed_timestamps": normed_timestamps}
        out = self.forward(positions, None, data, combine_static_dynamic=False,
                           query_feature_head=query_feature_head, query_pe_head=False)
        if "static_dino_feat" in out:
            d = out["density"].unsqueeze(-1)
            out["dino_feat"] = (out["static_density"].unsqueeze(-1) * out["static_dino_feat"]
                                + out["dynamic_density"].unsqueeze(-1) * out["dynamic_dino_feat"]) / (d + 1e-6)
        return out


class DensityField(nn.Module):
    """Proposal network: radiance_fields/radiance_field.py:788-841."""

    def __init__(self, xyz_encoder: HashEncoder, aabb: Union[Tensor, List[float]] = [[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]],
                 num_dims: int = 3, density_activation: Optional[Callable] = None, unbounded: bool = False,
                 base_mlp_layer_width: int = 64) -> None:
        super().__init__()
        if density_activation is not None:
            raise NotImplementedError("only the reference's default density activation trunc_exp(x - 1) is fused")
        if not isinstance(aabb, Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb)
        self.num_dims, self.unbounded = num_dims, unbounded
        self.xyz_encoder = xyz_encoder
        self.base_mlp = nn.Sequential(nn.Linear(self.xyz_encoder.n_output_dims, base_mlp_layer_width), nn.ReLU(),
                                      nn.Linear(base_mlp_layer_width, 1))

    @property
    def device(self) -> torch.device:
        return self.aabb.device

    def set_aabb(self, aabb: Union[Tensor, List[float]]) -> None:
        if not isinstance(aabb, Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        logger.info(f"Set propnet aabb from {self.aabb} to {aabb}")
        self.aabb.copy_(aabb)
        self.aabb = self.aabb.to(self.device)

    def density_from_normed(self, normed: Tensor) -> Tensor:
        """normed [..., 3] already contracted -> density [..., 1]."""
        enc_lm = self.xyz_encoder.tcnn_encoding.forward_level_major(normed.reshape(-1, self.num_dims))
        lin0, lin1 = self.base_mlp[0], self.base_mlp[2]
        d = fused.density_mlp(enc_lm, lin0.weight, lin0.bias, lin1.weight, lin1.bias)  # grid -> 64 -> 1 -> trunc_exp, one chain
        return d.view(*normed.shape[:-1], 1)

    def forward(self, positions: Tensor, data_dict: Dict[str, Tensor] = None) -> Dict[str, Tensor]:
        normed = ops.contract_points(positions, self.aabb.reshape(-1), self.unbounded)
        return {"density": self.density_from_normed(normed)}


def build_radiance_field_from_cfg(cfg, verbose=True) -> RadianceField:
    """radiance_fields/radiance_field.py:907-946 (including the hard-coded flow grid, :916-923)."""
    xyz_encoder = build_xyz_encoder_from_cfg(cfg.xyz_encoder, verbose=verbose)
    dynamic_xyz_encoder = flow_xyz_encoder = None
    if cfg.head.enable_dynamic_branch:
        dynamic_xyz_encoder = build_xyz_encoder_from_cfg(cfg.dynamic_xyz_encoder, verbose=verbose)
    if cfg.head.enable_flow_branch:
        flow_xyz_encoder = HashEncoder(n_input_dims=4, n_levels=10, base_resolution=16, max_resolution=4096,
                                       log2_hashmap_size=18, n_features_per_level=4, verbose=verbose)
    return RadianceField(
        xyz_encoder=xyz_encoder, dynamic_xyz_encoder=dynamic_xyz_encoder, flow_xyz_encoder=flow_xyz_encoder,
        unbounded=cfg.unbounded, num_cams=cfg.num_cams,
        geometry_feature_dim=cfg.neck.geometry_feature_dim, base_mlp_layer_width=cfg.neck.base_mlp_layer_width,
        head_mlp_layer_width=cfg.head.head_mlp_layer_width, enable_cam_embedding=cfg.head.enable_cam_embedding,
        enable_img_embedding=cfg.head.enable_img_embedding, appearance_embedding_dim=cfg.head.appearance_embedding_dim,
        enable_sky_head=cfg.head.enable_sky_head, enable_feature_head=cfg.head.enable_feature_head,
        semantic_feature_dim=cfg.neck.semantic_feature_dim, feature_mlp_layer_width=cfg.head.feature_mlp_layer_width,
        feature_embedding_dim=cfg.head.feature_embedding_dim, enable_shadow_head=cfg.head.enable_shadow_head,
        num_train_timesteps=cfg.num_train_timesteps, interpolate_xyz_encoding=cfg.head.interpolate_xyz_encoding,
        enable_learnable_pe=cfg.head.enable_learnable_pe,
        enable_temporal_interpolation=cfg.head.enable_temporal_interpolation)


def build_density_field(aabb=[[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]], type: str = "HashEncoder", n_input_dims: int = 3,
                        n_levels: int = 5, base_resolution: int = 16, max_resolution: int = 128,
                        log2_hashmap_size: int = 20, n_features_per_level: int = 2, unbounded: bool = True) -> DensityField:
    """radiance_fields/radiance_field.py:949-975."""
    if type != "HashEncoder":
        raise NotImplementedError(f"Unknown (xyz_encoder) type: {type}")
    enc = HashEncoder(n_input_dims=n_input_dims, n_levels=n_levels, base_resolution=base_resolution,
                      max_resolution=max_resolution, log2_hashmap_size=log2_hashmap_size,
                      n_features_per_level=n_features_per_level, verbose=False)
    return DensityField(xyz_encoder=enc, aabb=aabb, unbounded=unbounded)
