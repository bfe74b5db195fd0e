"""Encoders with the reference's interface (radiance_fields/encodings.py) on the HIP kernels."""
from __future__ import annotations

import json
import logging
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from . import tcnn_modules as tcnn

logger = logging.getLogger()


class XYZ_Encoder(nn.Module):
    encoder_type = "XYZ_Encoder"

    def __init__(self, n_input_dims):
        super().__init__()
        self.n_input_dims = n_input_dims

    @property
    def n_output_dims(self) -> int:
        raise NotImplementedError


class SinusoidalEncoder(XYZ_Encoder):
    """radiance_fields/encodings.py:60-104.  The reference only ever builds it as the direction
    encoder (3 inputs, degrees 0..4, identity on; radiance_field.py:126-128); that case runs on
    ``emer_dir_encode``.  Other shapes are not on the hot path."""
    encoder_type = "SinusoidalEncoder"

    def __init__(self, n_input_dims: int = 3, min_deg: int = 0, max_deg: int = 10, enable_identity: bool = True):
        super().__init__(n_input_dims)
        self.min_deg, self.max_deg, self.enable_identity = min_deg, max_deg, enable_identity
        self.register_buffer("scales", Tensor([2 ** i for i in range(min_deg, max_deg + 1)]))

    @property
    def n_output_dims(self) -> int:
        return (int(self.enable_identity) + (self.max_deg - self.min_deg + 1) * 2) * self.n_input_dims

    @torch.no_grad()
    def forward(self, x: Tensor, remap: bool = False) -> Tensor:
        if self.n_input_dims != 3 or self.min_deg != 0 or not self.enable_identity:
            raise NotImplementedError("SinusoidalEncoder: only the direction-encoder shape (3, 0, max_deg, identity) is implemented")
        return ops.dir_encode(x, self.max_deg, remap=remap)


class HashEncoder(XYZ_Encoder):
    """radiance_fields/encodings.py:107-160 with identical constructor, attributes and state_dict."""
    encoder_type = "HashEncoder"

    def __init__(self, n_input_dims: int = 3, n_levels: int = 16, base_resolution: int = 16, max_resolution: int = 2048,
                 log2_hashmap_size: int = 19, n_features_per_level: int = 2, dtype=torch.float32, verbose: bool = True) -> None:
        super().__init__(n_input_dims)
        self.num_levels = n_levels
        self.base_resolution = base_resolution
        self.max_resolution = max_resolution
        self.log2_hashmap_size = log2_hashmap_size
        self.n_features_per_level = n_features_per_level
        self.growth_factor = np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1))
        self.encoding_config = {
            "otype": "HashGrid",
            "n_levels": n_levels,
            "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size,
            "base_resolution": base_resolution,
            "per_level_scale": self.growth_factor,
            "interpolation": "linear",
        }
        self.tcnn_encoding = tcnn.Encoding(n_input_dims=n_input_dims, encoding_config=self.encoding_config, dtype=dtype)
        self.num_parameters = self.tcnn_encoding.params.shape
        if verbose:
            logger.info(f"HashGrid encoding config: \n {json.dumps(self.encoding_config)}")
            logger.info(f"HashGrid params: {self.tcnn_encoding.params.numel() / 1e6}M, dtype {self.tcnn_encoding.dtype}")

    @property
    def n_output_dims(self) -> int:
        return self.tcnn_encoding.n_output_dims

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.tcnn_encoding(in_tensor)


def build_xyz_encoder_from_cfg(xyz_encoder_cfg, verbose=True) -> XYZ_Encoder:
    """radiance_fields/encodings.py:163-187."""
    if xyz_encoder_cfg.type == "HashEncoder":
        return HashEncoder(
            n_input_dims=xyz_encoder_cfg.n_input_dims,
            n_levels=xyz_encoder_cfg.n_levels,
            n_features_per_level=xyz_encoder_cfg.n_features_per_level,
            base_resolution=xyz_encoder_cfg.base_resolution,
            max_resolution=xyz_encoder_cfg.max_resolution,
            log2_hashmap_size=xyz_encoder_cfg.log2_hashmap_size,
            verbose=verbose,
        )
    if xyz_encoder_cfg.type == "SinusoidalEncoder":
        return SinusoidalEncoder(n_input_dims=xyz_encoder_cfg.n_input_dims, min_deg=xyz_encoder_cfg.min_deg,
                                 max_deg=xyz_encoder_cfg.max_deg, enable_identity=xyz_encoder_cfg.enable_identity)
    raise NotImplementedError(f"Unknown / off-path encoder type: {xyz_encoder_cfg.type}")
