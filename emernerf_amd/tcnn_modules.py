"""Drop-in for the reference's ``third_party/tcnn_modules.py`` (vendored tiny-cuda-nn torch bindings).

Only what EmerNeRF instantiates is provided: ``Encoding`` with ``otype: HashGrid``
(radiance_fields/encodings.py:133-146).  ``Network`` / ``NetworkWithInputEncoding`` are vendored in the
reference (tcnn_modules.py:281-372) but never instantiated; SphericalHarmonics is reachable only through
``SHEncoder``, which no shipped config selects -- both raise NotImplementedError here.

The module keeps the reference's contract: ``.params`` is ONE flat fp32 ``nn.Parameter`` laid out
(level, entry, feature) so reference checkpoints load (state_dict key ``...tcnn_encoding.params``);
``.n_output_dims``; ``forward(x[N,D]) -> [N, L*F]`` with inputs forced to fp32 contiguous
(tcnn_modules.py:235-263; the batch-granularity padding there is an upstream implementation detail and
is not needed by the HIP kernels).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib, ops


def free_temporary_memory():
    """tcnn_modules.py:98-102 -- the HIP library owns no device memory, so there is nothing to free."""
    return None


class Encoding(torch.nn.Module):
    """tcnn.Encoding(n_input_dims, encoding_config, seed=1337, dtype=None) for HashGrid encodings."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        otype = encoding_config.get("otype")
        if otype != "HashGrid":
            raise NotImplementedError(f"emernerf_amd.tcnn_modules.Encoding: otype {otype!r} is not on the EmerNeRF hot path")
        if encoding_config.get("interpolation", "linear").lower() != "linear":
            raise NotImplementedError("only linear interpolation is implemented")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.seed = seed
        self.desc = _lib.make_grid_desc(
            n_input_dims, int(encoding_config["n_levels"]), int(encoding_config["n_features_per_level"]),
            int(encoding_config["log2_hashmap_size"]), int(encoding_config["base_resolution"]),
            float(encoding_config["per_level_scale"]))
        self.n_output_dims = self.desc.n_levels * self.desc.n_features
        # table precision: fp32 is what the reference runs (encodings.py:118,142-146 never override it)
        self.dtype = dtype or torch.float32
        if self.dtype not in (torch.float32, torch.float16):
            raise ValueError(f"unsupported encoding dtype {self.dtype}")
        g = torch.Generator().manual_seed(seed)
        n_params = self.desc.n_entries * self.desc.n_features
        # tcnn initial_params: U(-1e-4, 1e-4) (pcg32 stream upstream; torch generator here)
        init = (torch.rand(n_params, generator=g, dtype=torch.float32) * 2.0 - 1.0) * 1e-4
        self.params = torch.nn.Parameter(init)  # fp32 master copy, as tcnn_modules.py:219-221
        self.loss_scale = 128.0 if self.dtype == torch.float16 else 1.0

    def __getstate__(self):  # ctypes structs do not pickle; rebuild from the config (tcnn_modules.py:265-275)
        state = self.__dict__.copy()
        state["desc"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        c = self.encoding_config
        self.desc = _lib.make_grid_desc(self.n_input_dims, int(c["n_levels"]), int(c["n_features_per_level"]),
                                        int(c["log2_hashmap_size"]), int(c["base_resolution"]), float(c["per_level_scale"]))

    def forward(self, x: Tensor) -> Tensor:
        if not x.is_cuda:
            raise _lib.EmerError("Encoding.forward needs a GPU tensor (no CPU fallback)")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.n_input_dims)
        params = self.params if self.dtype == torch.float32 else self.params.to(torch.float16)
        out = ops.hashgrid_encode(x2, params, self.desc,
                                  grad_dtype=None if self.dtype == torch.float32 else torch.float16)
        return out.view(*lead, self.n_output_dims)

    def forward_level_major(self, x: Tensor, skip_dx_rows: int = 0) -> Tensor:
        """[N, D] -> [L, N, F]: the grid kernels' native layout, consumed directly by the fused MLP chains.
        dtype float16: the fp32 master is cast to fp16 for the call, the kernels gather from the fp16 copy, gradients are
        accumulated in fp32 into the master (ops._HashGridLMFn; BASELINE.md 2.2 "fp16 tables, fp32 grad accumulation")
        when the owner-computes backward covers the grid; otherwise tcnn's all-fp16 path (fp16 atomics)."""
        if not x.is_cuda:
            raise _lib.EmerError("Encoding.forward needs a GPU tensor (no CPU fallback)")
        x2 = x.reshape(-1, self.n_input_dims)
        if self.dtype == torch.float32:
            return ops.hashgrid_encode_lm(x2, self.params, self.desc, skip_dx_rows=skip_dx_rows)
        if ops.sliced_supported(self.desc):
            return ops.hashgrid_encode_lm(x2, self.params, self.desc, table_dtype=torch.float16, skip_dx_rows=skip_dx_rows)
        return ops.hashgrid_encode_lm(x2, self.params.to(torch.float16), self.desc, grad_dtype=torch.float16, skip_dx_rows=skip_dx_rows)

    def extra_repr(self):
        return f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, dtype={self.dtype}, {self.encoding_config}"
