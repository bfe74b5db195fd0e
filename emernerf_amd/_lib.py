"""ctypes binding of libemernerf_hip.so -- the C-ABI boundary declared in include/emernerf_hip.h.

There is NO fallback: if the shared library is missing or a symbol does not resolve, importing the
product path raises.  A CPU/PyTorch fallback would void every parity claim.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libemernerf_hip.so")
EMER_MAX_LEVELS = 32

F32, F16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TRUNC_EXP = 0, 1, 2, 3
STOT_TYPES = {"uniform": 0, "uniform_lindisp": 1, "lindisp": 2, "sqrt": 3, "log": 4, "uniform_lindisp_0": 5}  # TRANSFROM_DICT, nerfacc_prop_net.py:298-314


class GridDesc(ctypes.Structure):
    """struct emer_grid_desc (include/emernerf_hip.h)."""
    _fields_ = [
        ("n_dims", c_uint32), ("n_levels", c_uint32), ("n_features", c_uint32),
        ("log2_hashmap_size", c_uint32), ("base_resolution", c_uint32), ("per_level_scale", c_float),
        ("scale", c_float * EMER_MAX_LEVELS), ("res", c_uint32 * EMER_MAX_LEVELS),
        ("size", c_uint32 * EMER_MAX_LEVELS), ("offset", c_uint32 * EMER_MAX_LEVELS),
        ("hashed", c_uint32 * EMER_MAX_LEVELS), ("n_entries", c_uint32),
    ]


class EmerError(RuntimeError):
    pass


_P = c_void_p
_GP = ctypes.POINTER(GridDesc)

# name -> argtypes; every entry must be declared in include/emernerf_hip.h (tests check both ways)
SIGNATURES = {
    "emer_profile_next": [_P, _P],
    "emer_grid_desc_init": [_GP, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_float],
    "emer_hashgrid_fwd": [_GP, _P, _P, c_int, _P, c_int64, c_int64, _P, c_int64, _P],
    "emer_hashgrid_bwd_params": [_GP, _P, _P, c_int64, c_int64, _P, c_int, c_int64, _P],
    "emer_hashgrid_bwd_params_sliced": [_GP, _P, _P, c_int64, c_int64, _P, _P, c_int64, _P],
    "emer_hashgrid_bwd_params_sliced_add": [_GP, _P, _P, c_int64, c_int64, _P, _P, c_int64, _P],
    "emer_hashgrid_bwd_params_sliced_levels": [_GP, _P, _P, c_int64, c_int64, _P, _P, c_int64, c_int32, c_int32, _P],
    "emer_hashgrid_sliced_split_level": [_GP],
    "emer_hashgrid_sliced_plan": [_GP, _P, _P],
    "emer_hashgrid_slice_masks": [_GP, _P, _P, c_int64, _P],
    "emer_hashgrid_sliced_supported": [_GP],
    "emer_hashgrid_mask_rows": [_P],
    "emer_hashgrid_bwd_input": [_GP, _P, _P, c_int, _P, c_int64, c_int64, _P, c_int64, _P],
    "emer_hashgrid_fwd_jac": [_GP, _P, _P, _P, c_int64, c_int64, _P, _P, c_int64, c_int64, _P],
    "emer_hashgrid_bwd_input_jac": [_GP, _P, _P, c_int64, c_int64, _P, c_int64, _P],
    "emer_layout_transpose": [_P, _P, c_int32, c_int64, c_int32, c_int, _P],
    "emer_contract_fwd": [_P, _P, c_int, _P, c_int64, _P],
    "emer_contract_bwd": [_P, _P, c_int, _P, _P, c_int64, _P],
    "emer_flow_warp_fwd": [_P, _P, _P, _P, _P, c_float, _P, c_int, _P, _P, c_int64, _P],
    "emer_flow_warp_bwd": [_P, _P, _P, _P, c_int, _P, _P, _P, c_int64, _P],
    "emer_ray_points": [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int64, c_int32, _P],
    "emer_importance_sample": [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_float, c_float, c_int, _P],
    "emer_importance_sample_points": [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P, c_int, _P, _P, _P],
    "emer_stot": [_P, c_int64, c_float, c_float, c_int, _P, _P],
    "emer_prop_loss": [_P, _P, c_int32, _P, _P, c_int32, c_float, c_int, c_int64, c_float, _P, _P, c_int, _P, _P],
    "emer_reduce_sum": [_P, c_int64, c_int, _P, _P],
    "emer_scale": [_P, _P, c_float, _P, c_int64, _P],
    "emer_cast_f32_f16": [_P, _P, c_int64, _P],
    "emer_render_weights_fwd": [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_render_weights_bwd": [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P],
    "emer_composite_rgb_fwd": [_P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_composite_rgb_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P],
    "emer_blend_accumulate_fwd": [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P],
    "emer_blend_accumulate_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_blend_accumulate_wide_fwd": [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P],
    "emer_blend_accumulate_wide_bwd": [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P],
    "emer_ray_epilogue_fwd": [_P, _P, _P, c_int64, _P, _P, _P, _P, _P],
    "emer_ray_epilogue_bwd": [_P, _P, _P, _P, _P, c_int64, _P, _P, _P],
    "emer_pixel_loss_fwd": [_P, _P, _P, _P, c_int64, c_float, c_float, _P, _P, _P],
    "emer_pixel_loss_bwd": [_P, _P, _P, _P, c_int64, c_float, c_float, _P, _P, _P, _P],
    "emer_reg_losses_fwd": [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, _P, _P, _P, c_int64, c_float, _P, _P, _P, _P],
    "emer_reg_losses_bwd": [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, _P, _P, _P, c_int64, c_float, _P, c_float,
                            _P, _P, _P, _P, _P, _P],
    "emer_reg_losses_fwd6": [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, _P, _P, _P],
    "emer_reg_losses_bwd6": [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, _P, c_int64, c_float, _P, c_float,
                             _P, _P, _P, _P, _P],
    "emer_lidar_loss": [_P, _P, _P, _P, c_int64, c_int32, c_float, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P],
    "emer_accumulate_fwd": [_P, _P, c_int64, c_int32, c_int32, _P, _P],
    "emer_accumulate_bwd": [_P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P],
    "emer_linear_fwd": [_P, c_int64, _P, _P, _P, c_int64, c_int64, c_int32, c_int32, c_int, _P, _P],
    "emer_linear_bwd": [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P, _P, c_int64, c_int32,
                        c_int32, c_int, _P, _P, _P],
    "emer_mlp_chain": [_P, c_int64, _P],
    "emer_wgrad_segmented": [_P, c_int64, _P, _P, c_int32, _P, _P, c_int64, _P, c_int64, c_int32, c_int32, _P],
    "emer_neck_supported": [c_int32, c_int32, c_int32, c_int32],
    "emer_neck_fwd": [_P, c_int32, c_int32, c_int64, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P],
    "emer_neck_bwd": [_P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, c_int32, _P, _P, _P, _P, _P],
    "emer_neck_bwd_fused_supported": [c_int32, c_int32, c_int32, c_int32],
    "emer_neck_bwd_fused": [_P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P, c_int32, _P, _P, _P, c_int64, _P, _P, c_int64, _P, _P],
    "emer_rmlp_supported": [c_int32, c_int32, c_int32, c_int32, c_int32],
    "emer_rmlp_fwd": [_P, c_int64, c_int32, c_int32, c_int32, c_int64, c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, c_int64, _P],
    "emer_rmlp_bwd": [_P, c_int64, _P, _P, c_int32, c_int32, c_int32, c_int64, c_int32, _P, _P, _P, c_int32, _P, _P, _P, c_int64, _P],
    "emer_rmlp_bwd_fused_supported": [c_int32, c_int32, c_int32, c_int32, c_int32],
    "emer_rmlp_bwd_fused": [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, c_int32, c_int64, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32,
                            _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, _P],
    "emer_rgb_head_fwd": [_P, c_int64, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_field_fwd_supported": [c_int32, c_int32],
    "emer_density_bwd_fused": [_P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P],
    "emer_field_fwd": [_P, c_int32, c_int32, c_int64, c_int32, _P, _P, _P, _P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_rgb_head_bwd": [_P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P],
    "emer_rgb_head_bwd_fused_supported": [c_int32],
    "emer_rgb_head_bwd_recompute": [_P, _P, _P, _P, c_int64, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64,
                                    _P, c_int64, _P, _P],
    "emer_rgb_head_bwd_fused": [_P, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64,
                                _P, _P],
    "emer_trunc_exp_fwd": [_P, c_int64, _P, c_int64, _P],
    "emer_trunc_exp_bwd": [_P, _P, _P, c_int64, c_int64, _P],
    "emer_dir_encode": [_P, _P, c_int64, c_int32, c_int, _P],
    "emer_aggregate3_fwd": [_P, c_int64, _P, _P],
    "emer_aggregate3_bwd": [_P, c_int64, _P, _P],
    "emer_aggregate3_density_fwd": [_P, c_int64, c_int32, _P, _P, _P],
    "emer_aggregate3_density_bwd": [_P, _P, _P, c_int64, c_int32, _P, _P],
    "emer_ray_inputs_fwd": [_P, c_int64, _P, c_int64, _P, c_int32, c_int32, c_int32, c_int64, _P, c_int64, _P, c_int64, _P],
    "emer_embed_grad": [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, _P],
    "emer_ray_pre_fwd": [_P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P],
    "emer_ray_pre_bwd": [_P, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P, c_int64, _P],
    "emer_ray_head_fwd": [_P, c_int64, c_int64, _P, c_int64, _P, _P, c_int32, c_int, _P, _P, _P, _P],
    "emer_ray_head_bwd": [_P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int32, c_int, _P, _P, _P, _P],
    "emer_ray_wgrad": [_P, c_int32, c_int64, _P],
    "emer_sample_uniform": [_P, c_uint64, c_int64, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P],
    "emer_lidar_sample_rays": [_P, c_uint64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_sample_importance": [_P, c_int64, _P, c_uint64, c_int64, _P, _P, _P],
    "emer_buffer_to_pixels": [_P, c_int64, c_int32, c_int32, c_int32, _P, c_int32, c_int32, _P, c_uint64, _P, _P, _P, _P],
    "emer_gen_rays": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "emer_adam_step": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_int32, _P],
}

# workspace-size queries (return int64 floats, not an error code)
INT64_FUNCTIONS = {
    "emer_linear_bwd_workspace": [c_int64, c_int32, c_int32],
    "emer_neck_bwd_fused_workspace": [c_int32, c_int32, c_int64, c_int32],
    "emer_rmlp_bwd_fused_workspace": [c_int32, c_int32, c_int32, c_int64, c_int32],
    "emer_rgb_head_bwd_workspace": [c_int64],
    "emer_rgb_head_bwd_fused_workspace": [c_int64, c_int32],
    "emer_density_bwd_fused_workspace": [c_int32, c_int32, c_int64],
    "emer_importance_sample_points_capacity": [],
}

ALLOW_MISSING_SYMBOLS = False  # never set by the product path

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime (libamdhip64) and the process must end up with exactly one -- if this
    # library were loaded before torch, it would bind the system copy and see no device once torch initialises its own
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EmerError(
            f"{LIB_PATH} not found. Build it with `python -m emernerf_amd._build` (hipcc, gfx950). "
            "emernerf_amd has no CPU/PyTorch fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    for table, restype in ((SIGNATURES, c_int), (INT64_FUNCTIONS, c_int64)):
        for name, argtypes in table.items():
            try:
                fn = getattr(lib, name)  # AttributeError if the symbol is missing
            except AttributeError:
                if ALLOW_MISSING_SYMBOLS:  # tools/_libsel.py only: an older build of the library in a same-session A/B
                    continue
                raise
            fn.argtypes = argtypes
            fn.restype = restype
    lib.emer_last_error.restype = ctypes.c_char_p
    lib.emer_version.restype = c_int
    _lib = lib
    return lib


class KernelTimer:
    """Opt-in per-entry-point timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py to measure kernel durations live inside the timed region."""

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in self.names}
        self.tags = {n: [] for n in self.names}
        self.ns = {n: [] for n in self.names}   # grid entry points: samples of the launch (the roofline's algorithmic bytes are per sample)
        self.tag = None

    def elapsed_us(self):
        import torch
        torch.cuda.synchronize()
        return {n: [a.elapsed_time(b) * 1e3 for a, b in ev] for n, ev in self.events.items()}


TIMER = None  # set to a KernelTimer to enable


# position of the sample count among the arguments of the grid entry points (include/emernerf_hip.h)
_N_ARG = {"emer_hashgrid_fwd": 8, "emer_hashgrid_fwd_jac": 9, "emer_hashgrid_bwd_params_sliced": 7, "emer_hashgrid_bwd_params_sliced_add": 7,
          "emer_hashgrid_bwd_params_sliced_levels": 7, "emer_hashgrid_bwd_params": 7, "emer_hashgrid_bwd_input": 8, "emer_hashgrid_bwd_input_jac": 6}
_TIGHT = ("emer_hashgrid_fwd", "emer_hashgrid_fwd_jac", "emer_hashgrid_bwd_params_sliced", "emer_hashgrid_bwd_params_sliced_add", "emer_hashgrid_bwd_params_sliced_levels")  # entries that record events around their kernel themselves


def call(name: str, *args) -> None:
    lib = load()
    t = TIMER
    if t is not None and name in t.names:
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if name in _TIGHT:
            # the library re-records both events immediately around the kernel (hipExtLaunchKernelGGL): the measured
            # interval excludes the host's enqueue latency between two launches
            b.record()
            lib.emer_profile_next(c_void_p(a.cuda_event), c_void_p(b.cuda_event))
            rc = getattr(lib, name)(*args)
        else:
            rc = getattr(lib, name)(*args)
            b.record()
        t.events[name].append((a, b))
        tag = None
        if name.startswith("emer_hashgrid"):  # distinguish the main grid from the proposal grids
            d = args[0]._obj
            tag = (d.n_dims, d.n_levels, d.n_features)
        t.tags[name].append(tag)
        t.ns[name].append(int(args[_N_ARG[name]]) if name in _N_ARG else None)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EmerError(f"{name} failed (rc={rc}): {lib.emer_last_error().decode()}")


def make_grid_desc(n_dims, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale) -> GridDesc:
    d = GridDesc()
    call("emer_grid_desc_init", ctypes.byref(d), n_dims, n_levels, n_features, log2_hashmap_size,
         base_resolution, per_level_scale)
    return d
