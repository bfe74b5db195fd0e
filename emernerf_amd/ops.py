"""torch.autograd wrappers over the C-ABI kernels (PyTorch here is plumbing: device memory, streams,
autograd bookkeeping).  Every op launches hand-written HIP through ``_lib.call``; nothing in this
module computes on the CPU or through torch math, so a missing extension fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TRUNC_EXP, F16, F32, GridDesc, STOT_TYPES

_ACTS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "trunc_exp": ACT_TRUNC_EXP}


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check_cuda(*ts: Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EmerError("emernerf_amd ops need GPU (HIP) tensors; there is no CPU fallback")


def _ng(*args):
    """Outside autograd recording the inputs of a Function.apply call are detached, so that its forward does not prepare a
    backward nobody will run (see fused._ng)."""
    if torch.is_grad_enabled():
        return args
    return tuple(a.detach() if isinstance(a, Tensor) else a for a in args)


def _f32c(t: Tensor) -> Tensor:
    return t.detach().to(torch.float32).contiguous()


def _dtype_tag(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise _lib.EmerError(f"unsupported table dtype {t.dtype}")


# ------------------------------------------------------------------------------------ hash grid
MASK_SCRATCH = 2064  # EMER_SLICE_MASK_SCRATCH: work cursors (and pacing counters) of the backward behind the bitmaps


def sliced_supported(desc: GridDesc) -> bool:
    """True when the owner-computes LDS backward handles this grid (every level <= 64 LDS slices)."""
    return bool(_lib.load().emer_hashgrid_sliced_supported(ctypes.byref(desc)))


def sliced_split_level(desc) -> int:
    """Level k at which the owner-computes table backward is cut into the launches [k, L) and [0, k) without lengthening it by a
    round of work items (0: do not cut).  See emer_hashgrid_sliced_split_level (csrc/hashgrid.hip)."""
    return int(_lib.load().emer_hashgrid_sliced_split_level(ctypes.byref(desc)))


def mask_words(desc: GridDesc, n: int) -> int:
    """64-bit words of the slice bitmaps for n samples: n_levels x rows x ceil(n / 64) + the backward's work cursors
    (rows = 64, or 256 for tables with more than 64 LDS slices per level: emer_hashgrid_mask_rows)."""
    rows = int(_lib.load().emer_hashgrid_mask_rows(ctypes.byref(desc)))
    if rows <= 0:
        raise _lib.EmerError("slice bitmaps requested for a grid the owner-computes backward does not support")
    return desc.n_levels * rows * ((n + 63) // 64) + MASK_SCRATCH


def hashgrid_fwd_raw(desc: GridDesc, x: Tensor, params: Tensor, level_major: bool = True, want_masks: bool = False, jac_row0=None):
    """Encode; returns [L, N, F] (level_major) or [N, L*F] fp32 (and the int64 slice bitmaps, ``mask_words(desc, N)`` long, if asked).
    ``jac_row0`` (fp32 tables): also returns d out / d x of the rows jac_row0 .. N - 1 as [L, N - jac_row0, F, D] (last element of
    the returned tuple) -- see emer_hashgrid_fwd_jac."""
    _check_cuda(x, params)
    N, L, F = x.shape[0], desc.n_levels, desc.n_features
    assert x.shape[1] == desc.n_dims and params.numel() == desc.n_entries * F
    with torch.cuda.device(x.device):
        if level_major:
            out = torch.empty((L, N, F), device=x.device, dtype=torch.float32)
            sn, sl = F, N * F
        else:
            out = torch.empty((N, L * F), device=x.device, dtype=torch.float32)
            sn, sl = L * F, F
        masks = torch.empty((mask_words(desc, N),), device=x.device, dtype=torch.int64) if want_masks else None
        if jac_row0 is None:
            _lib.call("emer_hashgrid_fwd", ctypes.byref(desc), _ptr(x), _ptr(params), _dtype_tag(params), _ptr(out), sn, sl,
                      _ptr(masks), N, _stream(x))
            return (out, masks) if want_masks else out
        assert params.dtype == torch.float32 and 0 <= jac_row0 <= N
        jac = torch.empty((L, N - jac_row0, F, desc.n_dims), device=x.device, dtype=torch.float32)
        _lib.call("emer_hashgrid_fwd_jac", ctypes.byref(desc), _ptr(x), _ptr(params), _ptr(out), sn, sl, _ptr(masks), _ptr(jac),
                  int(jac_row0), N, _stream(x))
    return (out, masks, jac) if want_masks else (out, jac)


# [r4] input gradient of a grid encoding (the flow configs: warped positions) from the Jacobians the forward stores instead of a second
# gather pass over the table (emer_hashgrid_bwd_input): -0.4 ms per flow step at 2048 rays.  EMER_GRID_JAC=0: the gather pass.
GRID_JAC = os.environ.get("EMER_GRID_JAC", "1") != "0"
# The Jacobian is L * F * D * 4 bytes per row that needs dx (640 B for the D4/L10/F4 xyzt tables), held from the forward to the backward
# (under capture: in the graph's pool).  Evaluations whose Jacobian would exceed this budget take the gather pass instead, so a ray batch
# that fitted before the Jacobian path existed still fits.  EMER_GRID_JAC_MAX_MB=0: no limit.
GRID_JAC_MAX_BYTES = int(float(os.environ.get("EMER_GRID_JAC_MAX_MB", "4096")) * (1 << 20))


def _jac_fits(desc, rows: int) -> bool:
    return GRID_JAC_MAX_BYTES <= 0 or rows * desc.n_levels * desc.n_features * desc.n_dims * 4 <= GRID_JAC_MAX_BYTES


def slice_masks(desc: GridDesc, x: Tensor) -> Tensor:
    _check_cuda(x)
    with torch.cuda.device(x.device):
        masks = torch.empty((mask_words(desc, x.shape[0]),), device=x.device, dtype=torch.int64)
        _lib.call("emer_hashgrid_slice_masks", ctypes.byref(desc), _ptr(x), _ptr(masks), x.shape[0], _stream(x))
    return masks


def layout_transpose(src: Tensor, L: int, N: int, F: int, to_row_major: bool) -> Tensor:
    _check_cuda(src)
    with torch.cuda.device(src.device):
        dst = torch.empty((N, L * F) if to_row_major else (L, N, F), device=src.device, dtype=torch.float32)
        _lib.call("emer_layout_transpose", _ptr(src), _ptr(dst), L, N, F, int(to_row_major), _stream(src))
    return dst


class _LmToRmFn(torch.autograd.Function):
    """Level-major grid encoding [L, N, F] -> the reference's row-major [N, L*F] (differentiable; one transpose kernel)."""

    @staticmethod
    def forward(ctx, enc_lm: Tensor):
        L, N, F = enc_lm.shape
        ctx.shape = (L, N, F)
        return layout_transpose(_f32c(enc_lm), L, N, F, to_row_major=True)

    @staticmethod
    def backward(ctx, dout: Tensor):
        L, N, F = ctx.shape
        return layout_transpose(_f32c(dout), L, N, F, to_row_major=False)


def lm_to_rm(enc_lm: Tensor) -> Tensor:
    return _LmToRmFn.apply(enc_lm)


CHECK_FINITE = os.environ.get("EMER_CHECK_FINITE") == "1"


DEFERRED_FINITE: list = []   # (grid name, N, any-bad flag, flat index of the first bad (level, sample)) recorded during a graph capture


def check_deferred_finite(entries) -> None:
    """Raise FloatingPointError if a replayed graph's recorded checks (see _check_finite_grid_grad) saw a non-finite gradient."""
    for name, n, flag, first in entries:
        if bool(flag):
            i = int(first)
            raise FloatingPointError(f"non-finite gradient entering the {name} hash-grid backward at (level, sample) [{i // n}, {i % n}] "
                                     "(first offender; hipGraph replay)")


def _check_finite_grid_grad(dlm: Tensor, desc: GridDesc) -> None:
    """Debug aid (EMER_CHECK_FINITE=1, the analogue of the reference's optim.check_nan, loss/base.py:77-79).  The owner-computes
    backward reduces runs of equal cells with 0/1-masked multiply-adds, so ONE non-finite entry of the incoming gradient
    contaminates other table entries of its wave (0 x inf = NaN; upstream's atomics keep it in the sample's own cells): looking
    at the table gradient afterwards points at the wrong entries.  This check names the offending (level, sample) pairs BEFORE
    the scatter; it costs one reduction over the gradient and a host read, so it is off by default.  While a hipGraph is being
    captured the verdict is left on the device and read after each replay (``check_deferred_finite``)."""
    if torch.cuda.is_current_stream_capturing():
        # a host read cannot be captured: the verdict stays on the device (tensors of the graph's pool, rewritten by every replay)
        # and the owner of the graph reads it after g.replay() (check_deferred_finite; Trainer does when CHECK_FINITE is on)
        per = (~torch.isfinite(dlm)).any(dim=-1).view(-1)                      # [L * N]
        DEFERRED_FINITE.append((f"D{desc.n_dims}/L{desc.n_levels}/F{desc.n_features}", dlm.shape[1], per.any(), per.to(torch.uint8).argmax()))
        return
    bad = ~torch.isfinite(dlm)
    if bool(bad.any()):
        idx = torch.nonzero(bad.any(dim=-1))[:8].tolist()
        raise FloatingPointError(f"non-finite gradient entering the D{desc.n_dims}/L{desc.n_levels}/F{desc.n_features} hash-grid backward at "
                                 f"(level, sample) {idx}{' ...' if int(bad.any(dim=-1).sum()) > 8 else ''}")


def _table_grad_via_autograd(param, grad):
    """A table gradient that is handed BACK to autograd (AccumulateGrad will add it onto ``param.grad``).  A trainer that
    skips zeroing table gradients (``FlatParams.zero_grad`` marks them ``_emer_grad_fresh`` instead: the owner-computes
    backward overwrites them) leaves last step's values in ``.grad``; every path that does not overwrite must therefore zero
    the stale buffer first and clear the mark -- otherwise AccumulateGrad adds onto stale data and ``finish_grads`` later zeroes
    the sum (the row-major / atomic / fp16 fallbacks trained with a zero gradient)."""
    if grad is not None and param is not None and getattr(param, "_emer_grad_fresh", False):
        if param.grad is not None:
            param.grad.zero_()
        param._emer_grad_fresh = False
    return grad


class _HashGridFn(torch.autograd.Function):
    """tcnn ``_module_function`` (third_party/tcnn_modules.py:115-174) on HIP.

    The grid kernels run level-major (coalesced); the row-major [N, L*F] tensor the reference API
    returns is produced / consumed through the LDS transpose kernel.
    """

    @staticmethod
    def forward(ctx, x: Tensor, params: Tensor, desc: GridDesc, grad_dtype):
        xc, pc = _f32c(x), params.detach().contiguous()
        N, L, F = xc.shape[0], desc.n_levels, desc.n_features
        gdt = grad_dtype or torch.float32
        # the forward emits the slice masks of the owner-computes backward when it will be used
        ctx.sliced = bool(ctx.needs_input_grad[1] and gdt == torch.float32 and sliced_supported(desc))
        if ctx.sliced:
            lm, masks = hashgrid_fwd_raw(desc, xc, pc, level_major=True, want_masks=True)
        else:
            lm, masks = hashgrid_fwd_raw(desc, xc, pc, level_major=True), None
        out = layout_transpose(lm, L, N, F, to_row_major=True)
        ctx.desc, ctx.grad_dtype = desc, grad_dtype
        ctx.param_obj = params
        _count_table_eval(params)
        ctx.save_for_backward(xc, pc, masks)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        xc, pc, masks = ctx.saved_tensors
        desc = ctx.desc
        N, L, F = xc.shape[0], desc.n_levels, desc.n_features
        dx = dp = None
        _before_table_grad(ctx.param_obj)
        with torch.cuda.device(xc.device):
            dlm = layout_transpose(_f32c(dout), L, N, F, to_row_major=False)
            st = _stream(xc)
            if ctx.needs_input_grad[1]:
                gdt = ctx.grad_dtype or torch.float32
                if gdt == torch.float32 and masks is not None:
                    # owner-computes LDS scatter: writes every entry once (no memset, no global atomics)
                    grad = torch.empty(pc.numel(), device=xc.device, dtype=torch.float32)
                    _lib.call("emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), _ptr(xc), _ptr(dlm), F, N * F, _ptr(masks),
                              _ptr(grad), N, st)
                else:  # tcnn-style global atomics: fp16-gradient mode, or tables too large for 32 LDS slices
                    grad = torch.zeros(pc.numel(), device=xc.device, dtype=gdt)
                    _lib.call("emer_hashgrid_bwd_params", ctypes.byref(desc), _ptr(xc), _ptr(dlm), F, N * F, _ptr(grad),
                              _dtype_tag(grad), N, st)
                dp = _table_grad_via_autograd(ctx.param_obj, grad.to(pc.dtype) if grad.dtype != pc.dtype else grad)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(xc)
                _lib.call("emer_hashgrid_bwd_input", ctypes.byref(desc), _ptr(xc), _ptr(pc), _dtype_tag(pc), _ptr(dlm), F,
                          N * F, _ptr(dx), N, st)
        return dx, dp, None, None


def _count_table_eval(param) -> None:
    """Data-parallel trainers hang ``_emer_before_table_grad`` on the table whose backward runs LAST in a step and
    ``_emer_after_table_grad`` on the other tables of the main model.  If an encoder is evaluated more than once per step (warped
    positions, chunked training), only the backward of its FIRST forward evaluation is the last one to run; the evaluations are
    counted here so the callbacks fire exactly then (the trainer resets the count at the start of a step)."""
    if getattr(param, "_emer_before_table_grad", None) is not None or getattr(param, "_emer_after_table_grad", None) is not None:
        param._emer_pending_evals = getattr(param, "_emer_pending_evals", 0) + 1


def _before_table_grad(param) -> bool:
    """-> True when this is the last backward of ``param``'s encoder in the step (and a trainer asked to know)."""
    cb, after = getattr(param, "_emer_before_table_grad", None), getattr(param, "_emer_after_table_grad", None)
    if cb is None and after is None:
        return False
    left = getattr(param, "_emer_pending_evals", 1) - 1
    param._emer_pending_evals = max(left, 0)
    if left <= 0 and cb is not None:  # the last backward of this table in the step: everything upstream has its gradient enqueued by now
        cb()
    return left <= 0


def _after_table_grad(param, last: bool, in_place: bool) -> None:
    """The table's gradient of this step is complete IN the trainer's buffer (``in_place``: written or added there by this
    backward, not handed to autograd's AccumulateGrad): a data-parallel trainer starts the table's collective now, behind the
    backward kernels that follow."""
    cb = getattr(param, "_emer_after_table_grad", None)
    if cb is not None and last and in_place:
        cb(param)


class _HashGridLMFn(torch.autograd.Function):
    """Same as _HashGridFn but exchanges the LEVEL-MAJOR [L, N, F] tensors the grid kernels use natively
    (no transpose kernels): what the fused MLP chains read and write.

    Gradient sink (``fused.grad_sinks()``, a trainer that owns its gradient buffers): the owner-computes backward writes
    every table entry exactly once, so it can write STRAIGHT into the parameter's ``.grad`` -- no 49 MB temporary, no
    AccumulateGrad add, and the trainer does not zero the table's gradient beforehand (``params._emer_grad_fresh`` is set
    by the trainer's zero_grad: the first backward of a step overwrites, later ones of the same step -- the flow
    configs evaluate an encoder three times -- add).

    ``table_dtype=torch.float16`` with an fp32 ``params``: half-precision tables the way BASELINE.md 2.2 / tcnn run them --
    the fp32 MASTER is cast to fp16 for this call (emer_cast_f32_f16), the encode and the input gradient read the fp16 copy
    (half the gather bytes), and the gradient is accumulated in fp32 by the owner-computes backward (which never reads the
    table) straight into the master's ``.grad``.  ``skip_dx_rows``: leading rows of ``x`` that carry no gradient (the current
    positions in a batched [current | warped | warped] evaluation): the input-gradient kernel skips them."""

    @staticmethod
    def forward(ctx, x: Tensor, params: Tensor, desc: GridDesc, grad_dtype, table_dtype=None, skip_dx_rows: int = 0):
        xc, pc = _f32c(x), params.detach().contiguous()
        if table_dtype == torch.float16 and pc.dtype == torch.float32:
            ph = torch.empty(pc.shape, device=pc.device, dtype=torch.float16)
            with torch.cuda.device(pc.device):
                _lib.call("emer_cast_f32_f16", _ptr(pc), _ptr(ph), pc.numel(), _stream(pc))
            pc = ph  # what the kernels read; `params` stays the fp32 master (gradient sink)
        gdt = grad_dtype or torch.float32
        ctx.sliced = bool(ctx.needs_input_grad[1] and gdt == torch.float32 and sliced_supported(desc))
        k = min(int(skip_dx_rows), xc.shape[0])
        jac = None
        if GRID_JAC and ctx.needs_input_grad[0] and pc.dtype == torch.float32 and k < xc.shape[0] and _jac_fits(desc, xc.shape[0] - k):
            res = hashgrid_fwd_raw(desc, xc, pc, level_major=True, want_masks=ctx.sliced, jac_row0=k)
            lm, masks, jac = res if ctx.sliced else (res[0], None, res[1])
        elif ctx.sliced:
            lm, masks = hashgrid_fwd_raw(desc, xc, pc, level_major=True, want_masks=True)
        else:
            lm, masks = hashgrid_fwd_raw(desc, xc, pc, level_major=True), None
        ctx.jac = jac   # (not through save_for_backward: never an input or output of the function, freed with the node)
        ctx.desc, ctx.grad_dtype = desc, grad_dtype
        ctx.skip_dx_rows = int(skip_dx_rows)
        ctx.master_dtype = params.dtype
        from . import fused
        sink = fused._sink(params) if ctx.sliced else None
        ctx.param = params if (sink is not None and sink.is_contiguous() and sink.numel() == pc.numel()) else None
        ctx.param_obj = params
        _count_table_eval(params)
        ctx.save_for_backward(xc, pc, masks)
        return lm

    @staticmethod
    def backward(ctx, dlm: Tensor):
        xc, pc, masks = ctx.saved_tensors
        desc = ctx.desc
        N, L, F = xc.shape[0], desc.n_levels, desc.n_features
        dx = dp = None
        last = _before_table_grad(ctx.param_obj)  # data-parallel trainer: early gradient bucket (fires on the table's last backward)
        with torch.cuda.device(xc.device):
            dlm = _f32c(dlm)
            if CHECK_FINITE:
                _check_finite_grid_grad(dlm, desc)
            st = _stream(xc)
            if ctx.needs_input_grad[1]:
                gdt = ctx.grad_dtype or torch.float32
                if gdt == torch.float32 and masks is not None:
                    param = ctx.param
                    direct = param is not None and getattr(param, "_emer_grad_fresh", False) and param.grad is not None
                    # [r5] a further evaluation of the same encoder in this step adds to the table's buffer in the kernel's write-out
                    add = (not direct) and param is not None and param.grad is not None and param.grad.is_contiguous()
                    grad = param.grad.view(-1) if (direct or add) else torch.empty(pc.numel(), device=xc.device, dtype=torch.float32)
                    # (the cut is taken only by the table's LAST backward of the step: an earlier one would start the collective of a range
                    # that a later backward of the same table still adds to)
                    split = getattr(ctx.param_obj, "_emer_table_split", None) if (direct and last) else None
                    if split is not None and 0 < split[0] < L:
                        # data-parallel trainer: the levels [k, L) first, then ITS hook (the collective of that contiguous range of the
                        # table starts on the communication stream), then the levels [0, k) while it runs
                        k, hook = split
                        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), _ptr(xc), _ptr(dlm), F, N * F, _ptr(masks),
                                  _ptr(grad), N, k, L, st)
                        hook(param, int(desc.offset[k]) * F, pc.numel())
                        _lib.call("emer_hashgrid_bwd_params_sliced_levels", ctypes.byref(desc), _ptr(xc), _ptr(dlm), F, N * F, _ptr(masks),
                                  _ptr(grad), N, 0, k, st)
                    else:
                        _lib.call("emer_hashgrid_bwd_params_sliced_add" if add else "emer_hashgrid_bwd_params_sliced", ctypes.byref(desc), _ptr(xc),
                                  _ptr(dlm), F, N * F, _ptr(masks), _ptr(grad), N, st)
                    if direct:
                        param._emer_grad_fresh = False
                        grad = None
                    elif add:
                        grad = None
                    elif param is not None and param.grad is not None:
                        param.grad.view(-1).add_(grad)  # a further evaluation of the same encoder in this step
                        grad = None
                else:
                    grad = torch.zeros(pc.numel(), device=xc.device, dtype=gdt)
                    _lib.call("emer_hashgrid_bwd_params", ctypes.byref(desc), _ptr(xc), _ptr(dlm), F, N * F, _ptr(grad),
                              _dtype_tag(grad), N, st)
                dp = None if grad is None else _table_grad_via_autograd(ctx.param_obj, grad.to(ctx.master_dtype) if grad.dtype != ctx.master_dtype else grad)
                _after_table_grad(ctx.param_obj, last, dp is None)
            if ctx.needs_input_grad[0]:
                k, D = min(ctx.skip_dx_rows, N), desc.n_dims
                dx = torch.empty_like(xc)
                if k > 0:
                    dx[:k].zero_()  # rows without a consumer (torch.cat's backward slices them away)
                if N > k and ctx.jac is not None:
                    _lib.call("emer_hashgrid_bwd_input_jac", ctypes.byref(desc), _ptr(ctx.jac), dlm.data_ptr() + 4 * k * F, F, N * F,
                              dx.data_ptr() + 4 * k * D, N - k, st)
                    ctx.jac = None
                elif N > k:  # level-major dlm [L][N][F]: row k of every level is k*F floats in, the level stride stays N*F
                    _lib.call("emer_hashgrid_bwd_input", ctypes.byref(desc), xc.data_ptr() + 4 * k * D, _ptr(pc), _dtype_tag(pc),
                              dlm.data_ptr() + 4 * k * F, F, N * F, dx.data_ptr() + 4 * k * D, N - k, st)
        return dx, dp, None, None, None, None


def hashgrid_encode_lm(x: Tensor, params: Tensor, desc: GridDesc, grad_dtype=None, table_dtype=None, skip_dx_rows: int = 0) -> Tensor:
    """x [N,D] in [0,1] -> level-major [L, N, F] fp32, differentiable w.r.t. params and x (see _HashGridLMFn for
    ``table_dtype`` / ``skip_dx_rows``)."""
    return _HashGridLMFn.apply(*_ng(x, params, desc, grad_dtype, table_dtype, skip_dx_rows))


def hashgrid_encode(x: Tensor, params: Tensor, desc: GridDesc, grad_dtype=None) -> Tensor:
    """x [N,D] in [0,1] -> [N, L*F] fp32, differentiable w.r.t. params and x."""
    return _HashGridFn.apply(*_ng(x, params, desc, grad_dtype))


# ------------------------------------------------------------------------------ contraction
class _ContractFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos: Tensor, aabb: Tensor, unbounded: bool):
        pc = _f32c(pos).view(-1, 3)
        ab = _f32c(aabb).view(-1)
        with torch.cuda.device(pc.device):
            out = torch.empty_like(pc)
            _lib.call("emer_contract_fwd", _ptr(pc), _ptr(ab), int(unbounded), _ptr(out), pc.shape[0], _stream(pc))
        ctx.save_for_backward(pc, ab)
        ctx.unbounded, ctx.shape = unbounded, pos.shape
        return out.view(pos.shape)

    @staticmethod
    def backward(ctx, dout: Tensor):
        pc, ab = ctx.saved_tensors
        dc = _f32c(dout).view(-1, 3)
        with torch.cuda.device(pc.device):
            dpos = torch.empty_like(pc)
            _lib.call("emer_contract_bwd", _ptr(pc), _ptr(ab), int(ctx.unbounded), _ptr(dc), _ptr(dpos), pc.shape[0],
                      _stream(pc))
        return dpos.view(ctx.shape), None, None


def contract_points(pos: Tensor, aabb: Tensor, unbounded: bool) -> Tensor:
    """contract() + selector zeroing (nerf_utils.py:13-28, radiance_field.py:278-300)."""
    _check_cuda(pos, aabb)
    return _ContractFn.apply(pos, aabb, unbounded)


class _FlowWarpFn(torch.autograd.Function):
    """(x3 [3N, 4], x2 [2N, 4]) = xyzt query points of the batched flow branch (radiance_field.py:567-580), see emer_flow_warp_fwd;
    gradient only w.r.t. ``flow`` [N, 6] (positions, timestamps and noise are inputs of the step)."""

    @staticmethod
    def forward(ctx, positions: Tensor, normed: Tensor, ts: Tensor, flow: Tensor, noise: Tensor, time_diff: float, aabb: Tensor, unbounded: bool):
        pos, nrm, t, fl, nz, ab = _f32c(positions).view(-1, 3), _f32c(normed).view(-1, 3), _f32c(ts).view(-1), _f32c(flow).view(-1, 6), \
            _f32c(noise).view(-1), _f32c(aabb).view(-1)
        N = pos.shape[0]
        assert nrm.shape[0] == N and t.numel() == N and fl.shape[0] == N and nz.numel() == N
        with torch.cuda.device(pos.device):
            x3 = torch.empty((3 * N, 4), device=pos.device, dtype=torch.float32)
            x2 = torch.empty((2 * N, 4), device=pos.device, dtype=torch.float32)
            _lib.call("emer_flow_warp_fwd", _ptr(pos), _ptr(nrm), _ptr(t), _ptr(fl), _ptr(nz), float(time_diff), _ptr(ab), int(unbounded), _ptr(x3),
                      _ptr(x2), N, _stream(pos))
        ctx.save_for_backward(pos, fl, nz, ab)
        ctx.unbounded, ctx.flow_shape = bool(unbounded), flow.shape
        return x3, x2

    @staticmethod
    def backward(ctx, dx3: Optional[Tensor], dx2: Optional[Tensor]):
        pos, fl, nz, ab = ctx.saved_tensors
        if not ctx.needs_input_grad[3] or (dx3 is None and dx2 is None):
            return (None,) * 8
        N = pos.shape[0]
        d3 = None if dx3 is None else _f32c(dx3)
        d2 = None if dx2 is None else _f32c(dx2)
        with torch.cuda.device(pos.device):
            dflow = torch.empty((N, 6), device=pos.device, dtype=torch.float32)
            _lib.call("emer_flow_warp_bwd", _ptr(pos), _ptr(fl), _ptr(nz), _ptr(ab), int(ctx.unbounded), _ptr(d3), _ptr(d2), _ptr(dflow), N, _stream(pos))
        return None, None, None, dflow.view(ctx.flow_shape), None, None, None, None


def flow_warp(positions: Tensor, normed: Tensor, timestamps: Tensor, flow: Tensor, noise: Tensor, time_diff: float, aabb: Tensor,
              unbounded: bool) -> Tuple[Tensor, Tensor]:
    """Query points of the flow branch's three xyzt evaluations in one launch: x3 = [current | forward-warped | backward-warped] (the
    dynamic table's batch) and x2 = its last two thirds (the flow table's batch)."""
    _check_cuda(positions, normed, timestamps, flow, noise, aabb)
    return _FlowWarpFn.apply(positions, normed, timestamps, flow, noise, time_diff, aabb, unbounded)


def ray_points(origins: Tensor, dirs: Tensor, t_starts: Tensor, t_ends: Tensor, aabb: Tensor, unbounded: bool,
               times: Optional[Tensor] = None, want_positions: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """Sample positions along rays, contracted (no grad: sample positions never carry grad,
    nerfacc_prop_net.py:89).  Returns (normed [R,S,3|4], positions [R,S,3] | None)."""
    _check_cuda(origins, dirs, t_starts, t_ends, aabb)
    R, S = t_starts.shape
    o, d, ts, te, ab = _f32c(origins), _f32c(dirs), _f32c(t_starts), _f32c(t_ends), _f32c(aabb).view(-1)
    tm = None if times is None else _f32c(times).view(-1)
    od = 3 if tm is None else 4
    with torch.cuda.device(o.device):
        normed = torch.empty((R, S, od), device=o.device, dtype=torch.float32)
        pos = torch.empty((R, S, 3), device=o.device, dtype=torch.float32) if want_positions else None
        _lib.call("emer_ray_points", _ptr(o), _ptr(d), _ptr(ts), _ptr(te), _ptr(tm), _ptr(ab), int(unbounded), _ptr(normed),
                  od, _ptr(pos), R, S, _stream(o))
    return normed, pos


# ---------------------------------------------------------------------------------- sampler
SAMPLE_POINTS = os.environ.get("EMER_FUSE_SAMPLE_POINTS", "1") != "0"   # [r5] sampler + sample points in one launch (importance_sample(points=...))
_SAMPLE_POINTS_CAP = None


def sample_points_capacity() -> int:
    """Floats of LDS per ray the fused sampler + points launch may use (2 m + n + 1 must fit): asked from the device once
    (emer_importance_sample_points_capacity; 10240 on gfx950)."""
    global _SAMPLE_POINTS_CAP
    if _SAMPLE_POINTS_CAP is None:
        _SAMPLE_POINTS_CAP = int(_lib.load().emer_importance_sample_points_capacity())
    return _SAMPLE_POINTS_CAP


def importance_sample(vals: Tensor, cdfs: Tensor, n_intervals: int, jitter: Optional[Tensor] = None,
                      stot: Optional[Tuple[float, float, str]] = None, intervals: bool = False, points=None):
    """nerfacc.pdf.importance_sampling (batched).  Returns (s_edges [R,n+1], t_edges | None), or with
    ``intervals=True`` (s_edges, t_starts [R,n], t_ends [R,n]) written directly by the kernel (no slicing copies).
    ``points=(origins, dirs, aabb, unbounded, want_positions)`` with ``intervals=True``: the launch also computes the new intervals' sample
    points (what ``ray_points`` would from its result, bitwise) and returns (s_edges, t_starts, t_ends, normed [R,n,3], positions | None)."""
    _check_cuda(vals, cdfs)
    v, c = _f32c(vals), _f32c(cdfs)
    R, m = v.shape
    j = None if jitter is None else _f32c(jitter).view(-1)
    if j is not None:
        assert j.numel() == R
    assert not intervals or stot is not None
    if points is not None:
        assert intervals, "points are computed for the interval form"
        origins, dirs, aabb, unbounded, want_pos = points
        o, d, ab = _f32c(origins), _f32c(dirs), _f32c(aabb).view(-1)
        assert o.shape == (R, 3) and d.shape == (R, 3)
        with torch.cuda.device(v.device):
            s_out = torch.empty((R, n_intervals + 1), device=v.device, dtype=torch.float32)
            t_out = torch.empty((R, n_intervals), device=v.device, dtype=torch.float32)
            t_end = torch.empty_like(t_out)
            normed = torch.empty((R, n_intervals, 3), device=v.device, dtype=torch.float32)
            pos = torch.empty((R, n_intervals, 3), device=v.device, dtype=torch.float32) if want_pos else None
            t_min, t_max, typ = stot
            _lib.call("emer_importance_sample_points", _ptr(v), _ptr(c), R, m, n_intervals, _ptr(j), _ptr(s_out), _ptr(t_out), _ptr(t_end),
                      float(t_min), float(t_max), STOT_TYPES[typ], _ptr(o), _ptr(d), _ptr(ab), int(unbounded), _ptr(normed), _ptr(pos), _stream(v))
        return s_out, t_out, t_end, normed, pos
    with torch.cuda.device(v.device):
        s_out = torch.empty((R, n_intervals + 1), device=v.device, dtype=torch.float32)
        if intervals:
            t_out = torch.empty((R, n_intervals), device=v.device, dtype=torch.float32)
            t_end = torch.empty_like(t_out)
        else:
            t_out, t_end = (torch.empty_like(s_out) if stot is not None else None), None
        t_min, t_max, typ = stot if stot is not None else (0.0, 1.0, "uniform")
        _lib.call("emer_importance_sample", _ptr(v), _ptr(c), R, m, n_intervals, _ptr(j), _ptr(s_out), _ptr(t_out), _ptr(t_end),
                  float(t_min), float(t_max), STOT_TYPES[typ], _stream(v))
    return (s_out, t_out, t_end) if intervals else (s_out, t_out)


def stot(s: Tensor, t_min: float, t_max: float, transform_type: str) -> Tensor:
    _check_cuda(s)
    sc = _f32c(s)
    with torch.cuda.device(sc.device):
        t = torch.empty_like(sc)
        _lib.call("emer_stot", _ptr(sc), sc.numel(), float(t_min), float(t_max), STOT_TYPES[transform_type], _ptr(t),
                  _stream(sc))
    return t


# ------------------------------------------------------------------------- proposal supervision
class _PropLevelLossFn(torch.autograd.Function):
    """One proposal level of PropNetEstimator.compute_loss (nerfacc_prop_net.py:181-238): scalar loss (already
    multiplied by ``scale``) with its gradient w.r.t. the level's cdf computed in the same launch."""

    @staticmethod
    def forward(ctx, s_final: Tensor, trans: Tensor, s_prop: Tensor, cdf_prop: Tensor, pulse: float, anti_aliased: bool,
                scale: float):
        sf, tr, sp, cp = _f32c(s_final), _f32c(trans), _f32c(s_prop), _f32c(cdf_prop)
        R, n = tr.shape
        m = cp.shape[1] - 1
        assert sf.shape == (R, n + 1) and sp.shape == (R, m + 1)
        want_grad = ctx.needs_input_grad[3]
        with torch.cuda.device(tr.device):
            rays = torch.empty((R,), device=tr.device, dtype=torch.float32)
            loss = torch.empty((), device=tr.device, dtype=torch.float32)
            dcp = torch.empty_like(cp) if want_grad else None
            _lib.call("emer_prop_loss", _ptr(sf), _ptr(tr), n, _ptr(sp), _ptr(cp), m, float(pulse), int(anti_aliased), R, float(scale),
                      _ptr(rays), _ptr(loss), 0, _ptr(dcp), _stream(tr))
        ctx.save_for_backward(dcp)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        (dcp,) = ctx.saved_tensors
        if dcp is None:
            return (None,) * 7
        gc = _f32c(g).reshape(1)
        with torch.cuda.device(dcp.device):
            out = torch.empty_like(dcp)
            _lib.call("emer_scale", _ptr(dcp), _ptr(gc), 1.0, _ptr(out), dcp.numel(), _stream(dcp))
        return None, None, None, out, None, None, None


def prop_level_loss(s_final: Tensor, trans: Tensor, s_prop: Tensor, cdf_prop: Tensor, pulse: float, anti_aliased: bool,
                    scale: float) -> Tensor:
    """scale * sum over rays and intervals of the interlevel loss of one proposal level (0-dim tensor)."""
    _check_cuda(s_final, trans, s_prop, cdf_prop)
    return _PropLevelLossFn.apply(s_final, trans, s_prop, cdf_prop, pulse, anti_aliased, scale)


# ------------------------------------------------------------------------------- compositing
class _RenderWeightsFn(torch.autograd.Function):
    """(weights, trans, alphas, cdfs, ray_stats[, t_mid, t_dist]) from density; grads flow to sigma only (from every
    differentiable output)."""

    @staticmethod
    def forward(ctx, t_starts: Tensor, t_ends: Tensor, sigma: Tensor, want_t: bool):
        ctx.set_materialize_grads(False)
        ts, te, sg = _f32c(t_starts), _f32c(t_ends), _f32c(sigma)
        R, S = sg.shape
        with torch.cuda.device(sg.device):
            w, T, a = (torch.empty_like(sg) for _ in range(3))
            cdfs = torch.empty((R, S + 1), device=sg.device, dtype=torch.float32)
            stats = torch.empty((R, 4), device=sg.device, dtype=torch.float32)
            tm, td = (torch.empty_like(sg), torch.empty_like(sg)) if want_t else (None, None)
            _lib.call("emer_render_weights_fwd", _ptr(ts), _ptr(te), _ptr(sg), R, S, _ptr(w), _ptr(T), _ptr(a), _ptr(cdfs),
                      _ptr(stats), _ptr(tm), _ptr(td), _stream(sg))
        ctx.save_for_backward(ts, te, sg)
        if want_t:
            ctx.mark_non_differentiable(tm, td)
            return w, T, a, cdfs, stats, tm, td
        return w, T, a, cdfs, stats

    @staticmethod
    def backward(ctx, dw, dT, da, dcdfs, dstats, *_unused):
        ts, te, sg = ctx.saved_tensors
        R, S = sg.shape
        gT = None
        if dT is not None:
            gT = _f32c(dT)
        if dcdfs is not None:  # cdfs = 1 - [T, 0]
            g = -_f32c(dcdfs)[:, :S]
            gT = g.contiguous() if gT is None else gT + g
        gw = None if dw is None else _f32c(dw)
        ga = None if da is None else _f32c(da)
        gs = None if dstats is None else _f32c(dstats)  # [R,4]; the kernel reads columns 0, 1
        if gw is None and gT is None and gs is None and ga is None:
            return None, None, None, None
        with torch.cuda.device(sg.device):
            dsig = torch.empty_like(sg)
            _lib.call("emer_render_weights_bwd", _ptr(ts), _ptr(te), _ptr(sg), _ptr(gw), _ptr(gT), _ptr(ga), _ptr(gs), R, S, _ptr(dsig),
                      _stream(sg))
        return None, None, dsig, None


def render_weights(t_starts: Tensor, t_ends: Tensor, sigma: Tensor, want_t: bool = False):
    """Returns (weights, trans, alphas, cdfs [R,S+1], ray_stats [R,4] = (sum w, sum w*mid, median_depth, 0)) and, with
    ``want_t``, also (t_mid, t_dist) = ((t_starts + t_ends) / 2, t_ends - t_starts) from the same launch."""
    _check_cuda(t_starts, t_ends, sigma)
    return _RenderWeightsFn.apply(t_starts, t_ends, sigma, want_t)


class _CompositeRgbFn(torch.autograd.Function):
    """[r4] ``rendering`` of a model with one density and one colour per sample (render_utils.py:73-122,158-159,217-220) as one launch
    each way: (weights, trans, t_vals, t_dist, opacity [R,1], depth [R,1], median_depth [R,1], rgb [R,3] | None).  Same values as
    render_weights -> accumulate_along_rays -> ray_epilogue (bitwise: tests/test_kernels_gpu.py); ``rgb`` None: geometry only."""

    @staticmethod
    def forward(ctx, t_starts: Tensor, t_ends: Tensor, sigma: Tensor, rgb: Optional[Tensor], rgb_sky: Optional[Tensor]):
        ctx.set_materialize_grads(False)
        ts, te, sg = _f32c(t_starts), _f32c(t_ends), _f32c(sigma)
        R, S = sg.shape
        c = None if rgb is None else _f32c(rgb).reshape(R, S, 3)
        sk = None if (rgb_sky is None or rgb is None) else _f32c(rgb_sky).reshape(R, 3)
        with torch.cuda.device(sg.device):
            w, T, tm, td = (torch.empty_like(sg) for _ in range(4))
            stats = torch.empty((R, 4), device=sg.device, dtype=torch.float32)
            opa, dep, med = (torch.empty((R, 1), device=sg.device, dtype=torch.float32) for _ in range(3))
            out = torch.empty((R, 3), device=sg.device, dtype=torch.float32) if c is not None else None
            _lib.call("emer_composite_rgb_fwd", _ptr(ts), _ptr(te), _ptr(sg), _ptr(c), _ptr(sk), R, S, _ptr(w), _ptr(T), _ptr(tm), _ptr(td),
                      _ptr(stats), _ptr(opa), _ptr(dep), _ptr(med), _ptr(out), _stream(sg))
        ctx.save_for_backward(ts, te, sg, c, sk, w, stats)
        ctx.mark_non_differentiable(tm, td, med)
        return w, T, tm, td, opa, dep, med, out

    @staticmethod
    def backward(ctx, dw, dT, _dtm, _dtd, dopa, ddep, _dmed, dout):
        ts, te, sg, c, sk, w, stats = ctx.saved_tensors
        R, S = sg.shape
        if all(g is None for g in (dw, dT, dopa, ddep, dout)):
            return None, None, None, None, None
        f = lambda g: None if g is None else _f32c(g)
        gw, gT, go, gd, gr = f(dw), f(dT), f(dopa), f(ddep), f(dout)
        need_rgb = c is not None and ctx.needs_input_grad[3] and gr is not None
        need_sky = sk is not None and ctx.needs_input_grad[4] and gr is not None
        with torch.cuda.device(sg.device):
            dsig = torch.empty_like(sg)
            drgb = torch.empty_like(c) if need_rgb else None
            dsky = torch.empty_like(sk) if need_sky else None
            _lib.call("emer_composite_rgb_bwd", _ptr(ts), _ptr(te), _ptr(sg), _ptr(c), _ptr(sk), _ptr(w), _ptr(stats), _ptr(gr), _ptr(go),
                      _ptr(gd), _ptr(gw), _ptr(gT), R, S, _ptr(dsig), _ptr(drgb), _ptr(dsky), _stream(sg))
        if c is not None and ctx.needs_input_grad[3] and drgb is None:
            drgb = torch.zeros_like(c)
        return None, None, (dsig if ctx.needs_input_grad[2] else None), drgb, dsky


# one launch each way for the static model's rendering (EMER_FUSE_COMPOSITE=0: the three kernels of rounds 1-3)
FUSE_COMPOSITE = os.environ.get("EMER_FUSE_COMPOSITE", "1") != "0"


def composite_rgb(t_starts: Tensor, t_ends: Tensor, sigma: Tensor, rgb: Optional[Tensor], rgb_sky: Optional[Tensor]):
    _check_cuda(t_starts, t_ends, sigma)
    return _CompositeRgbFn.apply(t_starts, t_ends, sigma, rgb, rgb_sky)


class _AccumulateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights: Tensor, values: Optional[Tensor]):
        w = _f32c(weights)
        R, S = w.shape
        v = None if values is None else _f32c(values)
        C = 1 if v is None else v.shape[-1]
        with torch.cuda.device(w.device):
            out = torch.empty((R, C), device=w.device, dtype=torch.float32)
            _lib.call("emer_accumulate_fwd", _ptr(w), _ptr(v), R, S, C, _ptr(out), _stream(w))
        ctx.save_for_backward(w, v)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        w, v = ctx.saved_tensors
        R, S = w.shape
        C = 1 if v is None else v.shape[-1]
        g = _f32c(dout)
        need_w, need_v = ctx.needs_input_grad[0], (v is not None and ctx.needs_input_grad[1])
        with torch.cuda.device(w.device):
            dw = torch.empty_like(w) if need_w else None
            dv = torch.empty_like(v) if need_v else None
            if need_w or need_v:
                _lib.call("emer_accumulate_bwd", _ptr(w), _ptr(v), _ptr(g), R, S, C, _ptr(dw), _ptr(dv), _stream(w))
        return dw, dv


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None) -> Tensor:
    """nerfacc.accumulate_along_rays on dense (R,S)/(R,S,C) tensors -> (R,C)."""
    _check_cuda(weights, values)
    return _AccumulateFn.apply(weights, values)


class _RayEpilogueFn(torch.autograd.Function):
    """(opacity [R,1], depth [R,1], median_depth [R,1], rgb [R,3] | None) from the scan kernel's per-ray sums
    (render_utils.py:102-122,217-226)."""

    @staticmethod
    def forward(ctx, stats: Tensor, acc_rgb: Optional[Tensor], rgb_sky: Optional[Tensor]):
        ctx.set_materialize_grads(False)
        st = _f32c(stats)
        R = st.shape[0]
        ar = None if acc_rgb is None else _f32c(acc_rgb)
        sk = None if rgb_sky is None else _f32c(rgb_sky)
        with torch.cuda.device(st.device):
            opa, dep, med = (torch.empty((R, 1), device=st.device, dtype=torch.float32) for _ in range(3))
            rgb = torch.empty((R, 3), device=st.device, dtype=torch.float32) if ar is not None else None
            _lib.call("emer_ray_epilogue_fwd", _ptr(st), _ptr(ar), _ptr(sk), R, _ptr(opa), _ptr(dep), _ptr(med), _ptr(rgb), _stream(st))
        ctx.save_for_backward(st, sk)
        ctx.has_rgb = ar is not None
        ctx.mark_non_differentiable(med)
        return opa, dep, med, rgb

    @staticmethod
    def backward(ctx, do, dd, dm, drgb):
        st, sk = ctx.saved_tensors
        R = st.shape[0]
        if do is None and dd is None and drgb is None:
            return None, None, None
        go = None if do is None else _f32c(do)
        gd = None if dd is None else _f32c(dd)
        gr = None if drgb is None else _f32c(drgb)
        with torch.cuda.device(st.device):
            dst = torch.empty((R, 4), device=st.device, dtype=torch.float32)
            dsk = torch.empty((R, 3), device=st.device, dtype=torch.float32) if (sk is not None and gr is not None) else None
            _lib.call("emer_ray_epilogue_bwd", _ptr(st), _ptr(sk), _ptr(go), _ptr(gd), _ptr(gr), R, _ptr(dst), _ptr(dsk), _stream(st))
        return dst, (gr if ctx.has_rgb else None), dsk


def ray_epilogue(stats: Tensor, acc_rgb: Optional[Tensor] = None, rgb_sky: Optional[Tensor] = None):
    """Per-ray tail of `rendering`: opacity = clamp(sum w, 1e-6, 1), depth = sum(w t) / opacity, median depth, and
    rgb = acc_rgb + rgb_sky * (1 - opacity).  Returns (opacity, depth, median_depth, rgb)."""
    _check_cuda(stats, acc_rgb, rgb_sky)
    return _RayEpilogueFn.apply(stats, acc_rgb, rgb_sky)


class _PixelLossFn(torch.autograd.Function):
    """w_rgb * mse(rgb, pixels) + w_sky * bce(opacity, 1 - sky_mask) (loss/base.py:83-185), one launch each way."""

    @staticmethod
    def forward(ctx, rgb: Tensor, opacity: Optional[Tensor], pixels: Tensor, sky_mask: Optional[Tensor], w_rgb: float, w_sky: float,
                grad_scale: float = 1.0):
        r, px = _f32c(rgb).view(-1, 3), _f32c(pixels).view(-1, 3)
        R = r.shape[0]
        op = None if opacity is None else _f32c(opacity).view(-1)
        sm = None if sky_mask is None else _f32c(sky_mask).view(-1)
        with torch.cuda.device(r.device):
            rays = torch.empty((R,), device=r.device, dtype=torch.float32)
            loss = torch.empty((), device=r.device, dtype=torch.float32)
            _lib.call("emer_pixel_loss_fwd", _ptr(r), _ptr(px), _ptr(op), _ptr(sm), R, float(w_rgb), float(w_sky), _ptr(rays), _ptr(loss),
                      _stream(r))
        ctx.save_for_backward(r, px, op, sm)
        ctx.w, ctx.shapes = (float(w_rgb) * float(grad_scale), float(w_sky) * float(grad_scale)), (rgb.shape, None if opacity is None else opacity.shape)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        r, px, op, sm = ctx.saved_tensors
        R = r.shape[0]
        need_o = op is not None and ctx.needs_input_grad[1]
        gc = _f32c(g).reshape(1)
        with torch.cuda.device(r.device):
            dr = torch.empty_like(r) if ctx.needs_input_grad[0] else None
            do = torch.empty_like(op) if need_o else None
            _lib.call("emer_pixel_loss_bwd", _ptr(r), _ptr(px), _ptr(op), _ptr(sm), R, ctx.w[0], ctx.w[1], _ptr(gc), _ptr(dr), _ptr(do),
                      _stream(r))
        return (None if dr is None else dr.view(ctx.shapes[0])), (None if do is None else do.view(ctx.shapes[1])), None, None, None, None, None


def pixel_loss(rgb: Tensor, opacity: Optional[Tensor], pixels: Tensor, sky_mask: Optional[Tensor], w_rgb: float = 1.0,
               w_sky: float = 0.001, grad_scale: float = 1.0) -> Tensor:
    """rgb L2 + opacity-based sky BCE of a pixel-ray batch as one scalar (0-dim tensor).  ``grad_scale`` multiplies the
    GRADIENTS only (a trainer's loss scale folded into the backward kernel: the returned value stays the plain loss and
    no ``loss * scale`` launch, nor its backward, is needed)."""
    _check_cuda(rgb, opacity, pixels, sky_mask)
    return _PixelLossFn.apply(rgb, opacity, pixels, sky_mask, w_rgb, w_sky, grad_scale)


REG_MAX_BLOCKS = 1024  # EMER_REG_MAX_BLOCKS


class _RegLossesFn(torch.autograd.Function):
    """base + the mean-type regularisers (dynamic-density / shadow sparsity, feature L2, flow cycle) as one launch each way."""

    @staticmethod
    def forward(ctx, base: Optional[Tensor], dyn: Optional[Tensor], shadow: Optional[Tensor], feat: Optional[Tensor], feat_gt: Optional[Tensor],
                ff: Optional[Tensor], fpb: Optional[Tensor], bf: Optional[Tensor], bpf: Optional[Tensor], coefs, grad_scale: float):
        c_dyn, c_shadow, c_feat, c_cycle = (float(c) for c in coefs)
        ts = [None if t is None else _f32c(t) for t in (dyn, shadow, feat, feat_gt, ff, fpb, bf, bpf)]
        dyn_c, sh_c, ft_c, gt_c, ff_c, fpb_c, bf_c, bpf_c = ts
        ref = next(t for t in ts if t is not None)
        if ft_c is not None:
            assert gt_c is not None and gt_c.numel() == ft_c.numel(), "feature loss: prediction / target size mismatch"
        if fpb_c is not None:
            assert all(t is not None and t.numel() == fpb_c.numel() for t in (ff_c, bf_c, bpf_c)), "flow cycle loss: size mismatch"
        n = lambda t: 0 if t is None else t.numel()
        bc = None if base is None else _f32c(base).reshape(1)
        with torch.cuda.device(ref.device):
            ws = torch.empty((REG_MAX_BLOCKS,), device=ref.device, dtype=torch.float32)
            loss = torch.empty((), device=ref.device, dtype=torch.float32)
            ctx.args = (n(dyn_c), c_dyn, n(sh_c), c_shadow, n(ft_c), c_feat, n(fpb_c), c_cycle)
            _lib.call("emer_reg_losses_fwd", _ptr(dyn_c), n(dyn_c), c_dyn, _ptr(sh_c), n(sh_c), c_shadow, _ptr(ft_c), _ptr(gt_c), n(ft_c), c_feat,
                      _ptr(ff_c), _ptr(fpb_c), _ptr(bf_c), _ptr(bpf_c), n(fpb_c), c_cycle, _ptr(bc), _ptr(ws), _ptr(loss), _stream(ref))
        # the sparsity terms' gradients are constants: only the tensors a gradient READS are saved
        ctx.save_for_backward(ft_c, gt_c, ff_c, fpb_c, bf_c, bpf_c)
        ctx.present = (dyn_c is not None, sh_c is not None)
        ctx.shapes = tuple(None if t is None else t.shape for t in (dyn, shadow, feat, fpb, bpf))
        ctx.grad_scale, ctx.dev = float(grad_scale), ref.device
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        ft_c, gt_c, ff_c, fpb_c, bf_c, bpf_c = ctx.saved_tensors
        n_dyn, c_dyn, n_sh, c_shadow, n_ft, c_feat, n_fl, c_cycle = ctx.args
        ni = ctx.needs_input_grad
        gc = _f32c(g).reshape(1)
        with torch.cuda.device(ctx.dev):
            new = lambda shape: torch.empty(shape, device=ctx.dev, dtype=torch.float32)
            d_dyn = new(ctx.shapes[0]) if ctx.present[0] and ni[1] else None
            d_sh = new(ctx.shapes[1]) if ctx.present[1] and ni[2] else None
            d_ft = new(ctx.shapes[2]) if ft_c is not None and ni[3] else None
            d_fpb = new(ctx.shapes[3]) if fpb_c is not None and ni[6] else None
            d_bpf = new(ctx.shapes[4]) if bpf_c is not None and ni[8] else None
            if any(t is not None for t in (d_dyn, d_sh, d_ft, d_fpb, d_bpf)):
                # absent-term pointers: the sparsity terms are described by their counts alone (gradient = constant)
                _lib.call("emer_reg_losses_bwd", _ptr(d_dyn), n_dyn if d_dyn is not None else 0, c_dyn, _ptr(d_sh), n_sh if d_sh is not None else 0,
                          c_shadow, _ptr(ft_c), _ptr(gt_c), n_ft, c_feat, _ptr(ff_c), _ptr(fpb_c), _ptr(bf_c), _ptr(bpf_c), n_fl, c_cycle,
                          _ptr(gc), ctx.grad_scale, _ptr(d_dyn), _ptr(d_sh), _ptr(d_ft), _ptr(d_fpb), _ptr(d_bpf), _stream(gc))
        gb = g if ni[0] else None
        return gb, d_dyn, d_sh, d_ft, None, None, d_fpb, None, d_bpf, None, None


class _RegLosses6Fn(torch.autograd.Function):
    """_RegLossesFn with the cycle term read from the flow MLP's own outputs: ``flow`` [N, 6] at the sample positions (constants, as the
    reference detaches them) and ``flow2`` [2 N, 6] at the two warped sets -- no slice copies forward, ONE gradient tensor backward
    (the four slices cost four contiguous copies, two zero fills, two copies and a cat per step)."""

    @staticmethod
    def forward(ctx, base: Optional[Tensor], dyn: Optional[Tensor], shadow: Optional[Tensor], feat: Optional[Tensor], feat_gt: Optional[Tensor],
                flow: Tensor, flow2: Tensor, coefs, grad_scale: float):
        c_dyn, c_shadow, c_feat, c_cycle = (float(c) for c in coefs)
        dyn_c, sh_c, ft_c, gt_c = (None if t is None else _f32c(t) for t in (dyn, shadow, feat, feat_gt))
        f6, f26 = _f32c(flow).view(-1, 6), _f32c(flow2).view(-1, 6)
        assert f26.shape[0] == 2 * f6.shape[0], "flow cycle loss: the warped evaluations must hold two rows per sample"
        if ft_c is not None:
            assert gt_c is not None and gt_c.numel() == ft_c.numel(), "feature loss: prediction / target size mismatch"
        n = lambda t: 0 if t is None else t.numel()
        bc = None if base is None else _f32c(base).reshape(1)
        dev = f6.device
        with torch.cuda.device(dev):
            ws = torch.empty((REG_MAX_BLOCKS,), device=dev, dtype=torch.float32)
            loss = torch.empty((), device=dev, dtype=torch.float32)
            ctx.args = (n(dyn_c), c_dyn, n(sh_c), c_shadow, n(ft_c), c_feat, f6.shape[0], c_cycle)
            _lib.call("emer_reg_losses_fwd6", _ptr(dyn_c), n(dyn_c), c_dyn, _ptr(sh_c), n(sh_c), c_shadow, _ptr(ft_c), _ptr(gt_c), n(ft_c), c_feat,
                      _ptr(f6), _ptr(f26), f6.shape[0], c_cycle, _ptr(bc), _ptr(ws), _ptr(loss), _stream(f6))
        ctx.save_for_backward(ft_c, gt_c, f6, f26)
        ctx.present = (dyn_c is not None, sh_c is not None)
        ctx.shapes = tuple(None if t is None else t.shape for t in (dyn, shadow, feat, flow2))
        ctx.grad_scale, ctx.dev = float(grad_scale), dev
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        ft_c, gt_c, f6, f26 = ctx.saved_tensors
        n_dyn, c_dyn, n_sh, c_shadow, n_ft, c_feat, n_rows, c_cycle = ctx.args
        ni = ctx.needs_input_grad
        gc = _f32c(g).reshape(1)
        with torch.cuda.device(ctx.dev):
            new = lambda shape: torch.empty(shape, device=ctx.dev, dtype=torch.float32)
            d_dyn = new(ctx.shapes[0]) if ctx.present[0] and ni[1] else None
            d_sh = new(ctx.shapes[1]) if ctx.present[1] and ni[2] else None
            d_ft = new(ctx.shapes[2]) if ft_c is not None and ni[3] else None
            d_f2 = new(ctx.shapes[3]) if ni[6] else None
            if any(t is not None for t in (d_dyn, d_sh, d_ft, d_f2)):
                _lib.call("emer_reg_losses_bwd6", _ptr(d_dyn), n_dyn if d_dyn is not None else 0, c_dyn, _ptr(d_sh), n_sh if d_sh is not None else 0,
                          c_shadow, _ptr(ft_c), _ptr(gt_c), n_ft, c_feat, _ptr(f6), _ptr(f26), n_rows, c_cycle, _ptr(gc), ctx.grad_scale,
                          _ptr(d_dyn), _ptr(d_sh), _ptr(d_ft), _ptr(d_f2), _stream(gc))
        gb = g if ni[0] else None
        return gb, d_dyn, d_sh, d_ft, None, None, d_f2, None, None


def reg_losses(base: Optional[Tensor] = None, dynamic_density: Optional[Tensor] = None, shadow_ratio: Optional[Tensor] = None,
               feat: Optional[Tensor] = None, feat_gt: Optional[Tensor] = None, forward_flow: Optional[Tensor] = None,
               forward_pred_backward_flow: Optional[Tensor] = None, backward_flow: Optional[Tensor] = None,
               backward_pred_forward_flow: Optional[Tensor] = None, c_dyn: float = 0.01, c_shadow: float = 0.01, c_feat: float = 0.5,
               c_cycle: float = 0.005, grad_scale: float = 1.0, flow_pair=None) -> Tensor:
    """``base + c_dyn mean(dynamic_density) + c_shadow mean(shadow_ratio) + c_feat mse(feat, feat_gt) + c_cycle mean((ff + fpb)^2 +
    (bf + bpf)^2)`` as a 0-dim tensor: loss/base.py:394-398 (sparsity), :83-146 (feature L2), train_emernerf.py:700-716 (cycle;
    forward_flow / backward_flow are constants, as the reference detaches them).  Absent terms are None.  ``grad_scale`` multiplies
    the regularisers' GRADIENTS only (``base`` passes its gradient through unscaled: its own kernel already folded the scale)."""
    if flow_pair is not None:
        # [r5] (flow at the samples [., 6], flow at the two warped sets [2 N, 6]): the four flow arguments are column blocks of this pair
        flow, flow2 = flow_pair
        _check_cuda(*[t for t in (base, dynamic_density, shadow_ratio, feat, feat_gt, flow, flow2) if t is not None])
        return _RegLosses6Fn.apply(base, dynamic_density, shadow_ratio, feat, feat_gt, flow.detach(), flow2, (c_dyn, c_shadow, c_feat, c_cycle), grad_scale)
    present = [t for t in (dynamic_density, shadow_ratio, feat, forward_pred_backward_flow) if t is not None]
    if not present:
        if base is None:
            raise ValueError("reg_losses: no term given")
        return base
    _check_cuda(*[t for t in (base, dynamic_density, shadow_ratio, feat, feat_gt, forward_flow, forward_pred_backward_flow, backward_flow,
                              backward_pred_forward_flow) if t is not None])
    return _RegLossesFn.apply(base, dynamic_density, shadow_ratio, feat, feat_gt, forward_flow, forward_pred_backward_flow, backward_flow,
                              backward_pred_forward_flow, (c_dyn, c_shadow, c_feat, c_cycle), grad_scale)


class _LidarLossFn(torch.autograd.Function):
    """w_depth * depth loss + w_sight * line-of-sight loss of a lidar-ray batch (train_emernerf.py:770-808)."""

    @staticmethod
    def forward(ctx, depth: Tensor, weights: Tensor, ranges: Tensor, t_vals: Tensor, epsilon: float, max_depth: float,
                w_depth: float, w_sight: float):
        dp, w, rg, tv = _f32c(depth).view(-1), _f32c(weights), _f32c(ranges).view(-1), _f32c(t_vals)
        R, S = w.shape
        with torch.cuda.device(w.device):
            ws = torch.empty((R + 2,), device=w.device, dtype=torch.float32)
            loss = torch.empty((), device=w.device, dtype=torch.float32)
            _lib.call("emer_lidar_loss", _ptr(dp), _ptr(rg), _ptr(w), _ptr(tv), R, S, float(epsilon), float(max_depth), float(w_depth),
                      float(w_sight), None, _ptr(ws), _ptr(loss), None, None, _stream(w))
        ctx.save_for_backward(dp, w, rg, tv)
        ctx.cfg, ctx.dshape = (float(epsilon), float(max_depth), float(w_depth), float(w_sight)), depth.shape
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        dp, w, rg, tv = ctx.saved_tensors
        R, S = w.shape
        eps, md, wd, wsg = ctx.cfg
        gc = _f32c(g).reshape(1)
        with torch.cuda.device(w.device):
            ws = torch.empty((R + 2,), device=w.device, dtype=torch.float32)
            dd = torch.empty((R,), device=w.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
            dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
            _lib.call("emer_lidar_loss", _ptr(dp), _ptr(rg), _ptr(w), _ptr(tv), R, S, eps, md, wd, wsg, _ptr(gc), _ptr(ws), None, _ptr(dd),
                      _ptr(dw), _stream(w))
        return (None if dd is None else dd.view(ctx.dshape)), dw, None, None, None, None, None, None


def lidar_loss(depth: Tensor, weights: Tensor, lidar_ranges: Tensor, t_vals: Tensor, epsilon: float, max_depth: float = 80.0,
               w_depth: float = 1.0, w_sight: float = 0.1) -> Tensor:
    """Depth (loss/base.py:188-271, l2) + line-of-sight (loss/base.py:430-464) supervision as one scalar; gradients flow
    to the rendered depth [R,1] and the rendering weights [R,S]."""
    _check_cuda(depth, weights, lidar_ranges, t_vals)
    return _LidarLossFn.apply(depth, weights, lidar_ranges, t_vals, epsilon, max_depth, w_depth, w_sight)


# ---------------------------------------------------------------------------------- MLP heads
class _LinearFn(torch.autograd.Function):
    """act(x W^T + b) (+ optional density side output exp(pre[:,0] - 1))."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int, with_density: bool):
        ctx.set_materialize_grads(False)
        lead = x.shape[:-1]
        K = x.shape[-1]
        x2 = _f32c(x).view(-1, K)
        wc = _f32c(weight)
        bc = None if bias is None else _f32c(bias)
        M, N = x2.shape[0], wc.shape[0]
        with torch.cuda.device(x2.device):
            y = torch.empty((M, N), device=x2.device, dtype=torch.float32)
            aux = torch.empty((M,), device=x2.device, dtype=torch.float32) if with_density else None
            _lib.call("emer_linear_fwd", _ptr(x2), K, _ptr(wc), _ptr(bc), _ptr(y), N, M, N, K, act, _ptr(aux), _stream(x2))
        ctx.save_for_backward(x2, wc, y, aux)
        ctx.act, ctx.lead, ctx.has_bias = act, lead, bias is not None
        if with_density:
            return y.view(*lead, N), aux.view(*lead)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy: Optional[Tensor], daux: Optional[Tensor] = None):
        x2, wc, y, aux = ctx.saved_tensors
        M, K = x2.shape
        N = wc.shape[0]
        if dy is None and daux is None:
            return None, None, None, None, None
        g = None if dy is None else _f32c(dy).view(M, N)
        ga = None if daux is None else _f32c(daux).view(M)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        with torch.cuda.device(x2.device):
            want_w = need_w or need_b
            n_ws = int(_lib.load().emer_linear_bwd_workspace(M, N, K)) if want_w else 0
            ws = torch.empty((n_ws,), device=x2.device, dtype=torch.float32) if want_w else None
            dx = torch.empty((M, K), device=x2.device, dtype=torch.float32) if need_x else None
            dw = torch.zeros((N, K), device=x2.device, dtype=torch.float32) if (need_w or need_b) else None
            db = torch.zeros((N,), device=x2.device, dtype=torch.float32) if need_b else None
            _lib.call("emer_linear_bwd", _ptr(g), N, _ptr(y), N, _ptr(x2), K, _ptr(wc), _ptr(ws), _ptr(dx), K, _ptr(dw),
                      _ptr(db), M, N, K, ctx.act, _ptr(ga), _ptr(aux), _stream(x2))
        return (dx.view(*ctx.lead, K) if need_x else None), (dw if need_w else None), db, None, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: Optional[str] = None) -> Tensor:
    """act(x @ weight.T + bias) on the fp32 matrix cores; act in {None,'relu','sigmoid','trunc_exp'}."""
    _check_cuda(x, weight, bias)
    return _LinearFn.apply(x, weight, bias, _ACTS[act], False)


def linear_with_density(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(x @ weight.T + bias, exp(pre[..., 0] - 1)): last base-MLP layer with the density activation
    of geometry feature 0 fused into the epilogue (radiance_field.py:28,315,422)."""
    _check_cuda(x, weight, bias)
    return _LinearFn.apply(x, weight, bias, ACT_NONE, True)


class _TruncExpFn(torch.autograd.Function):
    """density = exp(x[..., col] - 1) read off one column of a feature tensor."""

    @staticmethod
    def forward(ctx, feats: Tensor, col: int):
        f = feats.detach()
        assert f.dtype == torch.float32 and f.is_contiguous()
        C = f.shape[-1]
        n = f.numel() // C
        with torch.cuda.device(f.device):
            y = torch.empty(f.shape[:-1], device=f.device, dtype=torch.float32)
            src = f.view(-1)[col:]
            _lib.call("emer_trunc_exp_fwd", _ptr(src), C, _ptr(y), n, _stream(f))
        ctx.save_for_backward(y)
        ctx.col, ctx.shape = col, feats.shape
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (y,) = ctx.saved_tensors
        C = ctx.shape[-1]
        g = _f32c(dy)
        with torch.cuda.device(y.device):
            dx = torch.zeros(ctx.shape, device=y.device, dtype=torch.float32)
            dst = dx.view(-1)[ctx.col:]
            _lib.call("emer_trunc_exp_bwd", _ptr(g), _ptr(y), _ptr(dst), C, y.numel(), _stream(y))
        return dx, None


def trunc_exp_column(feats: Tensor, col: int = 0) -> Tensor:
    """trunc_exp(feats[..., col] - 1) (radiance_field.py:28,461)."""
    _check_cuda(feats)
    return _TruncExpFn.apply(feats.contiguous(), col)


class _Aggregate3Fn(torch.autograd.Function):
    """(cur + 0.5 fwd + 0.5 bwd) / 2 of a [3 N, C] batch laid out [current | forward-warped | backward-warped]
    (temporal_aggregation, radiance_field.py:553-620) -> [N, C]; one launch each way."""

    @staticmethod
    def forward(ctx, x3: Tensor):
        x = _f32c(x3)
        n3, C = x.shape
        assert n3 % 3 == 0 and (n3 // 3 * C) % 4 == 0
        with torch.cuda.device(x.device):
            out = torch.empty((n3 // 3, C), device=x.device, dtype=torch.float32)
            _lib.call("emer_aggregate3_fwd", _ptr(x), out.numel(), _ptr(out), _stream(x))
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        gc = _f32c(g)
        with torch.cuda.device(gc.device):
            dx = torch.empty((3 * gc.shape[0], gc.shape[1]), device=gc.device, dtype=torch.float32)
            _lib.call("emer_aggregate3_bwd", _ptr(gc), gc.numel(), _ptr(dx), _stream(gc))
        return dx


def aggregate3(x3: Tensor) -> Tensor:
    _check_cuda(x3)
    return _Aggregate3Fn.apply(x3)


class _Aggregate3DensityFn(torch.autograd.Function):
    """[r4] (aggregated features [N, C], density [N] = trunc_exp(features[:, 0] - 1)) of a [3 N, C] batch, one launch each way: the
    density's gradient joins column 0 inside the backward kernel (bitwise the sum autograd would form from the separate trunc_exp)."""

    @staticmethod
    def forward(ctx, x3: Tensor):
        ctx.set_materialize_grads(False)
        x = _f32c(x3)
        n3, C = x.shape
        assert n3 % 3 == 0 and C % 4 == 0
        with torch.cuda.device(x.device):
            out = torch.empty((n3 // 3, C), device=x.device, dtype=torch.float32)
            dens = torch.empty((n3 // 3,), device=x.device, dtype=torch.float32)
            _lib.call("emer_aggregate3_density_fwd", _ptr(x), n3 // 3, C, _ptr(out), _ptr(dens), _stream(x))
        ctx.save_for_backward(dens)
        ctx.C = C
        return out, dens

    @staticmethod
    def backward(ctx, g, gd):
        (dens,) = ctx.saved_tensors
        if g is None and gd is None:
            return None
        gc = None if g is None else _f32c(g)
        gdc = None if gd is None else _f32c(gd).reshape(-1)
        n = dens.shape[0]
        with torch.cuda.device(dens.device):
            dx = torch.empty((3 * n, ctx.C), device=dens.device, dtype=torch.float32)
            _lib.call("emer_aggregate3_density_bwd", _ptr(gc), _ptr(gdc), _ptr(dens), n, ctx.C, _ptr(dx), _stream(dens))
        return dx


def aggregate3_density(x3: Tensor):
    _check_cuda(x3)
    return _Aggregate3DensityFn.apply(x3)


def dir_encode(dirs: Tensor, max_deg: int = 4, remap: bool = True) -> Tensor:
    """SinusoidalEncoder(3, 0, max_deg); remap=True applies (dirs+1)/2 first (radiance_field.py:629; no grad)."""
    _check_cuda(dirs)
    d = _f32c(dirs).view(-1, 3)
    width = 3 if max_deg == 0 else 3 * (1 + 2 * (max_deg + 1))
    with torch.cuda.device(d.device):
        out = torch.empty((d.shape[0], width), device=d.device, dtype=torch.float32)
        _lib.call("emer_dir_encode", _ptr(d), _ptr(out), d.shape[0], max_deg, int(remap), _stream(d))
    return out.view(*dirs.shape[:-1], width)


def adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float, beta2: float,
              eps: float, weight_decay: float, grad_scale: float, step: int) -> None:
    """In-place torch.optim.Adam step on one flat fp32 buffer."""
    _check_cuda(params, grads, exp_avg, exp_avg_sq)
    assert params.is_contiguous() and grads.is_contiguous() and params.dtype == torch.float32
    with torch.cuda.device(params.device):
        _lib.call("emer_adam_step", _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), float(lr),
                  float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale), int(step), _stream(params))


# ------------------------------------------------------------------ static / dynamic / shadow blend + accumulation
class _BlendAccumulateFn(torch.autograd.Function):
    """(acc_rgb [R,3], acc_shadow_sq [R,1] | None) of rendering's decomposed colour path (render_utils.py:125-175)."""

    @staticmethod
    def forward(ctx, weights, density, static_density, dynamic_density, static_rgb, dynamic_rgb, shadow_ratio):
        ctx.set_materialize_grads(False)
        w, sg, ss, sd = _f32c(weights), _f32c(density), _f32c(static_density), _f32c(dynamic_density)
        rs, rd = _f32c(static_rgb), _f32c(dynamic_rgb)
        sh = None if shadow_ratio is None else _f32c(shadow_ratio).reshape(w.shape)
        R, S = w.shape
        with torch.cuda.device(w.device):
            acc = torch.empty((R, 3), device=w.device, dtype=torch.float32)
            acs = torch.empty((R, 1), device=w.device, dtype=torch.float32) if sh is not None else None
            _lib.call("emer_blend_accumulate_fwd", _ptr(w), _ptr(sg), _ptr(ss), _ptr(sd), _ptr(rs), _ptr(rd), _ptr(sh), R, S, _ptr(acc),
                      _ptr(acs), _stream(w))
        ctx.save_for_backward(w, sg, ss, sd, rs, rd, sh)
        ctx.sh_shape = None if shadow_ratio is None else shadow_ratio.shape
        return acc, acs

    @staticmethod
    def backward(ctx, g_rgb, g_sh):
        w, sg, ss, sd, rs, rd, sh = ctx.saved_tensors
        R, S = w.shape
        if g_rgb is None and g_sh is None:
            return (None,) * 7
        gr = None if g_rgb is None else _f32c(g_rgb)
        gs = None if g_sh is None else _f32c(g_sh).view(-1)
        need = ctx.needs_input_grad
        with torch.cuda.device(w.device):
            mk = lambda t, on: torch.empty_like(t) if on else None  # noqa: E731
            dw, dsg, dss, dsd = mk(w, need[0]), mk(sg, need[1]), mk(ss, need[2]), mk(sd, need[3])
            drs, drd = mk(rs, need[4]), mk(rd, need[5])
            dsh = mk(sh, need[6]) if sh is not None else None
            _lib.call("emer_blend_accumulate_bwd", _ptr(w), _ptr(sg), _ptr(ss), _ptr(sd), _ptr(rs), _ptr(rd), _ptr(sh), _ptr(gr), _ptr(gs), R, S,
                      _ptr(dw), _ptr(dsg), _ptr(dss), _ptr(dsd), _ptr(drs), _ptr(drd), _ptr(dsh), _stream(w))
        return dw, dsg, dss, dsd, drs, drd, (None if dsh is None else dsh.view(ctx.sh_shape))


class _BlendAccumulateWideFn(torch.autograd.Function):
    """acc [R,C] of rendering's decomposed FEATURE path (render_utils.py:247-252), one launch each way."""

    @staticmethod
    def forward(ctx, weights, density, static_density, dynamic_density, static_feat, dynamic_feat):
        ctx.set_materialize_grads(False)
        w, sg, ss, sd = _f32c(weights), _f32c(density).reshape(weights.shape), _f32c(static_density).reshape(weights.shape), \
            _f32c(dynamic_density).reshape(weights.shape)
        fs, fd = _f32c(static_feat), _f32c(dynamic_feat)
        R, S = w.shape
        C = fs.shape[-1]
        with torch.cuda.device(w.device):
            acc = torch.empty((R, C), device=w.device, dtype=torch.float32)
            _lib.call("emer_blend_accumulate_wide_fwd", _ptr(w), _ptr(sg), _ptr(ss), _ptr(sd), _ptr(fs), _ptr(fd), R, S, C, _ptr(acc), _stream(w))
        ctx.save_for_backward(w, sg, ss, sd, fs, fd)
        ctx.shapes = (density.shape, static_density.shape, dynamic_density.shape)
        return acc

    @staticmethod
    def backward(ctx, g_acc):
        if g_acc is None:
            return (None,) * 6
        w, sg, ss, sd, fs, fd = ctx.saved_tensors
        R, S = w.shape
        C = fs.shape[-1]
        g = _f32c(g_acc)
        need = ctx.needs_input_grad
        with torch.cuda.device(w.device):
            mk = lambda t, on: torch.empty_like(t) if on else None  # noqa: E731
            dw, dsg, dss, dsd, dfs, dfd = mk(w, need[0]), mk(sg, need[1]), mk(ss, need[2]), mk(sd, need[3]), mk(fs, need[4]), mk(fd, need[5])
            _lib.call("emer_blend_accumulate_wide_bwd", _ptr(w), _ptr(sg), _ptr(ss), _ptr(sd), _ptr(fs), _ptr(fd), _ptr(g), R, S, C, _ptr(dw),
                      _ptr(dsg), _ptr(dss), _ptr(dsd), _ptr(dfs), _ptr(dfd), _stream(w))
        sh = ctx.shapes
        rs = lambda t, shape: None if t is None else t.view(shape)  # noqa: E731
        return dw, rs(dsg, sh[0]), rs(dss, sh[1]), rs(dsd, sh[2]), dfs, dfd


def blend_accumulate_wide(weights: Tensor, density: Tensor, static_density: Tensor, dynamic_density: Tensor, static_feat: Tensor,
                          dynamic_feat: Tensor) -> Tensor:
    """sum_s w (sigma_s / (sigma + 1e-6) feat_s + sigma_d / (sigma + 1e-6) feat_d) for [R,S,C] features -> [R,C]."""
    _check_cuda(weights, density, static_density, dynamic_density, static_feat, dynamic_feat)
    return _BlendAccumulateWideFn.apply(weights, density, static_density, dynamic_density, static_feat, dynamic_feat)


def blend_accumulate(weights: Tensor, density: Tensor, static_density: Tensor, dynamic_density: Tensor, static_rgb: Tensor,
                     dynamic_rgb: Tensor, shadow_ratio: Optional[Tensor] = None):
    """sum_s w (sigma_s / (sigma + 1e-6) rgb_s (1 - shadow) + sigma_d / (sigma + 1e-6) rgb_d) and sum_s w shadow^2."""
    _check_cuda(weights, density, static_density, dynamic_density, static_rgb, dynamic_rgb, shadow_ratio)
    return _BlendAccumulateFn.apply(weights, density, static_density, dynamic_density, static_rgb, dynamic_rgb, shadow_ratio)
