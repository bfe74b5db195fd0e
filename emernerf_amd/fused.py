"""Fused MLP heads: forward chains, dgrad chains and segmented weight gradients on ``emer_mlp_chain`` /
``emer_wgrad_segmented`` (csrc/mlp.hip).

Three autograd Functions cover the hot heads of the reference:
  * ``base_mlp``     -- nn.Sequential(Linear(K0,H), ReLU, Linear(H,NG)) + density trunc_exp(out[:,0]-1)
                        (radiance_field.py:74-80,89-96,315,422) fed straight from the LEVEL-MAJOR grid encoding;
  * ``rgb_head``     -- mlp.MLP(3 layers, skip at layer 1) + sigmoid on [dir-PE | appearance emb | geo]
                        (radiance_field.py:130-143,622-658); the per-ray part stays per ray, nothing is concatenated;
  * ``density_mlp``  -- proposal net Linear(K0,H) ReLU Linear(H,1) trunc_exp (radiance_field.py:808-812,836-840).
Activations never leave the CU inside a chain; what the backward needs (post-ReLU hidden activations) is written
once by the forward and read once as relu' masks / wgrad operands.

Two implementations sit behind each Function: the register-resident kernels of csrc/mlp_fused.hip (hidden width 64,
the shapes every shipped config uses: ``emer_neck_*``, ``emer_rgb_head_*``) and the generic LDS-staged
``emer_mlp_chain`` for any other width.  Both are HIP; there is no torch fallback.
"""
from __future__ import annotations

import contextlib
import os
import ctypes
from ctypes import c_int32, c_int64, c_void_p
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TRUNC_EXP

MAX_LAYERS, MAX_SEGS = 6, 3
E15 = 3269017.3724721107


class ChainSeg(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("ld", c_int64), ("n_total", c_int64), ("fix_a", c_void_p), ("fix_b", c_void_p),
                ("col", c_int32), ("width", c_int32), ("row_div", c_int32), ("mode", c_int32), ("f", c_int32), ("dst_col", c_int32)]


class ChainLayer(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("w_sn", c_int64), ("w_sk", c_int64), ("bias", c_void_p), ("mask", c_void_p),
                ("mask_ld", c_int64), ("store", c_void_p), ("store_ld", c_int64), ("store_ntotal", c_int64),
                ("store_exp0", c_void_p), ("in_col", c_int32), ("K", c_int32), ("out_col", c_int32), ("N", c_int32),
                ("act", c_int32), ("accumulate", c_int32), ("store_col", c_int32), ("store_n", c_int32),
                ("store_mode", c_int32), ("store_f", c_int32)]


class ChainDesc(ctypes.Structure):
    _fields_ = [("segs", ChainSeg * MAX_SEGS), ("layers", ChainLayer * MAX_LAYERS), ("n_segs", c_int32),
                ("n_layers", c_int32), ("buf_cols", c_int32), ("_pad", c_int32)]


class RayWgradJob(ctypes.Structure):
    """emer_ray_wgrad_job of include/emernerf_hip.h."""
    _fields_ = [("dy", c_void_p), ("ld_dy", c_int64), ("x", c_void_p * 2), ("ld_x", c_int64 * 2), ("dw", c_void_p), ("ld_dw", c_int64),
                ("dbias", c_void_p), ("n", c_int32), ("n_segs", c_int32), ("width", c_int32 * 2), ("dst_col", c_int32 * 2)]


def ray_wgrad(jobs, ref: Tensor) -> None:
    """Weight gradients of per-ray layers in ONE launch.  jobs: [(dy [M, n], [(x [M, >= width], width, dst_col), ...], dw, dbias | None)]
    -- dw[:, dst_col : dst_col + width] += dy^T x and dbias += colsum(dy), accumulated into the given tensors."""
    arr = (RayWgradJob * len(jobs))()
    M = jobs[0][0].shape[0]
    for a, (dy, blocks, dw, db) in zip(arr, jobs):
        assert dy.shape[0] == M and dy.stride(1) == 1 and dw.stride(1) == 1 and 1 <= len(blocks) <= 2
        a.dy, a.ld_dy, a.n, a.n_segs = dy.data_ptr(), dy.stride(0), dy.shape[1], len(blocks)
        a.dw, a.ld_dw, a.dbias = dw.data_ptr(), dw.stride(0), (None if db is None else db.data_ptr())
        for i, (x, width, dst) in enumerate(blocks):
            assert x.shape[0] == M and x.stride(1) == 1
            a.x[i], a.ld_x[i], a.width[i], a.dst_col[i] = x.data_ptr(), x.stride(0), width, dst
    with torch.cuda.device(ref.device):
        _lib.call("emer_ray_wgrad", arr, len(jobs), M, _stream(ref))


def _ng(*args):
    """Arguments for a Function.apply call outside autograd recording (torch.no_grad, set_grad_enabled(False): evaluation,
    the proposal sampling of the steps that do not train the proposal nets).  A Function's forward still sees
    needs_input_grad == requires_grad of its inputs there (and torch.is_grad_enabled() is False inside EVERY forward), so
    the inputs are detached here: the forwards then skip what only a backward needs -- hidden activations (256 B per sample),
    the grid backward's bitmaps -- instead of writing it for nobody."""
    if torch.is_grad_enabled():
        return args
    return tuple(a.detach() if isinstance(a, Tensor) else a for a in args)


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def seg(t: Tensor, col: int, width: int, ld: Optional[int] = None, row_div: int = 1, dst_col: Optional[int] = None) -> ChainSeg:
    """Row-major segment: value(row, c) = t[(row // row_div), c].  dst_col (wgrad only): where the segment's gradient
    columns land in the output matrix (default: its own position in the virtual concatenation)."""
    s = ChainSeg(ptr=_p(t), ld=ld if ld is not None else t.stride(-2), n_total=0, fix_a=None, fix_b=None, col=col,
                 width=width, row_div=row_div, mode=0, f=1, dst_col=col if dst_col is None else dst_col)
    s._t = t  # keeps the tensor reachable (side-stream bookkeeping in wgrad)
    return s


def seg_lm(t: Tensor, col: int) -> ChainSeg:
    """Level-major grid encoding [L, N, F] as L*F columns."""
    L, N, F = t.shape
    s = ChainSeg(ptr=_p(t), ld=0, n_total=N, fix_a=None, fix_b=None, col=col, width=L * F, row_div=1, mode=1, f=F, dst_col=col)
    s._t = t
    return s


def layer(w: Tensor, bias: Optional[Tensor], in_col: int, out_col: int, act: int = ACT_NONE, transposed: bool = False,
          n_slice: Optional[Tuple[int, int]] = None, mask: Optional[Tensor] = None, store: Optional[Tensor] = None,
          store_cols: Optional[Tuple[int, int]] = None, store_lm: bool = False, store_exp0: Optional[Tensor] = None,
          accumulate: bool = False) -> ChainLayer:
    """w is a torch Linear weight [out, in].  transposed=True uses W^T (dgrad): output index runs over `in`.
    n_slice=(a, b) restricts the chain layer's outputs to [a, b) of that output index."""
    out_f, in_f = w.shape
    if not transposed:
        K, n_tot, sn, sk = in_f, out_f, w.stride(0), w.stride(1)
    else:
        K, n_tot, sn, sk = out_f, in_f, w.stride(1), w.stride(0)
    a, b = n_slice if n_slice is not None else (0, n_tot)
    ptr = w.data_ptr() + 4 * a * sn
    N = b - a
    L = ChainLayer(w=ptr, w_sn=sn, w_sk=sk, bias=None if bias is None else bias.data_ptr() + 4 * a, mask=_p(mask),
                   mask_ld=0 if mask is None else mask.stride(0), store=None, store_ld=0, store_ntotal=0,
                   store_exp0=_p(store_exp0), in_col=in_col, K=K, out_col=out_col, N=N, act=act, accumulate=int(accumulate),
                   store_col=0, store_n=0, store_mode=0, store_f=1)
    if store is not None:
        L.store = store.data_ptr()
        c0, c1 = store_cols if store_cols is not None else (0, N)
        L.store_col, L.store_n = c0, c1 - c0
        if store_lm:
            Lv, Nt, F = store.shape
            L.store_mode, L.store_f, L.store_ntotal = 1, F, Nt
            assert Lv * F == L.store_n
        else:
            L.store_ld = store.stride(0)
            assert store.shape[1] >= L.store_n
    return L


def run_chain(segs, layers, buf_cols: int, n_rows: int, ref: Tensor) -> None:
    d = ChainDesc()
    for i, s in enumerate(segs):
        d.segs[i] = s
    for i, l in enumerate(layers):
        d.layers[i] = l
    d.n_segs, d.n_layers, d.buf_cols = len(segs), len(layers), buf_cols
    with torch.cuda.device(ref.device):
        _lib.call("emer_mlp_chain", ctypes.byref(d), n_rows, _stream(ref))


def wgrad(dpre: Tensor, segs, k_total: int, want_bias: bool = True, col0: Optional[Tensor] = None,
          out_w: Optional[Tensor] = None, out_b: Optional[Tensor] = None):
    """dW [N,K], db [N] for dpre [M,N] against the (virtually concatenated) segments.  col0 [M] replaces dpre[:, 0].
    out_w / out_b: ACCUMULATE into these instead of returning fresh tensors (out_w [N, >= K] with unit column stride;
    segment s lands at columns dst_col_s..).  Returns (dW or None, db or None)."""
    M, N = dpre.shape
    dev = dpre.device
    # Weight gradients that go straight into gradient sinks have no consumer inside the backward pass, so they can run
    # on a SIDE STREAM, concurrently with the data-gradient chain that continues on the main stream (neck backward,
    # grid backward: VALU/LDS-bound kernels that leave HBM idle, while these kernels are HBM-bound).  The trainer joins
    # the side stream before it touches the gradients (Trainer.train_step).
    side = SIDE_STREAM if (out_w is not None and (out_b is not None or not want_bias)) else None
    with torch.cuda.device(dev):
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)  # operands were produced on the main stream
            for t in [dpre, col0] + [getattr(sg, "_t", None) for sg in segs]:
                if t is not None:
                    t.record_stream(side)  # the caching allocator must not recycle them while the side stream reads
            ctx = torch.cuda.stream(side)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            n_ws = int(_lib.load().emer_linear_bwd_workspace(M, N, k_total))
            ws = torch.empty((n_ws,), device=dev, dtype=torch.float32)
            need_b = want_bias and out_b is None
            if out_w is None or need_b:
                buf = torch.zeros((N * (k_total if out_w is None else 0) + (N if need_b else 0),), device=dev, dtype=torch.float32)  # one fill
            dw = buf[:N * k_total].view(N, k_total) if out_w is None else None
            db = (buf[-N:] if need_b else None)
            tw = dw if out_w is None else out_w
            tb = out_b if out_b is not None else db
            assert tw.stride(1) == 1 and tw.dtype == torch.float32 and (tb is None or tb.is_contiguous())
            arr = (ChainSeg * MAX_SEGS)()
            for i, sg in enumerate(segs):
                arr[i] = sg
            _lib.call("emer_wgrad_segmented", _p(dpre), dpre.stride(0), _p(col0), arr, len(segs), _p(ws), _p(tw), tw.stride(0),
                      _p(tb) if want_bias else None, M, N, k_total, _stream(dpre))
    return dw, db


FUSED_WGRAD = True   # weight gradients inside the backward kernels where the library has them (emer_neck_bwd_fused); False: separate pass
FUSED_RMLP_WGRAD = os.environ.get("EMER_FUSE_RMLP_WGRAD", "1") != "0"  # ... of the plain 2- / 3-layer heads with <= 16 outputs (emer_rmlp_bwd_fused) [r5]
FUSED_RMLP_WIDE = os.environ.get("EMER_FUSE_RMLP_WIDE", "1") != "0"    # ... incl. the 64-wide feature heads (one wave per SIMD, 192 accumulator registers)
FUSED_RGB_WGRAD = os.environ.get("EMER_FUSE_RGB_WGRAD", "1") != "0"   # ... of the rgb head's layers 0 / 1 too (emer_rgb_head_bwd_fused) [r4]
# [r6] ... optionally with a1 / a2 recomputed in that kernel from geo and the per-ray pre-activations (emer_rgb_head_bwd_recompute): the
# forward stores no hidden activations (512 B / sample less written and kept alive until the backward, 512 B / sample less read); bitwise
# the saved-activation path's results.  EMER_RGB_RECOMPUTE = 0: both stored (default); 1: both recomputed; 2: a1 stored, a2 recomputed.
# Same-session A/B at the metric shape (profiles/r06_ab_rgb_recompute.txt): field_fwd 311 -> 212 / 236 us, backward 368 -> 504 / 484 us,
# step 2.365 -> 2.385 / 2.405 ms -- the recomputation's 168 / 120 matrix instructions per tile cost more in the one-wave-per-SIMD
# backward than the stores cost the four-waves-per-SIMD forward.  It stays as the MEMORY mode (-537 MB of saved activations per million
# samples), off by default.
RGB_RECOMPUTE = int(os.environ.get("EMER_RGB_RECOMPUTE", "0"))


def rgb_recompute(n_rays: int, samples_per_ray: int) -> int:
    """Which hidden activations the backward of an rgb head of this shape will recompute (so that its forward need not store them):
    0 none, 1 a1 and a2, 2 a2 only."""
    ok = FUSED_WGRAD and FUSED_RGB_WGRAD and _lib.load().emer_rgb_head_bwd_fused_workspace(n_rays, samples_per_ray) > 0
    return int(RGB_RECOMPUTE) if ok else 0
SIDE_STREAM = None  # a torch.cuda.Stream: set by a trainer that joins it before reading gradients (see wgrad)


def join_side_stream() -> None:
    """Make the current stream wait for the weight-gradient side streams (no-op when unused)."""
    if SIDE_STREAM is not None:
        torch.cuda.current_stream().wait_stream(SIDE_STREAM)


def _sink(p) -> Optional[Tensor]:
    """The tensor a parameter's gradient may be accumulated into directly: its existing fp32 ``.grad`` with unit
    column stride.  This is what autograd's AccumulateGrad would do after the backward (``grad += dW``), fused into
    the weight-gradient reduction -- it saves a zero-fill and an add launch per parameter.  Opt-in and SCOPED
    (``with grad_sinks():`` around a forward+backward): parameter hooks are bypassed, so only a trainer that owns its
    gradient buffers enables it, and only for its own step.  Parameters that do not require grad never get a sink."""
    if not _USE_GRAD_SINKS:
        return None
    if not getattr(p, "requires_grad", False):
        return None
    g = getattr(p, "grad", None)
    if g is None or g.dtype != torch.float32 or not g.is_cuda or g.stride(-1) != 1:
        return None
    return g


_USE_GRAD_SINKS = False


@contextlib.contextmanager
def grad_sinks(enabled: bool = True):
    """Enable direct accumulation into ``.grad`` for the heads whose FORWARD runs inside this block (the decision is
    recorded per autograd node at forward time, so the matching backward may run inside or outside the block)."""
    global _USE_GRAD_SINKS
    prev, _USE_GRAD_SINKS = _USE_GRAD_SINKS, bool(enabled)
    try:
        yield
    finally:
        _USE_GRAD_SINKS = prev
        join_side_stream()   # (launches the block moved to the auxiliary stream are ordered before whatever follows it)


def _target(sink: Optional[Tensor], shape, dev):
    """(tensor to accumulate into, value to hand back to autograd): the parameter's .grad (-> None) or fresh zeros."""
    if sink is not None:
        return sink, None
    t = torch.zeros(shape, device=dev, dtype=torch.float32)
    return t, t


def _r4(x: int) -> int:
    return (x + 3) // 4 * 4


def _c(t: Tensor) -> Tensor:
    return t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------ base MLP
class _BaseMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor):
        ctx.set_materialize_grads(False)
        enc, W0, B0, W1, B1 = _c(enc_lm), _c(w0), _c(b0), _c(w1), _c(b1)
        L, N, F = enc.shape
        K0, H, NG = L * F, W0.shape[0], W1.shape[0]
        dev = enc.device
        h1 = torch.empty((N, H), device=dev, dtype=torch.float32)
        g = torch.empty((N, NG), device=dev, dtype=torch.float32)
        dens = torch.empty((N,), device=dev, dtype=torch.float32)
        c_h = _r4(K0)
        c_g = c_h + _r4(H)
        run_chain([seg_lm(enc, 0)],
                  [layer(W0, B0, 0, c_h, ACT_RELU, store=h1), layer(W1, B1, c_h, c_g, ACT_NONE, store=g, store_exp0=dens)],
                  c_g + NG, N, enc)
        ctx.save_for_backward(enc, W0, W1, h1, dens)
        return g, dens

    @staticmethod
    def backward(ctx, dg: Optional[Tensor], ddens: Optional[Tensor]):
        enc, W0, W1, h1, dens = ctx.saved_tensors
        L, N, F = enc.shape
        K0, H, NG = L * F, W0.shape[0], W1.shape[0]
        dev = enc.device
        if dg is None and ddens is None:
            return None, None, None, None, None
        dgt = torch.zeros((N, NG), device=dev, dtype=torch.float32) if dg is None else _c(dg)
        # trunc_exp backward (nerf_utils.py:69-72) is merged into geometry feature 0 INSIDE the kernels:
        # column 0 += ddens * min(dens, e^15) while the operand is staged (no 268 MB clone / strided add)
        fa = None if ddens is None else _c(ddens)
        fb = None if ddens is None else dens
        dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
        denc = torch.empty((L, N, F), device=dev, dtype=torch.float32)
        c_h, c_e = _r4(NG), _r4(NG) + _r4(H)
        s0 = seg(dgt, 0, NG)
        s0.fix_a, s0.fix_b = _p(fa), _p(fb)
        run_chain([s0],
                  [layer(W1, None, 0, c_h, transposed=True, mask=h1, store=dpre0),
                   layer(W0, None, c_h, c_e, transposed=True, store=denc, store_lm=True)],
                  c_e + K0, N, enc)
        col0 = None if fa is None else dgt[:, 0] + fa * dens.clamp(max=E15)
        dw1, db1 = wgrad(dgt, [seg(h1, 0, H)], H, col0=col0)
        dw0, db0 = wgrad(dpre0, [seg_lm(enc, 0)], K0)
        return denc, dw0, db0, dw1, db1


def base_mlp(enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor) -> Tuple[Tensor, Tensor]:
    """(feats [N, NG], density [N]) from the level-major grid encoding [L, N, F]."""
    return _BaseMLPFn.apply(*_ng(enc_lm, w0, b0, w1, b1))



# ------------------------------------------------------------------------------- neck (register-resident)
def neck_supported(n_levels: int, n_feat: int, hidden: int, n_out: int) -> bool:
    return bool(_lib.load().emer_neck_supported(n_levels, n_feat, hidden, n_out))


class RgbRider:
    """A colour query that rides along in the neck's forward launch (emer_field_fwd): the neck Function computes the rgb head's
    activations and output together with the geometry features and parks them here; the rgb head Function that is called
    next WITH THOSE geometry features picks them up instead of launching its own forward.  The autograd graph (neck node ->
    rgb node) and both backward passes are exactly those of the separate calls."""

    def __init__(self, hray: Tensor, samples_per_ray: int, params, keep: bool):
        self.hray, self.S, self.params, self.keep = hray, int(samples_per_ray), tuple(params), bool(keep)
        self.geo_ptr = None
        self.rb = self.a1 = self.a2 = self.out = None   # rb: the per-ray pre-activations [R, 128] of layers 0 | 1

    def usable(self, L: int, F: int, N: int, n_out: int) -> bool:
        w0, _, w1, _, w2, _ = self.params
        R, Kh = self.hray.shape
        return bool(n_out == 64 and self.S % 16 == 0 and R * self.S == N and Kh <= 64 and w0.shape == (64, Kh + 64)
                    and w1.shape == (64, 128 + Kh) and w2.shape == (3, 64) and N * L * F < 2 ** 30
                    and _lib.load().emer_field_fwd_supported(L, F))


_RIDERS: dict = {}   # data_ptr of a neck's geometry features -> the rider whose rgb results were computed with them


def clear_riders() -> None:
    _RIDERS.clear()


def _neck_fwd(enc: Tensor, W0, B0, W1, B1, n_out: int, want_h: bool):
    L, N, F = enc.shape
    dev = enc.device
    h1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if want_h else None
    out0 = torch.empty((N, 64), device=dev, dtype=torch.float32) if n_out > 1 else None
    out1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if n_out == 128 else None
    dens = torch.empty((N,), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.call("emer_neck_fwd", _p(enc), L, F, N, _p(W0), _p(B0), _p(W1), _p(B1), n_out, _p(h1), _p(out0), _p(out1), _p(dens),
                  _stream(enc))
    return h1, out0, out1, dens


class _NeckFn(torch.autograd.Function):
    """(features 0..63, features 64..127 or None, density) = neck(enc_lm); density = trunc_exp(feature 0 - 1)."""

    @staticmethod
    def forward(ctx, enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor, rider: Optional[RgbRider] = None):
        ctx.set_materialize_grads(False)
        enc, W0, B0, W1, B1 = _c(enc_lm), _c(w0), _c(b0), _c(w1), _c(b1)
        n_out = W1.shape[0]
        L, N, F = enc.shape
        # the fused backward recomputes the hidden layer from enc: the forward then does not store it (268 MB at 1 M rows)
        ctx.fused_bwd = bool(FUSED_WGRAD and any(ctx.needs_input_grad) and _lib.load().emer_neck_bwd_fused_workspace(L, F, N, n_out) > 0)
        want_h = any(ctx.needs_input_grad) and not ctx.fused_bwd
        if rider is not None and not want_h and rider.usable(L, F, N, n_out):
            h1, out1 = None, None
            out0, dens = _field_fwd(enc, W0, B0, W1, B1, rider)
        else:
            h1, out0, out1, dens = _neck_fwd(enc, W0, B0, W1, B1, n_out, want_h)
        ctx.save_for_backward(enc, W0, W1, h1, dens, B0)
        ctx.n_out = n_out
        ctx.sinks = tuple(_sink(p) for p in (w0, b0, w1, b1))
        if out1 is None:
            return out0, dens
        return out0, out1, dens

    @staticmethod
    def backward(ctx, *grads):
        enc, W0, W1, h1, dens, B0 = ctx.saved_tensors
        n_out = ctx.n_out
        d0, d1, ddens = (grads[0], None, grads[1]) if n_out == 64 else grads
        if d0 is None and d1 is None and ddens is None:
            return None, None, None, None, None, None
        L, N, F = enc.shape
        dev = enc.device
        d0c = None if d0 is None else _c(d0)
        d1c = None if d1 is None else _c(d1)
        fa = None if ddens is None else _c(ddens)
        denc = torch.empty((L, N, F), device=dev, dtype=torch.float32)
        sw0, sb0, sw1, sb1 = ctx.sinks
        tw1, rw1 = _target(sw1, (n_out, 64), dev)
        tb1, rb1 = _target(sb1, (n_out,), dev)
        tw0, rw0 = _target(sw0, (64, L * F), dev)
        tb0, rb0 = _target(sb0, (64,), dev)
        if ctx.fused_bwd:
            # data gradients AND weight gradients in one kernel: neither h1 nor dpre0 reaches memory, enc / d are read once
            ws = torch.empty((int(_lib.load().emer_neck_bwd_fused_workspace(L, F, N, n_out)),), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_neck_bwd_fused", _p(d0c), _p(d1c), _p(fa), _p(dens), _p(enc), L, F, N, _p(W0), _p(B0), _p(W1), n_out,
                          _p(denc), _p(ws), _p(tw0), tw0.stride(0), _p(tb0), _p(tw1), tw1.stride(0), _p(tb1), _stream(enc))
            return denc, rw0, rb0, rw1, rb1, None
        dpre0 = torch.empty((N, 64), device=dev, dtype=torch.float32)
        # column 0 of the output-layer wgrad operand = d0[:, 0] + trunc_exp side gradient, written by the kernel
        col0 = None if fa is None else torch.empty((N,), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("emer_neck_bwd", _p(d0c), _p(d1c), _p(fa), _p(dens), _p(h1), L, F, N, _p(W0), _p(W1), n_out, None, _p(col0),
                      _p(dpre0), _p(denc), _stream(enc))
        # weight gradients, accumulated straight into the parameters' .grad when the trainer allows it (_sink).
        # Output rows whose gradient is structurally zero (an unused semantic half) cost nothing.
        if d0c is None:
            d0c = torch.zeros((N, 64), device=dev, dtype=torch.float32)
        wgrad(d0c, [seg(h1, 0, 64)], 64, col0=col0, out_w=tw1[:64], out_b=tb1[:64])
        if n_out == 128 and d1c is not None:
            wgrad(d1c, [seg(h1, 0, 64)], 64, out_w=tw1[64:], out_b=tb1[64:])
        wgrad(dpre0, [seg_lm(enc, 0)], L * F, out_w=tw0, out_b=tb0)
        return denc, rw0, rb0, rw1, rb1, None


def _field_fwd(enc: Tensor, W0, B0, W1, B1, rider: RgbRider):
    """emer_field_fwd: the neck and the rider's rgb head in one launch; the rgb results are parked on the rider."""
    L, N, F = enc.shape
    dev = enc.device
    hr = _c(rider.hray)
    RW0, RB0, RW1, RB1, RW2, RB2 = (_c(p) for p in rider.params)
    R, Kh = hr.shape
    geo = torch.empty((N, 64), device=dev, dtype=torch.float32)
    dens = torch.empty((N,), device=dev, dtype=torch.float32)
    rb = torch.empty((R, 128), device=dev, dtype=torch.float32)
    rec = rgb_recompute(R, rider.S) if rider.keep else 0   # [r6] recomputing backward: the activations it recomputes are not stored
    a1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if (rider.keep and rec != 1) else None
    a2 = torch.empty((N, 64), device=dev, dtype=torch.float32) if (rider.keep and rec == 0) else None
    out = torch.empty((N, 3), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.call("emer_ray_pre_fwd", _p(hr), hr.stride(0), R, Kh, 64, _p(RW0), RW0.stride(0), _p(RB0), _p(RW1[:, 64:]), RW1.stride(0),
                  _p(RB1), _p(rb), 128, _stream(enc))
        _lib.call("emer_field_fwd", _p(enc), L, F, R, rider.S, _p(W0), _p(B0), _p(W1), _p(B1), _p(rb), _p(rb[:, 64:]), 128, Kh,
                  _p(RW0), _p(RW1), _p(RW2), _p(RB2), _p(geo), _p(dens), _p(a1), _p(a2), _p(out), _stream(enc))
    rider.geo_ptr, rider.a1, rider.a2, rider.out, rider.rb = geo.data_ptr(), a1, a2, out, rb
    _RIDERS[geo.data_ptr()] = rider
    return geo, dens


def neck(enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor, rider: Optional[RgbRider] = None):
    """Register-resident neck: returns (feats[:, :64], feats[:, 64:128] or None, density [N]).  Requires
    ``neck_supported(L, F, hidden, n_out)`` with n_out in (64, 128).  ``rider``: a colour query on the resulting geometry
    features to evaluate in the same launch (see RgbRider)."""
    r = _NeckFn.apply(*_ng(enc_lm, w0, b0, w1, b1), rider)
    return (r[0], None, r[1]) if len(r) == 2 else r


# -------------------------------------------------------------------------------------- density MLP
class _DensityMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor):
        ctx.set_materialize_grads(False)
        enc, W0, B0, W1, B1 = _c(enc_lm), _c(w0), _c(b0), _c(w1), _c(b1)
        L, N, F = enc.shape
        K0, H = L * F, W0.shape[0]
        dev = enc.device
        need_grad = any(ctx.needs_input_grad)
        ctx.fast = neck_supported(L, F, H, 1)
        ctx.sinks = tuple(_sink(p) for p in (w0, b0, w1, b1))
        # narrow inputs (the proposal networks): one backward kernel with the weight gradients, hidden layer recomputed
        ctx.fused_bwd = bool(ctx.fast and FUSED_WGRAD and need_grad and _lib.load().emer_density_bwd_fused_workspace(L, F, N) > 0)
        if ctx.fast:
            h, _, _, dens = _neck_fwd(enc, W0, B0, W1, B1, 1, need_grad and not ctx.fused_bwd)
            ctx.save_for_backward(enc, W0, W1, h, dens, B0)
            return dens
        h = torch.empty((N, H), device=dev, dtype=torch.float32) if need_grad else None
        dens = torch.empty((N, 1), device=dev, dtype=torch.float32)
        c_h = _r4(K0)
        c_o = c_h + _r4(H)
        run_chain([seg_lm(enc, 0)],
                  [layer(W0, B0, 0, c_h, ACT_RELU, store=h), layer(W1, B1, c_h, c_o, ACT_TRUNC_EXP, store=dens)],
                  c_o + 4, N, enc)
        ctx.save_for_backward(enc, W0, W1, h, dens, B0)
        return dens.view(N)

    @staticmethod
    def backward(ctx, ddens: Optional[Tensor]):
        enc, W0, W1, h, dens, B0 = ctx.saved_tensors
        if ddens is None:
            return None, None, None, None, None
        L, N, F = enc.shape
        K0, H = L * F, W0.shape[0]
        dev = enc.device
        if ctx.fused_bwd:
            denc = torch.empty((L, N, F), device=dev, dtype=torch.float32)
            sw0, sb0, sw1, sb1 = ctx.sinks
            tw0, rw0 = _target(sw0, (H, K0), dev)
            tb0, rb0 = _target(sb0, (H,), dev)
            tw1, rw1 = _target(sw1, (1, H), dev)
            tb1, rb1 = _target(sb1, (1,), dev)
            assert tw1.is_contiguous()
            ws = torch.empty((int(_lib.load().emer_density_bwd_fused_workspace(L, F, N)),), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_density_bwd_fused", _p(_c(ddens).reshape(-1)), _p(dens), _p(enc), L, F, N, _p(W0), _p(B0), _p(W1), _p(denc), _p(ws),
                          _p(tw0), tw0.stride(0), _p(tb0), _p(tw1), _p(tb1), _stream(enc))
            return denc, rw0, rb0, rw1, rb1
        if ctx.fast:
            dpre1 = torch.empty((N, 1), device=dev, dtype=torch.float32)
            dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
            denc = torch.empty((L, N, F), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_neck_bwd", None, None, _p(_c(ddens)), _p(dens), _p(h), L, F, N, _p(W0), _p(W1), 1, _p(dpre1), None,
                          _p(dpre0), _p(denc), _stream(enc))
            sw0, sb0, sw1, sb1 = ctx.sinks
            tw1, rw1 = _target(sw1, (1, H), dev)
            tb1, rb1 = _target(sb1, (1,), dev)
            tw0, rw0 = _target(sw0, (H, K0), dev)
            tb0, rb0 = _target(sb0, (H,), dev)
            wgrad(dpre1, [seg(h, 0, H)], H, out_w=tw1, out_b=tb1)
            wgrad(dpre0, [seg_lm(enc, 0)], K0, out_w=tw0, out_b=tb0)
            return denc, rw0, rb0, rw1, rb1
        dpre1 = (_c(ddens).view(N, 1) * dens.view(N, 1).clamp(max=E15)).contiguous()
        dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
        denc = torch.empty((L, N, F), device=dev, dtype=torch.float32)
        c_h, c_e = 4, 4 + _r4(H)
        run_chain([seg(dpre1, 0, 1)],
                  [layer(W1, None, 0, c_h, transposed=True, mask=h, store=dpre0),
                   layer(W0, None, c_h, c_e, transposed=True, store=denc, store_lm=True)],
                  c_e + K0, N, enc)
        dw1, db1 = wgrad(dpre1, [seg(h, 0, H)], H)
        dw0, db0 = wgrad(dpre0, [seg_lm(enc, 0)], K0)
        return denc, dw0, db0, dw1, db1


def density_mlp(enc_lm: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor) -> Tensor:
    """trunc_exp(Linear(ReLU(Linear(enc))) - 1) -> [N]."""
    return _DensityMLPFn.apply(*_ng(enc_lm, w0, b0, w1, b1))


# ------------------------------------------------------------------------------------------ rgb head
class _RgbHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hray: Tensor, geo: Tensor, S: int, w0, b0, w1, b1, w2, b2, pre: Optional[RgbRider] = None):
        ctx.set_materialize_grads(False)
        hr, W0, B0, W1, B1, W2, B2 = _c(hray), _c(w0), _c(b0), _c(w1), _c(b1), _c(w2), _c(b2)
        g = geo.detach()
        assert g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1
        N, NG = g.shape
        R, Kh = hr.shape
        assert R * S == N
        H, K0, C = W0.shape[0], Kh + NG, W2.shape[0]
        assert W0.shape[1] == K0 and W1.shape[1] == H + K0 and W2.shape[1] == H
        dev = g.device
        keep = any(ctx.needs_input_grad)  # inference (inputs detached by rgb_head): the fast kernel stores no activations
        fast = (H == 64 and NG == 64 and C == 3 and Kh <= 64 and S % 16 == 0 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0)
        # [r6] the fused backward recomputes a1 / a2 from geo and the per-ray pre-activations: the forward then stores neither
        ctx.recompute = rgb_recompute(R, S) if (keep and fast) else 0
        a1 = torch.empty((N, H), device=dev, dtype=torch.float32) if ((keep and ctx.recompute != 1) or not fast) else None
        a2 = torch.empty((N, H), device=dev, dtype=torch.float32) if ((keep and ctx.recompute == 0) or not fast) else None
        out = torch.empty((N, C), device=dev, dtype=torch.float32)
        ctx.S = S
        ctx.sinks = tuple(_sink(p) for p in (w0, b0, w1, b1, w2, b2))
        ctx.fast = fast
        if pre is not None and ctx.fast and (pre.keep or not keep):
            # already evaluated inside the neck's launch on exactly these tensors (rgb_head checked): nothing to launch
            assert not keep or ((pre.a1 is None) == (ctx.recompute == 1) and (pre.a2 is None) == (ctx.recompute != 0)), \
                "rider and rgb head disagree about stored activations"
            ctx.save_for_backward(hr, g, W0, W1, W2, pre.a1, pre.a2, pre.out, pre.rb if ctx.recompute else None)
            return pre.out
        if ctx.fast:
            # per-ray part of layers 0 and 1 as per-ray pre-activations (ONE 8192-row launch instead of two 1M-row GEMMs),
            # reading the two column blocks of W0 / W1 in place
            rb = torch.empty((R, 2 * H), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_ray_pre_fwd", _p(hr), hr.stride(0), R, Kh, H, _p(W0), W0.stride(0), _p(B0), _p(W1[:, H:]), W1.stride(0),
                          _p(B1), _p(rb), 2 * H, _stream(g))
            rb0, rb1 = rb[:, :H], rb[:, H:]
            with torch.cuda.device(dev):
                _lib.call("emer_rgb_head_fwd", _p(g), g.stride(0), _p(rb0), _p(rb1), rb.stride(0), R, S, Kh, _p(W0), _p(W1), _p(W2), _p(B2),
                          _p(a1), _p(a2), _p(out), _stream(g))
            ctx.save_for_backward(hr, g, W0, W1, W2, a1, a2, out, rb if ctx.recompute else None)
            return out
        c_x = _r4(H)               # [A1 | hray | geo] laid out exactly like torch.cat([x, input]) of mlp.py:42
        # A2 overwrites A1 in place: a <= 64-wide layer is one column group, so every input column has been
        # consumed by the MFMAs before the epilogue writes.  The smaller row buffer buys more waves per CU.
        c_o = c_x + _r4(K0)
        run_chain([seg(hr, c_x, Kh, row_div=S), seg(g, c_x + Kh, NG, ld=g.stride(0))],
                  [layer(W0, B0, c_x, 0, ACT_RELU, store=a1),
                   layer(W1, B1, 0, 0, ACT_RELU, store=a2),
                   layer(W2, B2, 0, c_o, ACT_SIGMOID, store=out)],
                  c_o + 4, N, g)
        ctx.save_for_backward(hr, g, W0, W1, W2, a1, a2, out, None)
        return out

    @staticmethod
    def backward(ctx, dout: Optional[Tensor]):
        hr, g, W0, W1, W2, a1, a2, out, rb = ctx.saved_tensors
        if dout is None:
            return (None,) * 10
        S = ctx.S
        N, NG = g.shape
        R, Kh = hr.shape
        H, K0, C = W0.shape[0], Kh + NG, W2.shape[0]
        assert H == _r4(H)
        dev = g.device
        if ctx.fast:
            dgeo = torch.empty((N, NG), device=dev, dtype=torch.float32)
            s1 = torch.empty((R, H), device=dev, dtype=torch.float32)
            s0 = torch.empty((R, H), device=dev, dtype=torch.float32)
            sw0, sb0, sw1, sb1, sw2, sb2 = ctx.sinks
            tw2, rw2 = _target(sw2, (C, H), dev)
            tb2, rb2 = _target(sb2, (C,), dev)
            tw1, rw1 = _target(sw1, (H, H + K0), dev)
            tb1, rb1 = _target(sb1, (H,), dev)
            tw0, rw0 = _target(sw0, (H, K0), dev)
            tb0, rb0 = _target(sb0, (H,), dev)
            n_ws = int(_lib.load().emer_rgb_head_bwd_fused_workspace(R, S)) if (FUSED_WGRAD and FUSED_RGB_WGRAD) else 0
            if n_ws > 0:
                # [r4] the weight gradients of the per-sample column blocks of layers 0 / 1 (and layer 2) inside the backward kernel:
                # dpre1 / dpre0 never reach memory and the two streamed weight-gradient launches are gone
                ws = torch.empty((n_ws,), device=dev, dtype=torch.float32)
                with torch.cuda.device(dev):
                    if ctx.recompute:   # [r6] hidden activations recomputed from geo + the per-ray pre-activations
                        _lib.call("emer_rgb_head_bwd_recompute", _p(_c(dout)), _p(out), _p(a1), _p(g), g.stride(0), _p(rb), _p(rb[:, H:]), rb.stride(0), R, S, Kh,
                                  _p(W0), _p(W1), _p(W2), _p(dgeo), _p(s1), _p(s0), _p(ws), _p(tw0), tw0.stride(0), _p(tw1), tw1.stride(0), _p(tw2),
                                  tw2.stride(0), _p(tb2), _stream(g))
                    else:
                        _lib.call("emer_rgb_head_bwd_fused", _p(_c(dout)), _p(out), _p(a1), _p(a2), _p(g), g.stride(0), R, S, Kh, _p(W0), _p(W1), _p(W2),
                                  _p(dgeo), _p(s1), _p(s0), _p(ws), _p(tw0), tw0.stride(0), _p(tw1), tw1.stride(0), _p(tw2), tw2.stride(0), _p(tb2), _stream(g))
                ray_wgrad([(s1, [(hr, Kh, H)], tw1, tb1), (s0, [(hr, Kh, 0)], tw0, tb0)], g)
                dhray = torch.empty((R, Kh), device=dev, dtype=torch.float32)
                with torch.cuda.device(dev):
                    _lib.call("emer_ray_pre_bwd", _p(s0), _p(s1), H, R, Kh, H, _p(W0), W0.stride(0), _p(W1[:, H:]), W1.stride(0), _p(dhray), Kh,
                              _stream(g))
                return dhray, dgeo, None, rw0, rb0, rw1, rb1, rw2, rb2, None
            dpre2 = torch.empty((N, C), device=dev, dtype=torch.float32)
            dpre1 = torch.empty((N, H), device=dev, dtype=torch.float32)
            dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                # the output layer's weight gradient (3 x 64 + 3 numbers) rides along in the backward kernel, where a2 and
                # dpre2 are in registers: a separate pass would re-read 280 MB for it
                ws = torch.empty((int(_lib.load().emer_rgb_head_bwd_workspace(R)),), device=dev, dtype=torch.float32) if FUSED_WGRAD else None
                _lib.call("emer_rgb_head_bwd", _p(_c(dout)), _p(out), _p(a1), _p(a2), R, S, Kh, _p(W0), _p(W1), _p(W2), _p(dpre2),
                          _p(dpre1), _p(dpre0), _p(dgeo), _p(s1), _p(s0), _p(ws), _p(tw2) if FUSED_WGRAD else None, tw2.stride(0),
                          _p(tb2) if FUSED_WGRAD else None, _stream(g))
            if not FUSED_WGRAD:
                wgrad(dpre2, [seg(a2, 0, H)], H, out_w=tw2, out_b=tb2)
            # per-sample column blocks [A1 | . | geo] of dW1 and [. | geo] of dW0 ...
            wgrad(dpre1, [seg(a1, 0, H, dst_col=0), seg(g, H, NG, ld=g.stride(0), dst_col=H + Kh)], H + NG, want_bias=False, out_w=tw1)
            wgrad(dpre0, [seg(g, 0, NG, ld=g.stride(0), dst_col=Kh)], NG, want_bias=False, out_w=tw0)
            # ... and everything that multiplies the per-ray operand from the per-ray sums of dpre1 / dpre0
            # (8192-row GEMMs; colsum(s) is the bias gradient)
            ray_wgrad([(s1, [(hr, Kh, H)], tw1, tb1), (s0, [(hr, Kh, 0)], tw0, tb0)], g)
            dhray = torch.empty((R, Kh), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_ray_pre_bwd", _p(s0), _p(s1), H, R, Kh, H, _p(W0), W0.stride(0), _p(W1[:, H:]), W1.stride(0), _p(dhray), Kh,
                          _stream(g))
            return dhray, dgeo, None, rw0, rb0, rw1, rb1, rw2, rb2, None
        dpre2 = (_c(dout) * out * (1.0 - out)).contiguous()            # sigmoid'
        dpre1 = torch.empty((N, H), device=dev, dtype=torch.float32)
        dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
        dh = torch.empty((N, Kh), device=dev, dtype=torch.float32)
        dgeo = torch.empty((N, NG), device=dev, dtype=torch.float32)
        c1 = _r4(C)                 # dA2 -> dPre1
        cx = c1 + H                 # d[A1 | hray | geo]; dA1 -> dPre0 in place
        run_chain([seg(dpre2, 0, C)],
                  [layer(W2, None, 0, c1, transposed=True, mask=a2, store=dpre1),
                   layer(W1, None, c1, cx, transposed=True, n_slice=(0, H), mask=a1, store=dpre0),
                   layer(W1, None, c1, cx + H, transposed=True, n_slice=(H, H + K0)),
                   layer(W0, None, cx, cx + H, transposed=True, n_slice=(0, Kh), accumulate=True, store=dh),
                   layer(W0, None, cx, cx + H + Kh, transposed=True, n_slice=(Kh, K0), accumulate=True, store=dgeo)],
                  cx + H + K0, N, g)
        dw2, db2 = wgrad(dpre2, [seg(a2, 0, H)], H)
        dw1, db1 = wgrad(dpre1, [seg(a1, 0, H), seg(hr, H, Kh, row_div=S), seg(g, H + Kh, NG, ld=g.stride(0))], H + K0)
        dw0, db0 = wgrad(dpre0, [seg(hr, 0, Kh, row_div=S), seg(g, Kh, NG, ld=g.stride(0))], K0)
        dhray = dh.view(R, S, Kh).sum(dim=1)
        return dhray, dgeo, None, dw0, db0, dw1, db1, dw2, db2, None


def rgb_head(hray: Tensor, geo: Tensor, samples_per_ray: int, w0, b0, w1, b1, w2, b2) -> Tensor:
    """sigmoid(MLP3-skip1([hray[ray] | geo])) -> [N, 3]; hray [R, Kh] per ray, geo [N, NG] per sample."""
    pre = _RIDERS.pop(geo.data_ptr(), None)
    if pre is not None:  # the same query, tensor for tensor, that rode along in the neck's launch?
        same = (pre.S == samples_per_ray and pre.hray.data_ptr() == hray.data_ptr() and pre.hray.shape == hray.shape
                and geo.dim() == 2 and geo.shape[1] == 64 and geo.stride(0) == 64
                and all(a is b for a, b in zip(pre.params, (w0, b0, w1, b1, w2, b2))))
        pre = pre if same else None
    return _RgbHeadFn.apply(*_ng(hray, geo, samples_per_ray, w0, b0, w1, b1, w2, b2), pre)


# ------------------------------------------------------------------------------------------ per-ray inputs
class _RayInputsFn(torch.autograd.Function):
    """Input rows of the rgb head ([PE((d+1)/2) | emb[idx]]) and of the sky head ([PE(d) | emb[idx]]) in ONE launch
    (radiance_field.py:622-643,660-674; replaces two encoder launches, the gather and two cats), and the embedding
    table's gradient as one deterministic segment sum over both consumers' input gradients (replaces autograd's add,
    a zero fill and an atomic index_add)."""

    @staticmethod
    def forward(ctx, weight: Tensor, idx: Tensor, dirs: Tensor, max_deg: int):
        ctx.set_materialize_grads(False)
        w = _c(weight)
        assert idx.dtype == torch.int64 and idx.dim() == 1 and dirs.dim() == 2 and dirs.dtype == torch.float32 and dirs.stride(1) == 1
        R, E = dirs.shape[0], w.shape[1]
        P = 3 if max_deg == 0 else 3 * (1 + 2 * (max_deg + 1))
        dev = dirs.device
        with torch.cuda.device(dev):
            out_rgb = torch.empty((R, P + E), device=dev, dtype=torch.float32)
            out_sky = torch.empty((R, P + E), device=dev, dtype=torch.float32)
            _lib.call("emer_ray_inputs_fwd", _p(dirs), dirs.stride(0), _p(idx), idx.stride(0), _p(w), w.shape[0], E, max_deg, R,
                      _p(out_rgb), P + E, _p(out_sky), P + E, _stream(dirs))
        ctx.save_for_backward(idx)
        ctx.sink, ctx.shape, ctx.P = _sink(weight), tuple(w.shape), P
        return out_rgb, out_sky

    @staticmethod
    def backward(ctx, g_rgb: Optional[Tensor], g_sky: Optional[Tensor]):
        if g_rgb is None and g_sky is None:
            return None, None, None, None
        (idx,) = ctx.saved_tensors
        dev = idx.device
        P = ctx.P

        def cols(g):
            if g is None:
                return None, 0
            g = g if (g.dtype == torch.float32 and g.stride(1) == 1) else g.to(torch.float32).contiguous()
            return g[:, P:], g.stride(0)

        (ga, lda), (gb, ldb) = cols(g_rgb), cols(g_sky)
        tw, rw = _target(ctx.sink, ctx.shape, dev)
        with torch.cuda.device(dev):
            _lib.call("emer_embed_grad", _p(ga), lda, _p(gb), ldb, _p(idx), idx.stride(0), idx.shape[0], ctx.shape[0], ctx.shape[1],
                      _p(tw), _stream(idx))
        return rw, None, None, None



def ray_inputs(emb_weight: Tensor, idx: Tensor, dirs: Tensor, max_deg: int):
    """(rgb-head rows, sky-head rows), each [R, PE + emb_dim], for per-ray directions [R, 3] and embedding indices [R]."""
    return _RayInputsFn.apply(*_ng(emb_weight, idx, dirs, max_deg))


# ------------------------------------------------------------------------- 3-layer skip MLP on row-major input
class _SkipMLP3Fn(torch.autograd.Function):
    """act(MLP(x)) for mlp.MLP(num_layers=3, skip_connections=[1]) (mlp.py:20-46) on a plain row-major input -- the
    per-ray sky head (radiance_field.py:156-187,660-686).  One chain launch forward, one for the data gradients,
    instead of three Linear launches each way with [rows, 64] round trips."""

    @staticmethod
    def forward(ctx, x: Tensor, w0, b0, w1, b1, w2, b2, final_act: int):
        ctx.set_materialize_grads(False)
        X, W0, B0, W1, B1, W2, B2 = _c(x), _c(w0), _c(b0), _c(w1), _c(b1), _c(w2), _c(b2)
        N, K0 = X.shape
        H, C = W0.shape[0], W2.shape[0]
        assert W0.shape[1] == K0 and W1.shape[1] == H + K0 and W2.shape[1] == H and H % 4 == 0
        dev = X.device
        a1 = torch.empty((N, H), device=dev, dtype=torch.float32)
        a2 = torch.empty((N, H), device=dev, dtype=torch.float32)
        out = torch.empty((N, C), device=dev, dtype=torch.float32)
        ctx.fast = H == 64 and K0 <= 64 and C <= 16
        if ctx.fast:
            # a few thousand rows: two small launches (the input's share of layers 0 and 1, then the rest of the MLP)
            # instead of the general chain kernel, whose set-up dominates at this size
            rb = torch.empty((N, 2 * H), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_ray_pre_fwd", _p(X), X.stride(0), N, K0, H, _p(W0), W0.stride(0), _p(B0), _p(W1[:, H:]), W1.stride(0), _p(B1),
                          _p(rb), 2 * H, _stream(X))
                _lib.call("emer_ray_head_fwd", _p(rb), 2 * H, N, _p(W1), W1.stride(0), _p(W2), _p(B2), C, final_act, _p(a1), _p(a2), _p(out),
                          _stream(X))
        else:
            c_x = H                     # [A1 | x] laid out exactly like torch.cat([x_hidden, input]) of mlp.py:42
            c_o = c_x + _r4(K0)
            run_chain([seg(X, c_x, K0)],
                      [layer(W0, B0, c_x, 0, ACT_RELU, store=a1),
                       layer(W1, B1, 0, 0, ACT_RELU, store=a2),     # A2 overwrites A1 in place (one column group)
                       layer(W2, B2, 0, c_o, final_act, store=out)],
                      c_o + _r4(C), N, X)
        ctx.save_for_backward(X, W0, W1, W2, a1, a2, out)
        ctx.final_act = final_act
        ctx.sinks = tuple(_sink(p) for p in (w0, b0, w1, b1, w2, b2))
        return out

    @staticmethod
    def backward(ctx, dout: Optional[Tensor]):
        X, W0, W1, W2, a1, a2, out = ctx.saved_tensors
        if dout is None:
            return (None,) * 8
        N, K0 = X.shape
        H, C = W0.shape[0], W2.shape[0]
        dev = X.device
        d = _c(dout)
        dpre1 = torch.empty((N, H), device=dev, dtype=torch.float32)
        dpre0 = torch.empty((N, H), device=dev, dtype=torch.float32)
        dx = torch.empty((N, K0), device=dev, dtype=torch.float32)
        if ctx.fast:
            dpre2 = torch.empty((N, C), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("emer_ray_head_bwd", _p(d), _p(out), _p(a1), _p(a2), N, _p(W1), W1.stride(0), _p(W2), C, ctx.final_act, _p(dpre2),
                          _p(dpre1), _p(dpre0), _stream(X))
                _lib.call("emer_ray_pre_bwd", _p(dpre0), _p(dpre1), H, N, K0, H, _p(W0), W0.stride(0), _p(W1[:, H:]), W1.stride(0), _p(dx), K0,
                          _stream(X))
        else:
            dpre2 = torch.ops.aten.sigmoid_backward(d, out) if ctx.final_act == ACT_SIGMOID else d  # sigmoid' (one launch) / identity
            c1 = _r4(C)                 # dA2 -> dPre1
            cx = c1 + H                 # d[A1 | x]; dA1 -> dPre0 in place
            run_chain([seg(dpre2, 0, C)],
                      [layer(W2, None, 0, c1, transposed=True, mask=a2, store=dpre1),
                       layer(W1, None, c1, cx, transposed=True, n_slice=(0, H), mask=a1, store=dpre0),
                       layer(W1, None, c1, cx + H, transposed=True, n_slice=(H, H + K0)),
                       layer(W0, None, cx, cx + H, transposed=True, accumulate=True, store=dx)],
                      cx + H + _r4(K0), N, X)
        sw0, sb0, sw1, sb1, sw2, sb2 = ctx.sinks
        tw2, rw2 = _target(sw2, (C, H), dev)
        tb2, rb2 = _target(sb2, (C,), dev)
        tw1, rw1 = _target(sw1, (H, H + K0), dev)
        tb1, rb1 = _target(sb1, (H,), dev)
        tw0, rw0 = _target(sw0, (H, K0), dev)
        tb0, rb0 = _target(sb0, (H,), dev)
        if ctx.fast and N <= 65536:  # per-ray head: all three layers in one launch
            ray_wgrad([(dpre2, [(a2, H, 0)], tw2, tb2), (dpre1, [(a1, H, 0), (X, K0, H)], tw1, tb1), (dpre0, [(X, K0, 0)], tw0, tb0)], X)
        else:
            wgrad(dpre2, [seg(a2, 0, H)], H, out_w=tw2, out_b=tb2)
            wgrad(dpre1, [seg(a1, 0, H), seg(X, H, K0)], H + K0, out_w=tw1, out_b=tb1)
            wgrad(dpre0, [seg(X, 0, K0)], K0, out_w=tw0, out_b=tb0)
        return dx, rw0, rb0, rw1, rb1, rw2, rb2, None


def skip_mlp3(x: Tensor, w0, b0, w1, b1, w2, b2, final_act: int = ACT_SIGMOID) -> Tensor:
    """final_act(MLP3-skip1(x)) for x [rows, K0]; final_act ACT_SIGMOID or ACT_NONE."""
    assert final_act in (ACT_SIGMOID, ACT_NONE)
    return _SkipMLP3Fn.apply(*_ng(x, w0, b0, w1, b1, w2, b2, final_act))


# ----------------------------------------------------------------- plain nn.Sequential heads (shadow / flow / dino)
def seq_mlp_supported(weights) -> bool:
    """2..4 Linear layers whose weights fit the chain kernel's LDS budget and column buffer."""
    if not (2 <= len(weights) <= 4):
        return False
    lds = sum((-(-w.shape[0] // 16) * 16) * ((-(-w.shape[1] // 8) * 8) + 2) * 4 for w in weights)
    cols = _r4(weights[0].shape[1]) + sum(_r4(w.shape[0]) for w in weights)
    return lds <= 96 * 1024 and cols <= 500 and all(w.shape[0] <= 256 for w in weights)


class _SeqMLPFn(torch.autograd.Function):
    """final_act(Linear_n(ReLU(... ReLU(Linear_1(x))))) -- the shadow head (radiance_field.py:148-153), flow MLP
    (:101-111) and dino heads (:192-198) as ONE chain launch forward and one for the data gradients."""

    @staticmethod
    def forward(ctx, x: Tensor, final_act: int, *wb):
        ctx.set_materialize_grads(False)
        X = _c(x)
        Ws, Bs = [_c(w) for w in wb[0::2]], [None if b is None else _c(b) for b in wb[1::2]]
        N, K0 = X.shape
        dev = X.device
        n = len(Ws)
        cols = [0, _r4(K0)]
        for w in Ws:
            cols.append(cols[-1] + _r4(w.shape[0]))
        acts = [torch.empty((N, w.shape[0]), device=dev, dtype=torch.float32) for w in Ws]
        layers = [layer(Ws[i], Bs[i], cols[i], cols[i + 1], ACT_RELU if i + 1 < n else final_act, store=acts[i]) for i in range(n)]
        run_chain([seg(X, 0, K0)], layers, cols[-1], N, X)
        ctx.save_for_backward(X, *Ws, *acts)
        ctx.n, ctx.final_act = n, final_act
        ctx.sinks = tuple(_sink(p) for p in wb)
        ctx.need_dx = ctx.needs_input_grad[0]
        return acts[-1]

    @staticmethod
    def backward(ctx, dout: Optional[Tensor]):
        n = ctx.n
        if dout is None:
            return (None,) * (2 + 2 * n)
        saved = ctx.saved_tensors
        X, Ws, acts = saved[0], saved[1:1 + n], saved[1 + n:]
        N, K0 = X.shape
        dev = X.device
        d = _c(dout)
        out = acts[-1]
        dlast = (d * out * (1.0 - out)).contiguous() if ctx.final_act == ACT_SIGMOID else d
        dpre = [None] * n
        dpre[n - 1] = dlast
        cols = [0, _r4(Ws[n - 1].shape[0])]
        layers = []
        for i in range(n - 1, 0, -1):      # dA_{i-1} = dPre_i W_i, masked by relu'(act_{i-1}) -> dPre_{i-1}
            dpre[i - 1] = torch.empty((N, Ws[i].shape[1]), device=dev, dtype=torch.float32)
            cols.append(cols[-1] + _r4(Ws[i].shape[1]))
            layers.append(layer(Ws[i], None, cols[-3], cols[-2], transposed=True, mask=acts[i - 1], store=dpre[i - 1]))
        dx = None
        if ctx.need_dx:
            dx = torch.empty((N, K0), device=dev, dtype=torch.float32)
            cols.append(cols[-1] + _r4(K0))
            layers.append(layer(Ws[0], None, cols[-3], cols[-2], transposed=True, store=dx))
        if layers:
            run_chain([seg(dlast, 0, Ws[n - 1].shape[0])], layers, cols[-1], N, X)
        grads = []
        for i in range(n):
            sw, sb = ctx.sinks[2 * i], ctx.sinks[2 * i + 1]
            tw, rw = _target(sw, tuple(Ws[i].shape), dev)
            has_b = ctx.needs_input_grad[2 + 2 * i + 1]
            tb, rb = _target(sb, (Ws[i].shape[0],), dev) if has_b else (None, None)
            operand = X if i == 0 else acts[i - 1]
            wgrad(dpre[i], [seg(operand, 0, Ws[i].shape[1])], Ws[i].shape[1], want_bias=has_b, out_w=tw, out_b=tb)
            grads += [rw, rb]
        return (dx, None, *grads)


# ------------------------------------------------------------- plain 2- / 3-layer heads (register-resident)
def rmlp_supported(weights, k0: int, n_feat: int) -> bool:
    """Linear(k0, 64)-ReLU-[Linear(64, 64)-ReLU-]Linear(64, n_out <= 64) on row-major (n_feat 0) or level-major input."""
    n = len(weights)
    if n not in (2, 3):
        return False
    if any(w.shape[0] != 64 for w in weights[:-1]) or any(w.shape[1] != 64 for w in weights[1:]) or weights[0].shape[1] != k0:
        return False
    return bool(_lib.load().emer_rmlp_supported(n, k0, n_feat, 64, weights[-1].shape[0]))


class _RMlpFn(torch.autograd.Function):
    """final_act(Linear(ReLU(... ReLU(Linear(x))))) with hidden width 64 as ONE register-resident kernel each way
    (csrc/mlp_fused.hip: rmlp_fwd / rmlp_bwd): the flow MLP on the level-major xyzt encoding (radiance_field.py:101-111,
    359-389), the shadow head (:148-153) and the feature heads (:192-198)."""

    @staticmethod
    def forward(ctx, x: Tensor, level_major: bool, final_act: int, *wb):
        ctx.set_materialize_grads(False)
        X = _c(x)
        Ws, Bs = [_c(w) for w in wb[0::2]], [None if b is None else _c(b) for b in wb[1::2]]
        n = len(Ws)
        if level_major:
            L, N, F = X.shape
            K0, ldx = L * F, 0
        else:
            N, K0 = X.shape
            L, F, ldx = 0, 0, K0
        dev = X.device
        n_out = Ws[-1].shape[0]
        need_bwd = any(ctx.needs_input_grad)
        # [r5] stacks with at most 16 outputs (flow MLP, shadow head) and the three-layer feature heads: the backward is ONE kernel that
        # recomputes the hidden layers from x and keeps the weight gradients in registers (emer_rmlp_bwd_fused) -- the forward then stores
        # no activations
        ctx.fused_bwd = bool(FUSED_WGRAD and FUSED_RMLP_WGRAD and (n_out <= 16 or FUSED_RMLP_WIDE) and need_bwd and all(b is not None for b in Bs[:n - 1])
                             and _lib.load().emer_rmlp_bwd_fused_workspace(n, K0, F, N, n_out) > 0)
        keep = need_bwd and not ctx.fused_bwd
        h1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if keep else None
        h2 = torch.empty((N, 64), device=dev, dtype=torch.float32) if (keep and n == 3) else None
        out = torch.empty((N, n_out), device=dev, dtype=torch.float32)
        w2, b2 = (Ws[2], Bs[2]) if n == 3 else (None, None)
        with torch.cuda.device(dev):
            _lib.call("emer_rmlp_fwd", _p(X), ldx, L, F, K0, N, n, _p(Ws[0]), _p(Bs[0]), _p(Ws[1]), _p(Bs[1]), _p(w2), _p(b2),
                      n_out, final_act, _p(h1), _p(h2), _p(out), n_out, _stream(X))
        if ctx.fused_bwd:
            ctx.save_for_backward(X, out, *Ws, *Bs[:n - 1])
        else:
            ctx.save_for_backward(X, out, *Ws, *([h1] if h1 is not None else []), *([h2] if h2 is not None else []))
        ctx.n, ctx.final_act, ctx.level_major = n, final_act, level_major
        ctx.sinks = tuple(_sink(p) for p in wb)
        ctx.need_dx = ctx.needs_input_grad[0]
        return out

    @staticmethod
    def backward(ctx, dout: Optional[Tensor]):
        n = ctx.n
        if dout is None:
            return (None,) * (3 + 2 * n)
        saved = ctx.saved_tensors
        X, out, Ws = saved[0], saved[1], saved[2:2 + n]
        dev = X.device
        if ctx.level_major:
            L, N, F = X.shape
            K0 = L * F
        else:
            N, K0 = X.shape
            L, F = 0, 0
        n_out = Ws[-1].shape[0]
        d = _c(dout)
        if ctx.fused_bwd:
            Bh = saved[2 + n:2 + n + n - 1]   # biases of the hidden layers (the recomputation needs them)
            dx = torch.empty_like(X) if ctx.need_dx else None
            tgt = []
            for i in range(n):
                tw, rw = _target(ctx.sinks[2 * i], tuple(Ws[i].shape), dev)
                has_b = ctx.needs_input_grad[3 + 2 * i + 1]
                tb, rb = _target(ctx.sinks[2 * i + 1], (Ws[i].shape[0],), dev) if has_b else (None, None)
                tgt.append((tw, rw, tb, rb))
            ws = torch.empty((int(_lib.load().emer_rmlp_bwd_fused_workspace(n, K0, F, N, n_out)),), device=dev, dtype=torch.float32)
            w2 = Ws[2] if n == 3 else None
            b1 = Bh[1] if n == 3 else None
            t2 = tgt[2] if n == 3 else (None, None, None, None)
            with torch.cuda.device(dev):
                _lib.call("emer_rmlp_bwd_fused", _p(d), d.stride(0), _p(out), out.stride(0), _p(X), (0 if ctx.level_major else X.stride(0)), L, F, K0, N, n,
                          _p(Ws[0]), _p(Bh[0]), _p(Ws[1]), _p(b1), _p(w2), n_out, ctx.final_act, _p(dx), (0 if ctx.level_major else K0), _p(ws),
                          _p(tgt[0][0]), tgt[0][0].stride(0), _p(tgt[0][2]), _p(tgt[1][0]), tgt[1][0].stride(0), _p(tgt[1][2]),
                          _p(t2[0]), (t2[0].stride(0) if t2[0] is not None else 0), _p(t2[2]), _stream(X))
            grads = []
            for tw, rw, tb, rb in tgt:
                grads += [rw, rb]
            return (dx, None, None, *grads)
        h1 = saved[2 + n]
        h2 = saved[3 + n] if n == 3 else None
        dlast = (d * out * (1.0 - out)).contiguous() if ctx.final_act == ACT_SIGMOID else d
        dpre0 = torch.empty((N, 64), device=dev, dtype=torch.float32)
        dpre1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if n == 3 else None
        dx = torch.empty_like(X) if ctx.need_dx else None
        w2 = Ws[2] if n == 3 else None
        with torch.cuda.device(dev):
            _lib.call("emer_rmlp_bwd", _p(dlast), n_out, _p(h1), _p(h2), L, F, K0, N, n, _p(Ws[0]), _p(Ws[1]), _p(w2), n_out,
                      _p(dpre1), _p(dpre0), _p(dx), (0 if ctx.level_major else K0), _stream(X))
        dpre = [dpre0, dpre1, dlast] if n == 3 else [dpre0, dlast]
        operands = [None, [seg(h1, 0, 64)], [seg(h2, 0, 64)]] if n == 3 else [None, [seg(h1, 0, 64)]]
        operands[0] = [seg_lm(X, 0)] if ctx.level_major else [seg(X, 0, K0)]
        grads = []
        for i in range(n):
            sw, sb = ctx.sinks[2 * i], ctx.sinks[2 * i + 1]
            tw, rw = _target(sw, tuple(Ws[i].shape), dev)
            has_b = ctx.needs_input_grad[3 + 2 * i + 1]
            tb, rb = _target(sb, (Ws[i].shape[0],), dev) if has_b else (None, None)
            wgrad(dpre[i], operands[i], Ws[i].shape[1], want_bias=has_b, out_w=tw, out_b=tb)
            grads += [rw, rb]
        return (dx, None, None, *grads)


def rmlp_bwd_fused_supported(weights, k0: int, n_feat: int, n_rows: int) -> bool:
    """True when the backward of this stack runs as emer_rmlp_bwd_fused (weight gradients in the kernel, hidden layers recomputed)."""
    n = len(weights)
    return bool(FUSED_WGRAD and FUSED_RMLP_WGRAD and n in (2, 3) and (weights[-1].shape[0] <= 16 or FUSED_RMLP_WIDE)
                and _lib.load().emer_rmlp_bwd_fused_workspace(n, int(k0), int(n_feat), int(n_rows), int(weights[-1].shape[0])) > 0)


def seq_mlp(x: Tensor, weights, biases, final_act: int = ACT_NONE) -> Tensor:
    """Fused nn.Sequential(Linear, ReLU, ..., Linear[, Sigmoid]); x [rows, K0] row-major.  Stacks of hidden width 64
    run on the register-resident kernels, everything else on the LDS-staged chain."""
    assert final_act in (ACT_NONE, ACT_SIGMOID)
    wb = [t for pair in zip(weights, biases) for t in pair]
    if x.shape[-1] % 4 == 0 and rmlp_supported(weights, x.shape[-1], 0):
        return _RMlpFn.apply(*_ng(x, False, final_act, *wb))
    return _SeqMLPFn.apply(*_ng(x, final_act, *wb))


def seq_mlp_lm(enc_lm: Tensor, weights, biases, final_act: int = ACT_NONE) -> Tensor:
    """The same stack fed by a LEVEL-MAJOR grid encoding [L, N, F] (no row-major copy of the encoding, the input gradient
    comes back level-major for the grid backward).  Requires ``rmlp_supported(weights, L * F, F)``."""
    assert final_act in (ACT_NONE, ACT_SIGMOID)
    wb = [t for pair in zip(weights, biases) for t in pair]
    return _RMlpFn.apply(*_ng(enc_lm, True, final_act, *wb))
