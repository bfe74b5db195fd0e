"""CPU ORACLE of the whole hot path -- test infrastructure, not product code.

A compact functional torch-CPU restatement of the reference's Python layer on top of the C oracle:
RadianceField / DensityField forward (radiance_fields/radiance_field.py:391-551, 825-841),
PropNetEstimator.sampling / compute_loss (third_party/nerfacc_prop_net.py:89-238), rendering
(radiance_fields/render_utils.py:48-287, training outputs) and one optimizer step
(train_emernerf.py:634-745).  Parameters are addressed by the reference's state_dict names.

Why it exists: /root/reference cannot travel to the GPU box, so the reference's Python cannot be the
run-time checker there.  This file is PINNED against the reference itself: tests/test_oracle_cpu.py replays
the golden vectors recorded from the reference's own code (tests/golden/make_golden.py) through it.
Supported: static / dynamic (+shadow) / flow (+temporal aggregation) / sky head / appearance embedding (per image or per
camera) / feature head + learnable PE map + feature sky head (BASELINE config 5; radiance_field.py:192-217,509-537,
render_utils.py:228-267).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import oracle as O


# ------------------------------------------------------------------------------------- pieces
def contract_points(pos: Tensor, aabb: Tensor, unbounded: bool) -> Tensor:
    """nerf_utils.py:13-28 + radiance_field.py:278-300."""
    lo, hi = aabb.reshape(-1)[:3], aabb.reshape(-1)[3:]
    x = (pos - lo) / (hi - lo)
    if unbounded:
        x = x * 2 - 1
        mag = torch.linalg.norm(x, ord=float("inf"), dim=-1, keepdim=True)
        x = torch.where(mag < 1, x, (2 - 1 / mag) * (x / mag))
        x = x / 4 + 0.5
    sel = ((x > 0.0) & (x < 1.0)).all(dim=-1).to(pos)
    return x * sel.unsqueeze(-1)


class _TruncExp(torch.autograd.Function):
    """nerf_utils.py:59-75."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(torch.clamp(ctx.saved_tensors[0], max=15))


def density_act(x: Tensor) -> Tensor:
    return _TruncExp.apply(x - 1)  # radiance_field.py:28


def dir_encode(d: Tensor, max_deg: int = 4) -> Tensor:
    """SinusoidalEncoder(3, 0, 4) (encodings.py:86-104)."""
    scales = torch.tensor([2.0 ** i for i in range(max_deg + 1)])
    xb = (d[..., None, :] * scales[:, None]).reshape(*d.shape[:-1], (max_deg + 1) * 3)
    return torch.cat([d, torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))], dim=-1)


class Params:
    """state_dict-name addressed parameters; grid metas are derived from the table sizes + encoder args."""

    def __init__(self, tensors: Dict[str, Tensor], grids: Dict[str, O.GridMeta], prefix: str = ""):
        self.t, self.grids, self.prefix = tensors, grids, prefix

    def __getitem__(self, k: str) -> Tensor:
        return self.t[self.prefix + k]

    def has(self, k: str) -> bool:
        return (self.prefix + k) in self.t

    def grid(self, name: str, x: Tensor) -> Tensor:
        return O.hashgrid(x.reshape(-1, x.shape[-1]), self[name + ".tcnn_encoding.params"], self.grids[self.prefix + name])

    def lin(self, name: str, x: Tensor) -> Tensor:
        return F.linear(x, self[name + ".weight"], self[name + ".bias"])


def mlp_seq(p: Params, name: str, x: Tensor, n_layers: int) -> Tensor:
    """nn.Sequential(Linear, ReLU, ..., Linear) with layers at indices 0, 2, 4."""
    for i in range(n_layers):
        x = p.lin(f"{name}.{2 * i}", x)
        if i < n_layers - 1:
            x = F.relu(x)
    return x


def mlp_skip(p: Params, name: str, x: Tensor) -> Tensor:
    """radiance_fields/mlp.py:38-46 with num_layers=3, skip_connections=[1]."""
    inp = x
    x = F.relu(p.lin(f"{name}.layers.0", x))
    x = F.relu(p.lin(f"{name}.layers.1", torch.cat([x, inp], -1)))
    return p.lin(f"{name}.layers.2", x)


# ------------------------------------------------------------------------------------- fields
def density_field(p: Params, positions: Tensor, aabb: Tensor, unbounded: bool = True) -> Tensor:
    """DensityField.forward (radiance_field.py:825-841) -> density [..., 1]."""
    x = contract_points(positions, aabb, unbounded)
    enc = p.grid("xyz_encoder", x)
    return density_act(mlp_seq(p, "base_mlp", enc, 2)).view(*positions.shape[:-1], 1)


def radiance_field(p: Params, positions: Tensor, directions: Optional[Tensor], data: Dict[str, Tensor], aabb: Tensor,
                   geo_dim: int = 64, time_diff: float = 0.0, training: bool = True, return_density_only: bool = False,
                   noise_fn: Optional[Callable] = None, cam_embedding: bool = False) -> Dict[str, Tensor]:
    """RadianceField.forward (radiance_field.py:391-551)."""
    out = {}
    lead = positions.shape[:-1]
    normed = contract_points(positions, aabb, True)
    feats = mlp_seq(p, "base_mlp", p.grid("xyz_encoder", normed), 2).view(*lead, -1)
    geo = feats[..., :geo_dim]
    emb_key = "cam_idx" if cam_embedding else "img_idx"   # :637-643
    static_density = density_act(geo[..., 0])
    has_t = "normed_timestamps" in data or "lidar_normed_timestamps" in data
    dynamic = p.has("dynamic_xyz_encoder.tcnn_encoding.params") and has_t

    def dyn_hash(npos, t):
        if t.shape[-1] != 1:
            t = t.unsqueeze(-1)
        tp = torch.cat([npos, t], -1)
        enc = p.grid("dynamic_xyz_encoder", tp).view(*tp.shape[:-1], -1)
        return mlp_seq(p, "dynamic_base_mlp", enc, 2), enc

    def flow_hash(npos, t):
        if t.shape[-1] != 1:
            t = t.unsqueeze(-1)
        tp = torch.cat([npos, t], -1)
        return mlp_seq(p, "flow_mlp", p.grid("flow_xyz_encoder", tp).view(*tp.shape[:-1], -1), 3)

    if dynamic:
        ts = data["normed_timestamps"] if "normed_timestamps" in data else data["lidar_normed_timestamps"]
        dyn_feats, _ = dyn_hash(normed, ts)
        if p.has("flow_xyz_encoder.tcnn_encoding.params"):
            flow = flow_hash(normed, ts)
            ff, bf = flow[..., :3], flow[..., 3:]
            out["forward_flow"], out["backward_flow"] = ff, bf
            t1 = ts.unsqueeze(-1) if ts.shape[-1] != 1 else ts  # temporal_aggregation, :553-620
            noise = noise_fn(ff) if noise_fn is not None else (torch.rand_like(ff)[..., 0:1] if training else torch.ones_like(ff)[..., 0:1])
            fpos, bpos = contract_points(positions + ff * noise, aabb, True), contract_points(positions + bf * noise, aabb, True)
            ft, bt = torch.clamp(t1 + time_diff * noise, 0, 1.0), torch.clamp(t1 - time_diff * noise, 0, 1.0)
            f_feats, _ = dyn_hash(fpos, ft)
            b_feats, _ = dyn_hash(bpos, bt)
            out["forward_pred_backward_flow"] = flow_hash(fpos, ft)[..., 3:]
            out["backward_pred_forward_flow"] = flow_hash(bpos, bt)[..., :3]
            dyn_feats = (dyn_feats + 0.5 * f_feats + 0.5 * b_feats) / 2.0
        dyn_geo = dyn_feats[..., :geo_dim]
        dynamic_density = density_act(dyn_geo[..., 0])
        out.update(density=static_density + dynamic_density, static_density=static_density, dynamic_density=dynamic_density)
    else:
        out["density"] = static_density
    if return_density_only:
        return out

    if directions is not None:
        h = dir_encode((directions + 1.0) / 2.0)  # query_rgb, :622-658
        if p.has("appearance_embedding.weight"):
            h = torch.cat([h, F.embedding(data[emb_key], p["appearance_embedding.weight"])], -1)
        rgb = torch.sigmoid(mlp_skip(p, "rgb_head", torch.cat([h, geo], -1)))
        if dynamic:
            out["static_rgb"] = rgb
            out["dynamic_rgb"] = torch.sigmoid(mlp_skip(p, "rgb_head", torch.cat([h, dyn_geo], -1)))
        else:
            out["rgb"] = rgb
    if dynamic and p.has("shadow_head.0.weight"):
        out["shadow_ratio"] = torch.sigmoid(mlp_seq(p, "shadow_head", dyn_geo, 2))
    if p.has("sky_head.layers.0.weight") and directions is not None:  # query_sky on dirs[:, 0], :540-549,660-686
        dd = dir_encode(directions[:, 0])
        if p.has("appearance_embedding.weight"):
            dd = torch.cat([dd, F.embedding(data[emb_key][:, 0], p["appearance_embedding.weight"])], -1)
        out["rgb_sky"] = torch.sigmoid(mlp_skip(p, "sky_head", dd))
        if p.has("dino_sky_head.0.weight"):
            out["dino_sky_feat"] = mlp_seq(p, "dino_sky_head", dd, 3)
    if p.has("dino_head.0.weight"):  # feature head, :509-537
        if p.has("learnable_pe_map") and "pixel_coords" in data:
            pe = F.grid_sample(p["learnable_pe_map"], data["pixel_coords"].reshape(1, 1, -1, 2) * 2 - 1, align_corners=False,
                               mode="bilinear").squeeze(2).squeeze(0).permute(1, 0)
            out["dino_pe"] = p.lin("pe_head.0", pe)
        dino = mlp_seq(p, "dino_head", feats[..., geo_dim:], 3)
        if dynamic:
            out["static_dino_feat"], out["dynamic_dino_feat"] = dino, mlp_seq(p, "dino_head", dyn_feats[..., geo_dim:], 3)
        else:
            out["dino_feat"] = dino
    return out


# ---------------------------------------------------------------------------- sampling / render
def render_trans(t_starts, t_ends, sigmas):
    """nerfacc dense volrend."""
    sdt = sigmas * (t_ends - t_starts)
    cum = torch.cumsum(sdt, -1)
    trans = torch.exp(-torch.cat([torch.zeros_like(cum[..., :1]), cum[..., :-1]], -1))
    return trans, 1.0 - torch.exp(-sdt)


def sampling(level_fns: List[Callable], prop_samples, num_samples: int, n_rays: int, near: float, far: float,
             sampling_type: str, jitters: Optional[List[Tensor]], requires_grad: bool, cache: list):
    """PropNetEstimator.sampling (nerfacc_prop_net.py:89-179); jitters: one [R] tensor per resampling round
    (None entries / None list = centre of bin)."""
    cdfs = torch.cat([torch.zeros(n_rays, 1), torch.ones(n_rays, 1)], -1)
    s = cdfs
    jit = iter(jitters) if jitters is not None else None
    for i, (fn, n) in enumerate(zip(level_fns, prop_samples)):
        s = torch.from_numpy(O.importance_sample(s, cdfs.detach(), n, None if jit is None else next(jit)))
        t = torch.from_numpy(O.stot(s, near, far, sampling_type))
        t0, t1 = t[..., :-1], t[..., 1:]
        with torch.set_grad_enabled(requires_grad):
            sig = fn(t0, t1).squeeze(-1)
            trans, _ = render_trans(t0, t1, sig)
            cdfs = 1.0 - torch.cat([trans, torch.zeros_like(trans[..., :1])], -1)
            if requires_grad:
                cache.append((s, cdfs, i))
    s = torch.from_numpy(O.importance_sample(s, cdfs.detach(), num_samples, None if jit is None else next(jit)))
    t = torch.from_numpy(O.stot(s, near, far, sampling_type))
    if requires_grad:
        cache.append((s, None, None))
    return t[..., :-1], t[..., 1:]


def blur_stepfun(x, y, r):
    """nerfacc_prop_net.py:22-34."""
    xr, idx = torch.sort(torch.cat([x - r, x + r], -1))
    y1 = (torch.cat([y, torch.zeros_like(y[..., :1])], -1) - torch.cat([torch.zeros_like(y[..., :1]), y], -1)) / (2 * r)
    y2 = torch.cat([y1, -y1], -1).take_along_dim(idx[..., :-1], dim=-1)
    yr = torch.cumsum((xr[..., 1:] - xr[..., :-1]) * torch.cumsum(y2, -1), -1).clamp_min(0)
    return xr, torch.cat([torch.zeros_like(yr[..., :1]), yr], -1)


def sorted_interp_quad(x, xp, fpdf, fcdf):
    """nerfacc_prop_net.py:37-60 as written (masked max/min)."""
    mask = x[..., None, :] >= xp[..., :, None]

    def find(v, idx=False):
        v0, i0 = torch.max(torch.where(mask, v[..., None], v[..., :1, None]), -2)
        v1, i1 = torch.min(torch.where(~mask, v[..., None], v[..., -1:, None]), -2)
        return (v0, v1, i0, i1) if idx else (v0, v1)

    c0, c1, i0, i1 = find(fcdf, True)
    p0, p1 = fpdf.take_along_dim(i0, -1), fpdf.take_along_dim(i1, -1)
    x0, x1 = find(xp)
    off = torch.clip(torch.nan_to_num((x - x0) / (x1 - x0), 0), 0, 1)
    return c0 + (x - x0) * (p0 + p1 * off + p0 * (1 - off)) / 2


def prop_loss(cache: list, trans: Tensor, loss_scaler: float, pulse=(0.03, 0.003)) -> Tensor:
    """PropNetEstimator.compute_loss, anti-aliased branch (nerfacc_prop_net.py:181-238)."""
    s, _, _ = cache.pop()
    cdfs = (1.0 - torch.cat([trans, torch.zeros_like(trans[..., :1])], -1)).detach()
    wn = (cdfs[..., 1:] - cdfs[..., :-1]) / (s[..., 1:] - s[..., :-1])
    cs, ws, cds = [], [], []
    for r in pulse:
        c, w = blur_stepfun(s, wn, r)
        area = 0.5 * (w[..., 1:] + w[..., :-1]) * (c[..., 1:] - c[..., :-1])
        cs.append(c); ws.append(w)
        cds.append(torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, -1)], -1))
    loss = 0.0
    while cache:
        ps, pc, pid = cache.pop()
        wp = pc[..., 1:] - pc[..., :-1]
        w_s = torch.diff(sorted_interp_quad(ps, cs[pid], ws[pid], cds[pid]), dim=-1)
        loss = loss + ((w_s - wp).clamp_min(0) ** 2 / (wp + 1e-5)).mean()
    return loss * loss_scaler


def render(field_out: Dict[str, Tensor], t_starts: Tensor, t_ends: Tensor) -> Dict[str, Tensor]:
    """rendering() (render_utils.py:48-287), training outputs (no decomposition)."""
    density = field_out["density"]
    trans, alphas = render_trans(t_starts, t_ends, density)
    w = trans * alphas
    mid = (t_starts + t_ends) / 2.0
    opacity = w.sum(-1, keepdim=True).clamp(1e-6, 1.0)
    depth = (w * mid).sum(-1, keepdim=True) / opacity
    idx = torch.searchsorted(torch.cumsum(w, -1).detach().contiguous(), torch.full((w.shape[0], 1), 0.5), side="left")
    median = torch.gather(mid, -1, idx.clamp(0, w.shape[-1] - 1))
    res = {"depth": depth, "opacity": opacity, "median_depth": median}
    extras = {"weights": w, "trans": trans, "t_vals": mid, "t_dist": t_ends - t_starts, "density": density}
    for k in ("forward_flow", "backward_flow", "forward_pred_backward_flow", "backward_pred_forward_flow",
              "static_density", "dynamic_density"):
        if k in field_out:
            extras[k] = field_out[k]
    acc = lambda v: (w[..., None] * v).sum(-2)  # noqa: E731  accumulate_along_rays
    if "rgb" in field_out:
        res["rgb"] = acc(field_out["rgb"])
    elif "static_rgb" in field_out:
        sr = field_out["static_density"] / (density + 1e-6)
        dr = field_out["dynamic_density"] / (density + 1e-6)
        shadow = 0.0
        if "shadow_ratio" in field_out:
            shadow = field_out["shadow_ratio"]
            res["shadow_ratio"] = acc(shadow.square())
        res["rgb"] = acc(sr[..., None] * field_out["static_rgb"] * (1 - shadow) + dr[..., None] * field_out["dynamic_rgb"])
    if "rgb_sky" in field_out and "rgb" in res:
        res["rgb"] = res["rgb"] + field_out["rgb_sky"] * (1.0 - opacity)
    dino = None  # features, render_utils.py:228-267
    if "dino_feat" in field_out:
        dino = acc(field_out["dino_feat"])
    elif "static_dino_feat" in field_out:
        dino = acc(sr[..., None] * field_out["static_dino_feat"] + dr[..., None] * field_out["dynamic_dino_feat"])
    if dino is not None:
        if "dino_sky_feat" in field_out:
            dino = dino + field_out["dino_sky_feat"] * (1.0 - opacity)
        if "dino_pe" in field_out:
            res["dino_pe_free"] = dino.clone()
            res["dino_pe"] = field_out["dino_pe"]
            dino = dino + field_out["dino_pe"]
        res["dino_feat"] = dino
    res["extras"] = extras
    return res


def pixel_step_loss(res: Dict[str, Tensor], data: Dict[str, Tensor]) -> Tensor:
    """The pixel-ray losses of train_emernerf.py:655-716 with configs/default_config.yaml's coefficients: rgb L2 (1), opacity-based
    sky BCE (0.001), dynamic-density and shadow sparsity (0.01 each, loss/base.py:394-398), feature L2 (0.5), flow cycle
    consistency (0.5 * mean * 0.01)."""
    loss = F.mse_loss(res["rgb"], data["pixels"]) + 0.001 * F.binary_cross_entropy(res["opacity"].squeeze(-1), 1 - data["sky_masks"].float())
    ex = res["extras"]
    if "dynamic_density" in ex:
        loss = loss + 0.01 * ex["dynamic_density"].mean()
    if "shadow_ratio" in res:
        loss = loss + 0.01 * res["shadow_ratio"].mean()
    if "dino_feat" in res and "features" in data:
        loss = loss + 0.5 * F.mse_loss(res["dino_feat"], data["features"])
    if "forward_flow" in ex:
        loss = loss + 0.01 * 0.5 * ((ex["forward_flow"].detach() + ex["forward_pred_backward_flow"]) ** 2
                                    + (ex["backward_flow"].detach() + ex["backward_pred_forward_flow"]) ** 2).mean()
    return loss


# ---------------------------------------------------------------------------------- whole path
class RefPath:
    """Holds parameters (reference names) and evaluates render_rays / one training step on CPU."""

    def __init__(self, model_state: Dict[str, Tensor], prop_states: List[Dict[str, Tensor]], grids: Dict[str, O.GridMeta],
                 aabb, geo_dim: int = 64, time_diff: float = 0.0, cam_embedding: bool = False):
        self.cam_embedding = cam_embedding
        self.t: Dict[str, Tensor] = {}
        for k, v in model_state.items():
            self.t["model/" + k] = v.clone().float().requires_grad_(v.is_floating_point() and k not in (
                "aabb", "training_timesteps", "feats_reduction_mat", "feat_color_min", "feat_color_max"))
        for i, st in enumerate(prop_states):
            for k, v in st.items():
                self.t[f"prop{i}/" + k] = v.clone().float().requires_grad_(k != "aabb")
        self.grids, self.aabb = grids, torch.as_tensor(aabb, dtype=torch.float32)
        self.model = Params(self.t, grids, "model/")
        self.props = [Params(self.t, grids, f"prop{i}/") for i in range(len(prop_states))]
        self.geo_dim, self.time_diff = geo_dim, time_diff
        self.cache: list = []

    def trainable(self, prefix: str = "") -> List[Tensor]:
        return [v for k, v in self.t.items() if k.startswith(prefix) and v.requires_grad]

    def render_rays(self, data: Dict[str, Tensor], num_samples: int, prop_samples, near=0.1, far=1000.0,
                    sampling_type="uniform_lindisp", jitters=None, noise_fn=None, requires_grad=False, training=True, prefix=""):
        """render_rays (render_utils.py:290-389), single chunk."""
        o, d = data[prefix + "origins"], data[prefix + "viewdirs"]
        R = o.shape[0]
        last = self.props[-1]  # the reference's late-binding lambda: every level queries the LAST proposal net

        def prop_fn(t0, t1):
            pos = o[:, None, :] + d[:, None, :] * (t0 + t1)[..., None] / 2.0
            return density_field(last, pos, self.aabb)

        self.cache = []
        with torch.no_grad():
            t0, t1 = sampling([prop_fn] * len(self.props), prop_samples, num_samples, R, near, far, sampling_type, jitters,
                              requires_grad, self.cache)
        S = t0.shape[-1]
        pos = o[:, None, :] + d[:, None, :].expand(-1, S, -1) * (t0 + t1)[..., None] / 2.0
        sub = {k: v[..., None].expand(*v.shape, S) for k, v in data.items() if v.dim() == 1}
        if "pixel_coords" in data:
            sub["pixel_coords"] = data["pixel_coords"]   # per ray (render_utils.py:339-340)
        fo = radiance_field(self.model, pos, d[:, None, :].expand(-1, S, -1), sub, self.aabb, self.geo_dim, self.time_diff,
                            training, return_density_only=(prefix == "lidar_"), noise_fn=noise_fn, cam_embedding=self.cam_embedding)
        return render(fo, t0, t1)

    def train_step(self, data, opt_main, opt_prop, num_samples, prop_samples, jitters=None, loss_scale=1024.0, prop_grad=True,
                   noise_fn=None):
        """One pixel-ray optimizer step (train_emernerf.py:634-745): rgb L2 + opacity sky loss + the regularisers of the
        dynamic / flow / feature models."""
        res = self.render_rays(data, num_samples, prop_samples, jitters=jitters, requires_grad=prop_grad, noise_fn=noise_fn)
        if prop_grad:
            pl = prop_loss(self.cache, res["extras"]["trans"], loss_scale)
            opt_prop.zero_grad(); pl.backward(); opt_prop.step()
        loss = pixel_step_loss(res, data)
        opt_main.zero_grad()
        (loss * loss_scale).backward()
        opt_main.step()
        return float(loss), res
