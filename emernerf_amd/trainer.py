"""One optimizer step of the hot path, data-parallel over ray batches.

Shape of a step (train_emernerf.py:634-745, pixel rays): proposal_requires_grad schedule -> render_rays
(2 proposal rounds + final resample -> field -> compositing) -> proposal loss -> rgb + sky losses ->
backward -> optimizer step.  What is new relative to the reference (which is single-GPU):

  * every trainable tensor (hash tables, MLPs, embeddings; main model AND proposal nets) is a view into
    ONE contiguous fp32 buffer, and so is every gradient.  Data parallelism exchanges that buffer's gradients over
    ``torch.distributed`` (backend "nccl" = RCCL over xGMI) in up to THREE buckets per step, two of them hidden:
    the trained proposal net's range right after ITS backward (1 step in 6, while the whole main backward is still
    ahead), the dense MLP / embedding ranges of the main model when the last table backward starts, and -- the only
    exposed one -- the main model's hash tables after the backward.  ``dp_mode="rs_ag"`` (env EMER_DP_MODE) replaces the
    exchange by reduce-scatter -> Adam on this rank's 1/W shard of (params, m, v) -> all-gather of the parameters;
  * Adam runs as one fused HIP kernel per optimizer group over that buffer (torch.optim.Adam semantics of
    builders.py:50-60,114-120, including the reference's never-unscaled GradScaler(2**10) quirk,
    train_emernerf.py:475-476,742-745: gradients enter Adam multiplied by 1024);
  * the proposal-net update and the main update are applied after the exchange; neither backward depends on the
    other's updated weights (samples are detached, nerfacc_prop_net.py:89), so this is numerically the reference's order.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor

from . import fused, ops
from .prop_net import PropNetEstimator, get_proposal_requires_grad_fn
from .radiance_field import DensityField, RadianceField, build_density_field, build_radiance_field_from_cfg
from .render_utils import render_rays


def ns(**kw):
    return SimpleNamespace(**{k: ns(**v) if isinstance(v, dict) else v for k, v in kw.items()})


AABB = [-20.0, -40.0, 0.0, 80.0, 40.0, 20.0]  # configs/default_config.yaml:42


def model_config(kind: str = "static", num_train_timesteps: int = 50, num_cams: int = 1):
    """Model configs named by BASELINE.json.

    static  (configs[1]): HashEncoder defaults D3/L16/F2/T2^19 (encodings.py:110-118), base_mlp 32->64->64,
             rgb head 113->64->[64+113]->64->3, sky head, per-image appearance embedding.
    dynamic (configs[2]): configs/default_dynamic.yaml on top of default_config.yaml:60-105.
    flow    (configs[3]): configs/default_flow.yaml.
    feature (configs[4]): flow + enable_feature_head + enable_learnable_pe (default_config.yaml:19,93-94: 64-d features).
    """
    if kind == "static":
        xyz = dict(type="HashEncoder", n_input_dims=3, n_levels=16, n_features_per_level=2, base_resolution=16,
                   max_resolution=2048, log2_hashmap_size=19)
    else:
        xyz = dict(type="HashEncoder", n_input_dims=3, n_levels=10, n_features_per_level=4, base_resolution=16,
                   max_resolution=8192, log2_hashmap_size=20)
    dyn = kind in ("dynamic", "flow", "feature")
    feat = kind == "feature"  # BASELINE configs[4]: the flow model + DINO feature head (64-d after PCA) + learnable PE map
    return ns(
        xyz_encoder=xyz,
        dynamic_xyz_encoder=dict(type="HashEncoder", n_input_dims=4, n_levels=10, n_features_per_level=4, base_resolution=32,
                                 max_resolution=8192, log2_hashmap_size=18),
        neck=dict(base_mlp_layer_width=64, geometry_feature_dim=64, semantic_feature_dim=64),
        head=dict(head_mlp_layer_width=64, enable_cam_embedding=False, enable_img_embedding=True, appearance_embedding_dim=16,
                  enable_sky_head=True, enable_feature_head=feat, feature_embedding_dim=64, feature_mlp_layer_width=64,
                  enable_learnable_pe=True, enable_dynamic_branch=dyn, enable_shadow_head=dyn,
                  interpolate_xyz_encoding=True, enable_temporal_interpolation=False, enable_flow_branch=kind in ("flow", "feature")),
        unbounded=True, num_cams=num_cams, num_train_timesteps=num_train_timesteps)


def render_config(num_samples: int = 128, prop_samples=(128, 64), chunk: int = 16384):
    return ns(nerf=dict(sampling=dict(num_samples=num_samples),
                        propnet=dict(num_samples_per_prop=list(prop_samples), near_plane=0.1, far_plane=1000.0,
                                     sampling_type="uniform_lindisp")),
              render=dict(render_chunk_size=chunk))


# default_config.yaml:51-58 / builders.py:99-108 (base_resolutions_per_prop is ignored by the reference)
PROP_KW = [dict(n_levels=8, max_resolution=512, log2_hashmap_size=20, n_features_per_level=1),
           dict(n_levels=8, max_resolution=2048, log2_hashmap_size=20, n_features_per_level=1)]


def synthetic_rays(R: int, device, seed: int = 0, n_timesteps: int = 50, num_cams: int = 1, feature_dim: int = 0) -> Dict[str, Tensor]:
    """Seeded synthetic ray batch of SURVEY.md section 8d / BASELINE.md 2.2 (no dataset on the box)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.stack([torch.rand(R, generator=g) * 60, torch.rand(R, generator=g) * 4 - 2, torch.rand(R, generator=g) + 1.5], -1)
    d = F.normalize(torch.tensor([1.0, 0.0, 0.0]) + 0.6 * torch.randn(R, 3, generator=g), dim=-1)
    img_idx = torch.randint(0, n_timesteps * num_cams, (R,), generator=g)
    data = {"origins": o, "viewdirs": d, "direction_norms": torch.ones(R, 1), "pixel_coords": torch.rand(R, 2, generator=g),
            "normed_timestamps": torch.randint(0, n_timesteps, (R,), generator=g).float() / (n_timesteps - 1),
            "img_idx": img_idx, "cam_idx": img_idx % num_cams, "pixels": torch.rand(R, 3, generator=g),
            "sky_masks": (torch.rand(R, generator=g) < 0.15).float()}
    if feature_dim:
        data["features"] = torch.rand(R, feature_dim, generator=g)
    return {k: v.to(device) for k, v in data.items()}


def synthetic_lidar_rays(R: int, device, seed: int = 0, n_timesteps: int = 50) -> Dict[str, Tensor]:
    """Seeded lidar-ray batch with the key names of the reference's lidar source (datasets/base/lidar_source.py:223-308,
    prefix "lidar_"): a spinning sensor on the ego path; ranges ~U(2, 70) m, 3 % dropped returns (range 0)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.stack([torch.rand(R, generator=g) * 60, torch.zeros(R), torch.full((R,), 2.0)], -1)
    az, el = torch.rand(R, generator=g) * 6.2831853, (torch.rand(R, generator=g) - 0.7) * 0.45
    d = torch.stack([torch.cos(az) * torch.cos(el), torch.sin(az) * torch.cos(el), torch.sin(el)], -1)
    rng = torch.rand(R, generator=g) * 68 + 2
    rng[torch.rand(R, generator=g) < 0.03] = 0.0
    data = {"lidar_origins": o, "lidar_viewdirs": d, "lidar_ranges": rng[:, None],
            "lidar_normed_timestamps": torch.randint(0, n_timesteps, (R,), generator=g).float() / (n_timesteps - 1)}
    return {k: v.to(device) for k, v in data.items()}


class FlatParams:
    """Re-home every parameter (and gradient) of a list of modules into one contiguous fp32 buffer.

    groups: {name: [modules]} -> contiguous [start, end) ranges, e.g. "main" and "prop".
    """

    def __init__(self, groups: Dict[str, List[torch.nn.Module]], device, align: int = 1):
        """``align``: every group's length is padded up to a multiple of it (zero parameters with zero gradients), so a group
        splits into equal shards for reduce-scatter / all-gather (align = 4 * world_size keeps shards 16-byte aligned)."""
        plist, self.ranges = [], {}
        off = 0
        for gname, mods in groups.items():
            start = off
            for m in mods:
                for p in m.parameters():
                    plist.append((p, off))
                    off += p.numel()
            off = start + -(-(off - start) // align) * align
            self.ranges[gname] = (start, off)
        self.numel = off
        self.params = torch.zeros(off, device=device, dtype=torch.float32)
        self.grads = torch.zeros(off, device=device, dtype=torch.float32)
        for p, o in plist:
            n = p.numel()
            self.params[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
        self._plist = plist
        # hash tables (large 1-D parameters named "...tcnn_encoding.params") vs everything else, merged into few ranges
        names = {id(p): n for m in [m for mods in groups.values() for m in mods] for n, p in m.named_parameters()}
        self._table_offsets = [(p, o) for p, o in plist if names.get(id(p), "").endswith("tcnn_encoding.params")]
        self._tables = [p for p, _ in self._table_offsets]
        table_ids = {id(p) for p in self._tables}
        self._dense_ranges = []
        for p, o in plist:
            if id(p) in table_ids:
                continue
            if self._dense_ranges and self._dense_ranges[-1][1] == o:
                self._dense_ranges[-1][1] = o + p.numel()
            else:
                self._dense_ranges.append([o, o + p.numel()])

    def zero_grad(self):
        """Zero what is ACCUMULATED into (MLP weights, embeddings: a few hundred KB) and only mark the hash tables:
        their gradient is overwritten by the owner-computes grid backward (``ops._HashGridLMFn``), which writes every
        entry exactly once, so zeroing 90 MB per step would be wasted HBM traffic.  A table whose backward did not run
        this step is zeroed lazily by ``finish_grads`` before anything reads it."""
        torch._foreach_zero_([self.grads[a:b] for a, b in self._dense_ranges])  # one multi-tensor launch
        for p, o in self._plist:  # autograd may have replaced .grad; re-pin the views
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self.grads[o:o + p.numel()].view(p.shape)
        for p in self._tables:
            p._emer_grad_fresh = True
            p._emer_pending_evals = 0   # (a forward without a backward -- an eval render between steps -- must not shift the count)

    def finish_grads(self, group: str) -> None:
        """Tables of ``group`` that no backward wrote this step get their zero gradient now."""
        a, b = self.ranges[group]
        for p, o in self._table_offsets:
            if a <= o < b and getattr(p, "_emer_grad_fresh", False):
                self.grads[o:o + p.numel()].zero_()
                p._emer_grad_fresh = False


def capture_main_grid_positions(trainer: "Trainer", data: Dict[str, Tensor]) -> Tensor:
    """Contracted positions [R*S, 3] the MAIN grid is evaluated at during one training step on ``data`` (the
    proposal-resampled "training distribution" of the grid kernels): tests and profiling tools feed the grid kernels
    with it.  Runs one optimizer step."""
    enc = trainer.model.xyz_encoder.tcnn_encoding
    orig, cap = enc.forward_level_major, {}

    def hook(x):
        cap["x"] = x.detach().reshape(-1, enc.n_input_dims).contiguous()
        return orig(x)
    enc.forward_level_major = hook
    try:
        trainer.train_step(data)
    finally:
        enc.forward_level_major = orig
    return cap["x"]


def _subtract_ranges(span, holes):
    """[a, b) minus the sorted, disjoint ``holes`` -> list of remaining [lo, hi) ranges."""
    out, cur = [], span[0]
    for lo, hi in sorted(holes):
        if hi <= span[0] or lo >= span[1]:
            continue
        if lo > cur:
            out.append((cur, lo))
        cur = max(cur, hi)
    if cur < span[1]:
        out.append((cur, span[1]))
    return out


def agree_any(flag: bool, device) -> bool:
    """True on EVERY rank when ``flag`` is true on ANY rank (one MAX all-reduce of a scalar; no-op without a process group).  Used where
    ranks must take the same branch although the condition is local: a failed hipGraph capture switches a rank to eager launches, whose
    gradient buckets differ from the replaying ranks' -- so either all ranks switch or none does."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device, dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0.0)


def lr_factor(step: int, num_iters: int) -> float:
    """ChainedScheduler(LinearLR(0.01 -> 1 over num_iters//10), MultiStepLR(gamma 0.33)) of builders.py:64-89."""
    warm = num_iters // 10
    f = 0.01 + (1.0 - 0.01) * min(step, warm) / max(warm, 1)
    milestones = [num_iters // 2, num_iters * 3 // 4, num_iters * 9 // 10]
    if num_iters >= 10000:
        milestones.insert(0, num_iters // 4)
    return f * (0.33 ** sum(step >= m for m in milestones))


class Trainer:
    """Owns the model, proposal nets, estimator, flat buffers and optimizer state of one rank."""

    def __init__(self, kind: str = "static", device="cuda:0", num_samples: int = 128, prop_samples=(128, 64), lr: float = 0.01,
                 weight_decay: float = 1e-5, num_iters: int = 25000, loss_scale: float = 1024.0, seed: int = 0,
                 world_size: int = 1, table_init: Optional[float] = None, use_graph: bool = False, table_dtype: str = "f32",
                 dp_mode: Optional[str] = None):
        self.device = torch.device(device)
        self.kind = kind
        torch.manual_seed(seed)  # identical initial parameters on every rank
        self.cfg = model_config(kind, num_cams=3 if kind == "feature" else 1)
        self.rcfg = render_config(num_samples, prop_samples)
        self.model: RadianceField = build_radiance_field_from_cfg(self.cfg, verbose=False)
        self.model.set_aabb(AABB)
        if self.model.dynamic_xyz_encoder is not None:
            self.model.register_normalized_training_timesteps(torch.linspace(0, 1, self.cfg.num_train_timesteps),
                                                              time_diff=1 / self.cfg.num_train_timesteps)
        self.props: List[DensityField] = [build_density_field(aabb=AABB, unbounded=True, **kw) for kw in PROP_KW]
        if table_init is not None:  # "trained-like" tables instead of tcnn's +-1e-4 init
            g = torch.Generator().manual_seed(seed + 1)
            with torch.no_grad():
                for m in [self.model] + self.props:
                    for n, p in m.named_parameters():
                        if n.endswith("tcnn_encoding.params"):
                            p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 2 * table_init)
        # table precision: "f32" is what the reference runs; "f16" = tcnn's half-precision tables -- fp32 master
        # parameters cast to fp16 per call, fp32 gradient accumulation through the owner-computes backward
        assert table_dtype in ("f32", "f16")
        self.table_dtype = table_dtype
        if table_dtype == "f16":
            for mod in [self.model] + self.props:
                for sub in mod.modules():
                    if hasattr(sub, "desc") and hasattr(sub, "params") and hasattr(sub, "dtype"):
                        sub.dtype = torch.float16
        self.model.to(self.device)
        for p in self.props:
            p.to(self.device)
        self.estimator = PropNetEstimator(None, None).to(self.device)
        # The reference's late-binding closures make every proposal level query the LAST proposal network
        # (render_utils.py:356-358); the earlier ones never receive a gradient, their .grad stays None and torch's Adam
        # skips them (no weight decay, no moment update).  They live in their own range so the fused Adam skips them too.
        import os as _os
        self.dp_mode = dp_mode or _os.environ.get("EMER_DP_MODE", "allreduce")
        # [r6] EMER_DP_SINGLE=1 (or dp_mode="single"): BASELINE.json's literal exchange -- ONE all-reduce of the step's gradient range after
        # the backward, no bucket launched from inside it (what a hipGraph-replayed step does anyway), also with eager launches, so that
        # a multi-GPU run can A/B "one all-reduce" against the bucketed default in one command
        if self.dp_mode == "allreduce" and _os.environ.get("EMER_DP_SINGLE") == "1":
            self.dp_mode = "single"
        assert self.dp_mode in ("allreduce", "rs_ag", "single"), self.dp_mode
        # test hook, decided ONCE (a per-call fallback would let one rank issue all_reduce while its peers sit in reduce_scatter):
        # a backend without reduce-scatter can emulate it with an all-reduce when the environment says so explicitly
        self._rs_emulate = _os.environ.get("EMER_DP_RS_EMULATE") == "1"
        self.dp_debug = _os.environ.get("EMER_DP_DEBUG") == "1"   # check the ordering assumption of the early bucket (no overlap then)
        self.comm_events = None    # bench.py: list of (start, end) HIP events around the EXPOSED part of the exchange
        self.flat = FlatParams({"main": [self.model], "prop": self.props[-1:], "prop_idle": self.props[:-1]}, self.device,
                               align=4 * max(world_size, 1) if self.dp_mode == "rs_ag" else 1)
        # this trainer owns every gradient buffer (views of flat.grads, zeroed each step, no parameter hooks), so the
        # fused heads may accumulate weight gradients straight into .grad: enabled around ITS forward+backward only
        # (fused.grad_sinks), never process-wide
        # (fused.SIDE_STREAM -- weight gradients on a second stream, overlapping the grid backward -- is implemented and
        # tested but OFF: measured 3 % slower on MI355X; the grid backward's workgroups fill the LDS of every CU, so
        # the weight-gradient workgroups only delay them)
        self.m = torch.zeros_like(self.flat.params)
        self.v = torch.zeros_like(self.flat.params)
        self.opt_steps = {"main": 0, "prop": 0}
        self.lr, self.wd, self.num_iters, self.loss_scale = lr, weight_decay, num_iters, loss_scale
        self.world_size = world_size
        self.use_graph, self._graphs, self._static_data = use_graph, {}, None
        self.requires_grad_fn = get_proposal_requires_grad_fn()
        self.step_count = 0
        # LR-scheduler ticks: the reference calls scheduler.step() after the pixel optimizer step AND after the lidar one
        # (train_emernerf.py:745, :826), so with lidar supervision the warm-up and the milestones advance twice per iteration
        self.sched_ticks = 0
        # early all-reduce bucket: the dense (non-table) parameters of the main model
        a, b = self.flat.ranges["main"]
        self._early_ranges = [(lo, hi) for lo, hi in self.flat._dense_ranges if a <= lo and hi <= b]
        self._early_done, self._early_work = False, []
        self._prop_work = None  # async all-reduce of the trained proposal net's range (launched right after ITS backward)
        self._hold_buckets = False  # graph warm-up / capture: no collectives from inside the forward+backward
        self._one = torch.ones((), device=self.device, dtype=torch.float32)
        # the exchange runs when there is more than one rank -- or when EMER_DP_FORCE=1 asks for it on a single rank (a 1-GPU box can
        # then execute the real RCCL collectives of both modes, trivially: tests/test_multi_gpu.py)
        self._dp_on = world_size > 1 or _os.environ.get("EMER_DP_FORCE") == "1"
        self._table_work, self._table_ranges = [], []   # collectives of table level ranges launched from inside the table's backward
        self._table_snapshots = []
        if self._dp_on:
            assert dist.is_initialized(), "data-parallel exchange needs an initialised torch.distributed process group"
            tab = self.model.xyz_encoder.tcnn_encoding.params
            tab._emer_before_table_grad = self._launch_early_bucket
            # [r4] The exposed exchange of a step used to be the whole static table (48 MB at configs[1]) AFTER its backward, the
            # longest kernel of the step, with nothing left to hide it behind.  The owner-computes backward now runs as two launches
            # over a partition of the levels: the finest levels that fill one round of the resident owners first (cfg 2: levels
            # 12..15, a contiguous 34 % of the table, 157 us), whose all-reduce then overlaps the second launch (411 us; the pair
            # costs +6 % over the single launch, a cut by bytes +26 %: profiles/r04_table_split.txt).  EMER_DP_SPLIT_TABLE=0: off;
            # EMER_DP_SPLIT_LEVEL=k: cut there instead.
            if self.dp_mode == "allreduce" and _os.environ.get("EMER_DP_SPLIT_TABLE", "1") != "0":
                desc = self.model.xyz_encoder.tcnn_encoding.desc
                k = int(_os.environ.get("EMER_DP_SPLIT_LEVEL", "0")) or ops.sliced_split_level(desc)
                if 0 < k < desc.n_levels and ops.sliced_supported(desc):
                    tab._emer_table_split = (k, self._launch_table_bucket)
            # the other tables of the main model (dynamic and flow xyzt grids): their backwards run BEFORE the static table's (their
            # encoders come later in the forward), so each one's all-reduce starts when its last backward of the step has written
            # it and runs behind the backward kernels that follow -- three serial collectives after the backward before [r4]
            if self.dp_mode == "allreduce" and _os.environ.get("EMER_DP_SPLIT_TABLE", "1") != "0":
                a, b = self.flat.ranges["main"]
                for p, o in self.flat._table_offsets:
                    if a <= o < b and p is not tab:
                        p._emer_after_table_grad = self._launch_whole_table_bucket
        self.model.train(); self.estimator.train()
        for p in self.props:
            p.train()

    def set_step(self, step: int, ticks_per_iter: int = 1) -> None:
        """Fast-forward (or restore on resume) the iteration counter AND the LR-schedule ticks that go with it:
        ``ticks_per_iter`` = 2 when every iteration also takes a lidar step (train_emernerf.py:745,826), else 1.  The proposal
        schedule's ``since_last`` counter depends on its whole history, so it is replayed from step 0 (``ticks_per_iter`` calls per
        step, as ``train_step`` / ``lidar_step`` make them): after ``set_step(k)`` the trainer is in the state k iterations leave."""
        self.step_count = int(step)
        self.sched_ticks = int(step) * int(ticks_per_iter)
        sched = self.requires_grad_fn
        fresh = type(sched)(sched.target, sched.num_steps)
        for s_ in range(int(step)):
            for _ in range(int(ticks_per_iter)):
                fresh(s_)
        sched.since_last = fresh.since_last

    def state_dict(self) -> Dict[str, object]:
        """Optimizer-side state of this rank (parameters live in the modules' own state_dicts): a SNAPSHOT -- the moments are cloned,
        so a dictionary held across further steps does not change -- together with the layout of the flat buffer it belongs to
        (group ranges incl. the rs_ag padding, world size, exchange mode), the loss scale and the length of the LR schedule."""
        return {"m": self.m.clone(), "v": self.v.clone(), "opt_steps": dict(self.opt_steps), "step_count": self.step_count,
                "sched_ticks": self.sched_ticks, "since_last": self.requires_grad_fn.since_last,
                "ranges": {k: tuple(v) for k, v in self.flat.ranges.items()}, "world_size": self.world_size, "dp_mode": self.dp_mode,
                "loss_scale": self.loss_scale, "num_iters": self.num_iters}

    def load_state_dict(self, sd: Dict[str, object], restore_schedule: bool = False) -> None:
        """Restore ``state_dict()``.  The moments are indexed by flat-buffer offset, so the saved layout must be this trainer's
        (same model kind, same rs_ag padding = same world size in that mode); dictionaries of rounds before the layout was recorded
        are accepted when the buffer length matches.  The length of the LR schedule and the loss scale are the CONSTRUCTOR's (the
        reference takes them from the config of the resuming run, not from the checkpoint: train_emernerf.py:475-476, builders.py:64-89);
        a checkpoint that was written with other values is reported with a warning, ``restore_schedule=True`` adopts its values."""
        if "ranges" in sd:
            mine = {k: tuple(v) for k, v in self.flat.ranges.items()}
            theirs = {k: tuple(v) for k, v in sd["ranges"].items()}
            if mine != theirs:
                raise ValueError(f"optimizer state was saved for flat-buffer ranges {theirs} (world_size {sd.get('world_size')}, "
                                 f"dp_mode {sd.get('dp_mode')}); this trainer has {mine} (world_size {self.world_size}, dp_mode {self.dp_mode})")
        if tuple(sd["m"].shape) != tuple(self.m.shape) or tuple(sd["v"].shape) != tuple(self.v.shape):
            raise ValueError(f"optimizer moments of {tuple(sd['m'].shape)} elements do not fit a flat buffer of {tuple(self.m.shape)}")
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.opt_steps = dict(sd["opt_steps"])
        self.step_count, self.sched_ticks = int(sd["step_count"]), int(sd["sched_ticks"])
        self.requires_grad_fn.since_last = int(sd["since_last"])
        for key, cast in (("loss_scale", float), ("num_iters", int)):
            if key in sd and cast(sd[key]) != getattr(self, key):
                if restore_schedule:
                    setattr(self, key, cast(sd[key]))
                else:
                    import warnings
                    warnings.warn(f"checkpoint was written with {key} = {sd[key]}, this trainer was constructed with {getattr(self, key)}: "
                                  f"keeping the constructor's (load_state_dict(..., restore_schedule=True) adopts the checkpoint's)")

    def _with_regularisers(self, base: Tensor, results, data, grad_scale: float = 1.0) -> Tensor:
        """``base`` + the regularisers of the dynamic / flow / feature models (``base`` itself for the static model): dynamic-density
        and shadow sparsity (loss/base.py:394-398, coefficient 0.01 each, default_config.yaml:144-152), feature L2 (coefficient 0.5,
        :141-143) and the flow cycle loss (train_emernerf.py:700-716: 0.5 * mean * 0.01) -- ONE kernel each way
        (``ops.reg_losses``: row N4 of SURVEY.md section 8f) instead of ~25 elementwise torch launches."""
        ex = results["extras"]
        flows = "forward_flow" in ex
        feat = "dino_feat" in results and "features" in data
        pair = getattr(ex["forward_pred_backward_flow"], "_emer_flow_pair", None) if flows else None
        if pair is not None:   # batched flow branch: the cycle term from the flow MLP's two outputs, unsliced
            return ops.reg_losses(base, dynamic_density=ex.get("dynamic_density"), shadow_ratio=results.get("shadow_ratio"),
                                  feat=results["dino_feat"] if feat else None, feat_gt=data["features"] if feat else None,
                                  flow_pair=pair, c_dyn=0.01, c_shadow=0.01, c_feat=0.5, c_cycle=0.01 * 0.5, grad_scale=grad_scale)
        return ops.reg_losses(base, dynamic_density=ex.get("dynamic_density"), shadow_ratio=results.get("shadow_ratio"),
                              feat=results["dino_feat"] if feat else None, feat_gt=data["features"] if feat else None,
                              forward_flow=ex["forward_flow"] if flows else None,
                              forward_pred_backward_flow=ex["forward_pred_backward_flow"] if flows else None,
                              backward_flow=ex["backward_flow"] if flows else None,
                              backward_pred_forward_flow=ex["backward_pred_forward_flow"] if flows else None,
                              c_dyn=0.01, c_shadow=0.01, c_feat=0.5, c_cycle=0.01 * 0.5, grad_scale=grad_scale)

    def losses(self, results, data) -> Tensor:
        """rgb L2 (loss/base.py:83-146, coef 1) + opacity-based sky BCE (loss/base.py:149-185, coef 0.001) + the regularisers."""
        loss = ops.pixel_loss(results["rgb"], results["opacity"], data["pixels"], data["sky_masks"], w_rgb=1.0, w_sky=0.001)  # one launch
        return self._with_regularisers(loss, results, data)

    def _scaled_losses(self, results, data):
        """(tensor to back-propagate, plain loss value): the loss scale is folded into the backward kernels of the pixel loss and of
        the regularisers, so a step needs no ``loss * scale`` launch, no backward of it, and -- with the persistent ``self._one`` as
        the seed -- no ones_like fill.  The VALUE of the returned tensor is the plain (unscaled) loss."""
        pix = ops.pixel_loss(results["rgb"], results["opacity"], data["pixels"], data["sky_masks"], w_rgb=1.0, w_sky=0.001,
                             grad_scale=self.loss_scale)
        total = self._with_regularisers(pix, results, data, grad_scale=self.loss_scale)
        return total, total.detach()

    def lidar_losses(self, results, data, step: int) -> Tensor:
        """Depth + line-of-sight supervision of the lidar step (train_emernerf.py:770-808 with
        configs/default_config.yaml:120-137: depth l2 coefficient 1, line of sight coefficient 0.1 from iteration 2000,
        margin decaying linearly from 6.0 to 2.5 m, coefficient halved every 5000 steps after the start) as ONE kernel each way."""
        start, e0, e1 = 2000, 6.0, 2.5
        w_sight = 0.0
        eps = e0
        if step > start:
            m = (e1 - e0) / (self.num_iters - start)
            eps = e1 if step > self.num_iters else m * step + (e0 - m * start)
            w_sight = 0.1 * (0.5 ** ((step - start) // 5000))  # train_emernerf.py:620-628
        loss = ops.lidar_loss(results["depth"], results["extras"]["weights"], data["lidar_ranges"], results["extras"]["t_vals"],
                              eps, 80.0, 1.0, w_sight)
        if "dynamic_density" in results["extras"]:  # the dynamic regulariser also supervises lidar rays (train_emernerf.py:797-802)
            loss = ops.reg_losses(loss, dynamic_density=results["extras"]["dynamic_density"], c_dyn=0.01)
        return loss

    def lidar_step(self, data: Dict[str, Tensor]) -> Dict[str, float]:
        """The second optimizer step of a reference iteration (train_emernerf.py:747-826): lidar rays through the same
        kernels (density only: no colour heads), depth + line-of-sight losses, its own backward and Adam step."""
        step = self.step_count
        prop_grad = self.requires_grad_fn(step)
        self.flat.zero_grad()
        with fused.grad_sinks(True):
            results = render_rays(radiance_field=self.model, proposal_estimator=self.estimator, proposal_networks=self.props,
                                  data_dict=data, cfg=self.rcfg, proposal_requires_grad=prop_grad, prefix="lidar_")
            if prop_grad:
                pl = self.estimator.compute_loss(results["extras"]["trans"], loss_scaler=self.loss_scale)
                pl.backward(gradient=self._seed(pl))
                self._launch_prop_bucket()
            loss = self.lidar_losses(results, data, step)
            scaled = loss * self.loss_scale
            scaled.backward(gradient=self._seed(scaled))
        fused.join_side_stream()
        self._exchange_grads(prop_grad)
        lr = self.lr * lr_factor(self.sched_ticks, self.num_iters)
        if prop_grad:
            self._adam("prop", lr)
        self._adam("main", lr)
        self.sched_ticks += 1
        return {"loss": loss.detach(), "prop_grad": prop_grad}

    def _seed(self, like: Tensor) -> Tensor:
        """A persistent 1.0 as the backward seed of a scalar loss (autograd would launch a ones_like fill per backward)."""
        return self._one if like.shape == self._one.shape else self._one.expand(like.shape)

    # ------------------------------------------------------------------------------------------- data parallel
    def _launch_early_bucket(self) -> None:
        """Called from the LAST table backward of a step (ops._HashGridLMFn.backward of the static grid, the first
        encoder of the forward pass): every MLP / embedding gradient of the main model is enqueued by now, so their
        (small) all-reduce starts here and runs on RCCL's stream while the grid backward -- the longest kernel of the
        step -- computes the table gradient.  Only the table bucket is exposed at the end of the backward."""
        if self._dp_on and self.dp_mode in ("rs_ag", "single"):
            return  # one reduce-scatter per group / one all-reduce after the backward
        if self._dp_on and self.dp_debug and not self._early_done and not self._hold_buckets:
            # debug: record what the early ranges hold NOW instead of reducing them; _exchange_grads checks nothing wrote later
            fused.join_side_stream()   # (as the real bucket does: launches moved to a side stream count as enqueued)
            self._early_snapshot = [self.flat.grads[a:b].clone() for a, b in self._early_ranges]
            return
        if self._dp_on and not self._early_done and not self._hold_buckets and not torch.cuda.is_current_stream_capturing():
            fused.join_side_stream()  # weight gradients written on the side stream (off by default) must be complete
            self._early_work = [dist.all_reduce(self.flat.grads[a:b], async_op=True) for a, b in self._early_ranges]
            self._early_done = True

    def _launch_table_bucket(self, param, lo: int, hi: int) -> None:
        """Called by the static table's backward between its two launches (ops._HashGridLMFn.backward): elements [lo, hi) of the
        table's gradient are final; their all-reduce starts now and overlaps the second launch.  Not inside a graph capture and not
        during its warm-up (then the whole table goes with the late bucket)."""
        if not self._dp_on or self.dp_mode != "allreduce" or self._hold_buckets or torch.cuda.is_current_stream_capturing():
            return
        base = next(o for p, o in self.flat._table_offsets if p is param)
        a, b = base + lo, base + hi
        if self.dp_debug:   # record instead of reducing: _exchange_grads checks that nothing wrote the range afterwards
            self._table_snapshots.append(((a, b), self.flat.grads[a:b].clone()))
            return
        self._table_work.append(dist.all_reduce(self.flat.grads[a:b], async_op=True))
        self._table_ranges.append((a, b))

    def _launch_whole_table_bucket(self, param) -> None:
        """Called by the last backward of a dynamic / flow table in a step, after it wrote the table's gradient (ops._after_table_grad)."""
        self._launch_table_bucket(param, 0, param.numel())

    def _launch_prop_bucket(self) -> None:
        """On the steps that train the proposal net its loss is back-propagated BEFORE the main loss, so its gradient range
        is final while the whole main backward (~2 ms) is still ahead: its all-reduce (40 MB at the metric configuration)
        starts here and is hidden completely.  Eager launches only (a collective cannot be captured into the step's graph)."""
        if self._dp_on and self.dp_mode != "allreduce":
            return
        if self._dp_on and self._prop_work is None and not self._hold_buckets and not torch.cuda.is_current_stream_capturing():
            fused.join_side_stream()
            self.flat.finish_grads("prop")
            a, b = self.flat.ranges["prop"]
            self._prop_work = dist.all_reduce(self.flat.grads[a:b], async_op=True)

    def _exchange_grads(self, prop_grad: bool) -> None:
        """The data-parallel exchange of a step (sum; 1/W is folded into Adam's grad_scale): the table bucket (and, on the
        steps that train them -- same schedule on every rank -- the proposal net's range), after the early MLP bucket.
        50 MB instead of 90 MB on the steps without proposal training at the metric configuration."""
        self.flat.finish_grads("main")
        if prop_grad:
            self.flat.finish_grads("prop")
        for (lo, hi), old in self._table_snapshots:   # EMER_DP_DEBUG=1: a table range must be final when its collective would start
            assert torch.equal(self.flat.grads[lo:hi], old), f"table gradient range [{lo}, {hi}) was written after its bucket would have been launched"
        self._table_snapshots = []
        snap = getattr(self, "_early_snapshot", None)
        if snap is not None:  # EMER_DP_DEBUG=1: the early ranges must be final when the last table backward starts
            for (lo, hi), old in zip(self._early_ranges, snap):
                assert torch.equal(self.flat.grads[lo:hi], old), \
                    f"gradient range [{lo}, {hi}) was written after the early bucket would have been launched"
            self._early_snapshot = None
        ev0 = ev1 = None
        if self._dp_on and self.comm_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if self._dp_on and self.dp_mode == "rs_ag":
            self._rs_shards = {}
            for grp in (["main", "prop"] if prop_grad else ["main"]):
                a, b = self.flat.ranges[grp]
                n = (b - a) // self.world_size
                r = dist.get_rank()
                shard = self.flat.grads[a + r * n: a + (r + 1) * n]
                if self._rs_emulate:   # EMER_DP_RS_EMULATE=1, test mode only: same result, more bytes
                    dist.all_reduce(self.flat.grads[a:b])
                else:                  # in place: the output shard is a slice of the input; errors propagate (no per-rank fallback)
                    dist.reduce_scatter_tensor(shard, self.flat.grads[a:b])
                self._rs_shards[grp] = (a + r * n, a + (r + 1) * n)
        elif self._dp_on:
            a, b = self.flat.ranges["main"]
            if prop_grad and self._prop_work is None:
                b = self.flat.ranges["prop"][1]  # main and the trained proposal net are adjacent in the flat buffer
            done = (self._early_ranges if self._early_done else []) + self._table_ranges
            if done:
                late = _subtract_ranges((a, b), done)
                for lo, hi in late:
                    dist.all_reduce(self.flat.grads[lo:hi])
                for w in self._early_work + self._table_work:
                    w.wait()
            else:
                dist.all_reduce(self.flat.grads[a:b])
            if self._prop_work is not None:
                self._prop_work.wait()
        if ev0 is not None:
            ev1.record()
            self.comm_events.append((ev0, ev1))
        self._early_done, self._early_work, self._prop_work = False, [], None
        self._table_work, self._table_ranges = [], []
        if self._dp_on:
            self.model.xyz_encoder.tcnn_encoding.params._emer_pending_evals = 0

    def _adam(self, group: str, lr: float):
        a, b = self.flat.ranges[group]
        self.opt_steps[group] += 1
        if self._dp_on and self.dp_mode == "rs_ag":
            # this rank owns 1/W of the group: update its shard (the only part of m / v it ever touches), then everyone
            # gathers the updated parameters
            lo, hi = self._rs_shards[group]
            ops.adam_step(self.flat.params[lo:hi], self.flat.grads[lo:hi], self.m[lo:hi], self.v[lo:hi], lr, 0.9, 0.99, 1e-15, self.wd,
                          1.0 / self.world_size, self.opt_steps[group])
            dist.all_gather_into_tensor(self.flat.params[a:b], self.flat.params[lo:hi])
            return
        ops.adam_step(self.flat.params[a:b], self.flat.grads[a:b], self.m[a:b], self.v[a:b], lr, 0.9, 0.99, 1e-15, self.wd,
                      1.0 / self.world_size, self.opt_steps[group])

    def _forward_backward(self, data: Dict[str, Tensor], prop_grad: bool) -> Tensor:
        """zero grads -> render -> losses -> backward (everything of a step that is the same from step to step)."""
        self.flat.zero_grad()
        with fused.grad_sinks(True):
            results = render_rays(radiance_field=self.model, proposal_estimator=self.estimator, proposal_networks=self.props,
                                  data_dict=data, cfg=self.rcfg, proposal_requires_grad=prop_grad)
            if prop_grad:
                prop_loss = self.estimator.compute_loss(results["extras"]["trans"], loss_scaler=self.loss_scale)
                prop_loss.backward(gradient=self._seed(prop_loss))
                self._launch_prop_bucket()
            target, loss = self._scaled_losses(results, data)
            target.backward(gradient=self._seed(target))
        fused.join_side_stream()  # weight gradients written on the side stream are complete from here on
        return loss

    def _graphed_forward_backward(self, data: Dict[str, Tensor], prop_grad: bool) -> Tensor:
        """hipGraph replay of _forward_backward: one graph per step type (with / without proposal-net training),
        captured on first use after a short side-stream warm-up; inputs are copied into static buffers.  The stratified
        jitter still differs from replay to replay (graph-safe Philox offsets).  All launches go through the stream
        torch hands out, so the ctypes kernel launches are captured like torch's own; nothing in the step synchronises."""
        if self._static_data is None:
            self._static_data = {k: v.clone() for k, v in data.items()}
        elif data is not self._static_data:
            pairs = [(self._static_data[k], v) for k, v in data.items() if v.data_ptr() != self._static_data[k].data_ptr()]
            # one multi-tensor launch per dtype instead of one device-to-device copy per tensor (a list of mixed dtypes makes
            # _foreach_copy_ fall back to per-tensor copies: six 5-us launches per step in the trace)
            by_dtype: Dict[object, list] = {}
            for d, s_ in pairs:
                by_dtype.setdefault((d.dtype, s_.dtype, s_.is_contiguous() and d.is_contiguous()), []).append((d, s_))
            for grp in by_dtype.values():
                torch._foreach_copy_([d for d, _ in grp], [s_ for _, s_ in grp])
        sd = self._static_data
        if prop_grad not in self._graphs:
            self._hold_buckets = True  # the warm-up passes and the capture must not start gradient collectives
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):  # allocator and lazy-init warm-up outside the capture
                    self._forward_backward(sd, prop_grad)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            ops.DEFERRED_FINITE.clear()
            with torch.cuda.graph(g):
                out = self._forward_backward(sd, prop_grad)
            self._graphs[prop_grad] = (g, out, list(ops.DEFERRED_FINITE))   # EMER_CHECK_FINITE=1: device-side verdicts of this graph
            ops.DEFERRED_FINITE.clear()
            self._hold_buckets = False
        g, out, finite_checks = self._graphs[prop_grad]
        g.replay()
        if finite_checks:
            ops.check_deferred_finite(finite_checks)   # one host read per replay, debug mode only
        return out

    def train_step(self, data: Dict[str, Tensor]) -> Dict[str, float]:
        step = self.step_count
        prop_grad = self.requires_grad_fn(step)
        if self.use_graph:
            first = prop_grad not in self._graphs   # the capture of this step type happens in this call (same schedule on every rank)
            failed, err = False, None
            fatal = None
            try:
                loss = self._graphed_forward_backward(data, prop_grad)
            except FloatingPointError as e:  # EMER_CHECK_FINITE=1: a replayed step saw a non-finite gradient -- not a capture problem
                fatal = e
            except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly, once
                if self._dp_on and not first:
                    raise   # a replay that fails later cannot be agreed on (the peers are not at a collective)
                failed, err = True, e
            # A rank that switched to eager launches on its own would issue the early / table / xyzt buckets while its peers, still
            # replaying graphs, issue one all-reduce over the whole range: mismatched collectives hang or corrupt the gradients.  So the
            # ranks agree right after the capture attempt: if it failed anywhere, everybody falls back.  The vote is taken on EVERY way
            # out of a first-capture call -- also when this rank is about to raise (its peers sit in the vote's all-reduce and would
            # wait for it forever otherwise); the error is re-raised after the vote.
            if self._dp_on and first:
                failed_here, failed = failed, agree_any(failed or fatal is not None, self.device)
                if failed and not failed_here and fatal is None:
                    err = RuntimeError("a peer rank's hipGraph capture failed")
            if fatal is not None:
                raise fatal
            if failed:
                import warnings
                warnings.warn(f"hipGraph capture failed ({err!r}); continuing with eager launches")
                self.use_graph = False
                self._hold_buckets = False
                self._graphs.clear()
                loss = self._forward_backward(data, prop_grad)
        else:
            loss = self._forward_backward(data, prop_grad)
        self._exchange_grads(prop_grad)
        lr = self.lr * lr_factor(self.sched_ticks, self.num_iters)
        if prop_grad:
            self._adam("prop", lr)
        self._adam("main", lr)
        self.step_count += 1
        self.sched_ticks += 1
        return {"loss": loss, "prop_grad": prop_grad}
