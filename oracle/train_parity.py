"""K-step training parity harness -- test infrastructure, not product code.

Trains the HIP ``Trainer`` and the CPU oracle (``oracle/ref_path.RefPath`` on the C oracle) from IDENTICAL parameters for K
optimizer steps on rays whose colours were rendered from a fixed synthetic "ground-truth" field, with the stratified jitter
replayed on both sides, and reports what the second half of BASELINE.json's metric asks for ("PSNR ... of
both vs a synthetic GT scene after the same K training steps", datasets/metrics.py:31-46, train_emernerf.py:634-745): per-step
losses of both, the parameters after K steps, and the PSNR of an evaluation render of each against the ground truth.

The oracle side is driven by the reference's own optimizer construction (builders.py:50-89,114-142: Adam(eps=1e-15,
weight_decay, betas=(0.9, 0.99)) + ChainedScheduler(LinearLR(0.01, num_iters // 10), MultiStepLR(gamma 0.33))) -- torch's classes,
not the product's lr_factor / emer_adam_step -- and by the never-unscaled GradScaler(2**10) quirk (gradients enter Adam x1024).

Only tests/ and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import oracle as O
from .ref_path import RefPath


def ref_from_trainer(tr) -> RefPath:
    """A RefPath holding copies of the trainer's CURRENT parameters (reference state_dict names)."""
    from emernerf_amd.trainer import AABB, PROP_KW
    c = tr.cfg
    x = c.xyz_encoder
    grids = {"model/xyz_encoder": O.grid_meta_from_encoder_args(3, x.n_levels, x.base_resolution, x.max_resolution, x.log2_hashmap_size,
                                                                 x.n_features_per_level)}
    if tr.model.dynamic_xyz_encoder is not None:
        d = c.dynamic_xyz_encoder
        grids["model/dynamic_xyz_encoder"] = O.grid_meta_from_encoder_args(4, d.n_levels, d.base_resolution, d.max_resolution,
                                                                           d.log2_hashmap_size, d.n_features_per_level)
    if tr.model.flow_xyz_encoder is not None:
        grids["model/flow_xyz_encoder"] = O.grid_meta_from_encoder_args(4, 10, 16, 4096, 18, 4)  # radiance_field.py:916-923
    for i, kw in enumerate(PROP_KW):
        grids[f"prop{i}/xyz_encoder"] = O.grid_meta_from_encoder_args(3, kw["n_levels"], 16, kw["max_resolution"], kw["log2_hashmap_size"],
                                                                      kw["n_features_per_level"])
    ms = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
    ps = [{k: v.detach().cpu() for k, v in p.state_dict().items()} for p in tr.props]
    return RefPath(ms, ps, grids, AABB, time_diff=1 / c.num_train_timesteps)


def reference_optimizers(ref: RefPath, lr: float, weight_decay: float, num_iters: int):
    """builders.py:50-89 (main) and :114-142 (proposal nets: ONE optimizer over every proposal net's parameters)."""
    def build(params):
        opt = torch.optim.Adam(params, lr=lr, eps=1e-15, weight_decay=weight_decay, betas=(0.9, 0.99))
        milestones = [num_iters // 2, num_iters * 3 // 4, num_iters * 9 // 10]
        if num_iters >= 10000:
            milestones.insert(0, num_iters // 4)
        sched = torch.optim.lr_scheduler.ChainedScheduler([
            torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=num_iters // 10),
            torch.optim.lr_scheduler.MultiStepLR(opt, milestones=milestones, gamma=0.33)])
        return opt, sched
    om, sm = build(ref.trainable("model/"))
    op, sp = build(ref.trainable("prop"))
    return om, sm, op, sp


def psnr(prediction: Tensor, target: Tensor) -> float:
    """datasets/metrics.py:31-46."""
    mse = float(torch.nn.functional.mse_loss(prediction.double(), target.double()))
    return float("inf") if mse == 0.0 else -10.0 * math.log10(mse)


def gt_batches(kind: str, device, n_batches: int, rays: int, samples: int, prop_samples, seed: int = 100) -> List[Dict[str, Tensor]]:
    """Ray batches whose ``pixels`` (and ``sky_masks`` = rendered opacity < 0.5) come from a fixed synthetic ground-truth field: a
    second model of the same kind with its own random tables (+-0.5), rendered once by the HIP path in eval mode."""
    from emernerf_amd.render_utils import render_rays
    from emernerf_amd.trainer import Trainer, synthetic_rays
    teacher = Trainer(kind=kind, device=device, num_samples=samples, prop_samples=prop_samples, table_init=0.5, seed=seed)
    feat = kind == "feature"   # configs[4]: three cameras, 64-d feature targets (default_config.yaml:19,93-94)
    mods = [teacher.model, teacher.estimator] + list(teacher.props)
    for m in mods:
        m.eval()
    out = []
    with torch.no_grad():
        for b in range(n_batches):
            data = synthetic_rays(rays, device, seed=seed + 1 + b, num_cams=3 if feat else 1, feature_dim=64 if feat else 0)
            res = render_rays(radiance_field=teacher.model, proposal_estimator=teacher.estimator, proposal_networks=teacher.props,
                              data_dict=data, cfg=teacher.rcfg, proposal_requires_grad=False)
            data["pixels"] = res["rgb"].clamp(0, 1).contiguous()
            data["sky_masks"] = (res["opacity"].squeeze(-1) < 0.5).float()
            if feat:   # the feature target is the teacher's rendered feature map (with its learnable PE), as the pixels are its colours
                data["features"] = res["dino_feat"].contiguous()
            out.append(data)
    return out


def cotrain(kind: str, device, K: int, rays: int, samples: int, prop_samples=(64, 32), num_iters: int = 200, table_init: Optional[float] = 0.3,
            use_graph: bool = False, seed: int = 11, n_batches: int = 4, eval_rays: int = 2048, schedule_steps: int = 10,
            run_oracle: bool = True, time_oracle_from: int = 0) -> Dict[str, object]:
    """Train HIP and oracle for K steps from identical parameters; see the module docstring.  ``schedule_steps``: the proposal
    schedule's ramp (nerfacc_prop_net.py:280-296 with num_steps = schedule_steps, so that K steps see both step types).
    Returns a dict of per-step losses, parameter statistics, PSNRs vs the ground truth and the oracle's seconds per step."""
    from emernerf_amd.prop_net import get_proposal_requires_grad_fn
    from emernerf_amd.render_utils import render_rays
    from emernerf_amd.trainer import Trainer
    batches = gt_batches(kind, device, n_batches + 1, max(rays, eval_rays), samples, prop_samples)
    eval_batch = {k: v[:eval_rays].contiguous() for k, v in batches[-1].items()}
    train = [{k: v[:rays].contiguous() for k, v in b.items()} for b in batches[:-1]]
    tr = Trainer(kind=kind, device=device, num_samples=samples, prop_samples=prop_samples, table_init=table_init, seed=seed, num_iters=num_iters,
                 use_graph=use_graph)
    tr.requires_grad_fn = get_proposal_requires_grad_fn(5.0, schedule_steps)
    ref = ref_from_trainer(tr) if run_oracle else None
    p_init = tr.flat.params.clone()
    # replayed randomness: one U(0,1) per ray and resampling round (and, for the flow models, the temporal-aggregation noise);
    # device buffers with fixed addresses so that a captured step graph reads the new draws
    g = torch.Generator().manual_seed(seed + 1)
    n_rounds = len(prop_samples) + 1
    jit_all = [[torch.rand(rays, generator=g) for _ in range(n_rounds)] for _ in range(K)]
    noise_all = [torch.rand(rays, samples, 1, generator=g) for _ in range(K)] if kind in ("flow", "feature") else None
    jit_buf = [torch.empty(rays, device=device) for _ in range(n_rounds)]
    calls = [0]

    def jitter_fn(n, d):
        calls[0] += 1
        return jit_buf[(calls[0] - 1) % n_rounds]
    tr.estimator.jitter_fn = jitter_fn
    if noise_all is not None:
        noise_buf = torch.empty(rays, samples, 1, device=device)
        tr.model._noise = lambda like: noise_buf
    flags, hip_losses = [], []
    for k in range(K):
        for b, j in zip(jit_buf, jit_all[k]):
            b.copy_(j)
        if noise_all is not None:
            noise_buf.copy_(noise_all[k])
        calls[0] = 0
        out = tr.train_step(train[k % len(train)])
        flags.append(bool(out["prop_grad"]))
        hip_losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    res: Dict[str, object] = {"kind": kind, "K": K, "rays": rays, "samples": samples, "hip_losses": hip_losses, "prop_flags": flags,
                              "launch_mode": "hipgraph" if tr.use_graph else "eager"}

    def hip_eval(trainer) -> Dict[str, Tensor]:
        mods = [trainer.model, trainer.estimator] + list(trainer.props)
        for m in mods:
            m.eval()
        with torch.no_grad():
            o = render_rays(radiance_field=trainer.model, proposal_estimator=trainer.estimator, proposal_networks=trainer.props,
                            data_dict=eval_batch, cfg=trainer.rcfg, proposal_requires_grad=False)
        for m in mods:
            m.train()
        return o
    tr.estimator.jitter_fn = None
    if noise_all is not None:
        del tr.model._noise   # back to the class's own draw (ones outside training, radiance_field.py:567-568)
    res["hip_psnr_vs_gt_db"] = psnr(hip_eval(tr)["rgb"].cpu(), eval_batch["pixels"].cpu())
    res["travel"] = float((tr.flat.params - p_init).norm())
    if not run_oracle:
        return res
    # ------------------------------------------------------------------------------------------------ the oracle's K steps
    om, sm, op, sp = reference_optimizers(ref, tr.lr, tr.wd, num_iters)
    cpu_train = [{k: v.cpu() for k, v in b.items()} for b in train]
    ref_losses, t_timed = [], 0.0
    for k in range(K):
        t0 = time.perf_counter()
        nf = (lambda like, kk=k: noise_all[kk]) if noise_all is not None else None
        loss, _ = ref.train_step(cpu_train[k % len(train)], om, op, samples, list(prop_samples), jitters=jit_all[k], loss_scale=tr.loss_scale,
                                 prop_grad=flags[k], noise_fn=nf)
        sm.step(); sp.step()   # train_emernerf.py:745 and nerfacc_prop_net.py:240-277: both schedules tick every iteration
        ref_losses.append(loss)
        if k >= time_oracle_from:
            t_timed += time.perf_counter() - t0
    res["ref_losses"] = ref_losses
    res["oracle_s_per_step"] = t_timed / max(K - time_oracle_from, 1)
    cpu_eval = {k: v.cpu() for k, v in eval_batch.items()}
    with torch.no_grad():
        ro = ref.render_rays(cpu_eval, samples, list(prop_samples), jitters=None, training=False)
    res["ref_psnr_vs_gt_db"] = psnr(ro["rgb"], cpu_eval["pixels"])
    res["loss_max_rel_diff"] = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(hip_losses, ref_losses))
    # parameters after K steps, by reference name: error relative to the distance the parameter travelled
    # A hash-table entry whose gradient is rounding-sized (tcnn's +-1e-4 initialisation: most of them) can take Adam's first,
    # sign-sized steps in OPPOSITE directions on the two sides -- it then ends >= one smallest step (lr * warm-up start factor) away
    # although nothing is wrong.  Such entries are COUNTED (``n_sign_flipped``) and left out of ``l2_diff_excl``; every other entry
    # and every MLP / embedding parameter is held to the unwidened bounds by the test.
    from emernerf_amd.trainer import lr_factor
    flip_thr = 0.5 * tr.lr * lr_factor(0, num_iters)
    stats = {}
    for prefix, mod in [("model/", tr.model)] + [(f"prop{i}/", p) for i, p in enumerate(tr.props)]:
        for name, q in mod.named_parameters():
            want = ref.t[prefix + name].detach()
            got = q.detach().cpu()
            diff = got - want
            st = {"max_abs_diff": float(diff.abs().max()), "l2_diff": float(diff.norm()), "numel": got.numel()}
            if name.endswith("tcnn_encoding.params"):
                far = diff.abs() > flip_thr
                st["n_sign_flipped"] = int(far.sum())
                st["l2_diff_excl"] = float(diff[~far].norm())
            else:
                st["n_sign_flipped"], st["l2_diff_excl"] = 0, st["l2_diff"]
            stats[prefix + name] = st
    res["param_stats"] = stats
    res["flip_threshold"] = flip_thr
    res["param_l2_diff"] = math.sqrt(sum(v["l2_diff"] ** 2 for v in stats.values()))
    res["param_l2_diff_excl"] = math.sqrt(sum(v["l2_diff_excl"] ** 2 for v in stats.values()))
    res["n_sign_flipped"] = sum(v["n_sign_flipped"] for v in stats.values())
    res["n_table_entries"] = sum(v["numel"] for k, v in stats.items() if k.endswith("tcnn_encoding.params"))
    res["param_max_abs_diff"] = max(v["max_abs_diff"] for v in stats.values())
    return res
