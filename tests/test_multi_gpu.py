"""Data-parallel equivalence of the Trainer on ONE GPU shared by two processes (SURVEY.md section 8e "Parity for W>1":
the reference has no multi-GPU path, so W ranks on rays split W ways must equal one rank on the concatenated batch).

Two spawned processes, each a ``Trainer(world_size=2)`` on its half of the rays, exchange gradients through
``torch.distributed`` (backend gloo on device tensors -- the same ``dist.all_reduce`` call site RCCL serves on a multi-GPU
node) and take one optimizer step; the parent runs the same step on all rays in one process and compares parameters.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

R_HALF, S, SEED = 512, 32, 21


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(world_size, kind="static", dp_mode="allreduce"):
    from emernerf_amd.trainer import Trainer
    tr = Trainer(kind=kind, device="cuda:0", num_samples=S, prop_samples=(32, 16), table_init=0.3, seed=SEED, world_size=world_size,
                 dp_mode=dp_mode)
    tr.requires_grad_fn(0)  # (the schedule never fires on its very first call)
    tr.step_count = 7       # ... and on every early step after it: this step also trains the proposal nets
    return tr


def _data(lo, hi, mode="pixel"):
    from emernerf_amd.trainer import synthetic_lidar_rays, synthetic_rays
    full = synthetic_lidar_rays(2 * R_HALF, "cuda:0", seed=5) if mode == "lidar" else synthetic_rays(2 * R_HALF, "cuda:0", seed=5)
    jit = [torch.rand(2 * R_HALF, generator=torch.Generator().manual_seed(100 + i)).to("cuda:0") for i in range(3)]
    noise = torch.rand(2 * R_HALF, S, 1, generator=torch.Generator().manual_seed(99)).to("cuda:0")
    return {k: v[lo:hi].contiguous() for k, v in full.items()}, [j[lo:hi].contiguous() for j in jit], noise[lo:hi].contiguous()


def _step(tr, data, jit, noise, mode):
    it = iter(jit)
    tr.estimator.jitter_fn = lambda n, d: next(it)
    tr.model._noise = lambda like: noise   # temporal-aggregation noise of the flow model, replayed per ray
    return tr.lidar_step(data) if mode == "lidar" else tr.train_step(data)


def _worker(rank, port, out_dir, kind, mode, dp_mode, debug):
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if debug:
        os.environ["EMER_DP_DEBUG"] = "1"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    tr = _make(2, kind, dp_mode)
    data, jit, noise = _data(rank * R_HALF, (rank + 1) * R_HALF, mode)
    calls = {"prop": 0, "early": 0}
    orig_prop, orig_early = tr._launch_prop_bucket, tr._launch_early_bucket

    def spy_prop():
        calls["prop"] += 1
        orig_prop()
        if dp_mode == "allreduce":
            assert tr._prop_work is not None, "the proposal bucket must be in flight before the main backward"

    def spy_early():
        calls["early"] += 1
        orig_early()
    tr._launch_prop_bucket = spy_prop
    tab = tr.model.xyz_encoder.tcnn_encoding.params
    tab._emer_before_table_grad = spy_early
    split = getattr(tab, "_emer_table_split", None)
    calls["table"] = 0
    if dp_mode == "allreduce":
        assert split is not None, "all-reduce mode: the static table's backward must be split by level range"
        k, orig_table = split

        def spy_table(param, lo, hi):
            calls["table"] += 1
            assert param is tab and 0 < lo < hi == tab.numel()
            n_before = len(tr._table_work)
            orig_table(param, lo, hi)
            if not debug:
                assert len(tr._table_work) == n_before + 1, "the first level range's all-reduce must be in flight before the second launch"
        tab._emer_table_split = (k, spy_table)
    else:
        assert split is None
    calls["xyzt"] = 0
    for p, _ in tr.flat._table_offsets:   # the dynamic / flow tables: their collectives start from their own last backward
        orig_after = getattr(p, "_emer_after_table_grad", None)
        if orig_after is not None:
            def spy_after(param, _orig=orig_after):
                calls["xyzt"] += 1
                n_before = len(tr._table_work) + len(tr._table_snapshots)
                _orig(param)
                assert len(tr._table_work) + len(tr._table_snapshots) == n_before + 1
            p._emer_after_table_grad = spy_after
    calls["all_reduce"] = 0
    orig_all_reduce = dist.all_reduce

    def counting_all_reduce(*a, **k):
        calls["all_reduce"] += 1
        return orig_all_reduce(*a, **k)
    dist.all_reduce = counting_all_reduce
    out = _step(tr, data, jit, noise, mode)
    dist.all_reduce = orig_all_reduce
    if dp_mode == "single":   # BASELINE.json north_star: "a single RCCL all-reduce of grads ... per step" (EMER_DP_SINGLE=1)
        assert calls["all_reduce"] == 1, f"single-collective mode issued {calls['all_reduce']} all-reduces"
    assert calls["xyzt"] == (2 if kind == "flow" and dp_mode == "allreduce" else 0), f"xyzt table buckets: {calls}"
    assert out["prop_grad"], "the test step must exercise the proposal-net range of the exchange"
    assert calls["prop"] == 1 and calls["early"] == 1, f"buckets not launched exactly once: {calls}"
    assert calls["table"] == (1 if dp_mode == "allreduce" else 0), f"table level-range bucket: {calls}"
    torch.cuda.synchronize()
    torch.save({"params": tr.flat.params.cpu(), "ranges": dict(tr.flat.ranges)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,mode,dp_mode,debug", [("static", "pixel", "allreduce", False), ("static", "pixel", "allreduce", True),
                                                     ("flow", "pixel", "allreduce", True), ("flow", "pixel", "allreduce", False),
                                                     ("static", "lidar", "allreduce", True),
                                                     ("static", "pixel", "rs_ag", False), ("flow", "lidar", "rs_ag", False),
                                                     ("static", "pixel", "single", False), ("flow", "pixel", "single", False)])
def test_two_ranks_equal_one_rank_on_concatenated_rays(hip_lib, tmp_path, kind, mode, dp_mode, debug):
    """static and flow models, the pixel step and the lidar step, the bucketed all-reduce, the reduce-scatter -> sharded
    Adam -> all-gather exchange and the single all-reduce after the backward (dp_mode="single" / EMER_DP_SINGLE=1: exactly one collective).  debug=True (EMER_DP_DEBUG=1): the trainer asserts that no gradient of the early bucket's
    ranges is written after the point where the bucket is launched (the ordering assumption behind hiding it)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path), kind, mode, dp_mode, debug), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    p0, p1 = r0["params"], r1["params"]
    assert torch.equal(p0, p1), "replicas diverged after one step"
    tr = _make(1, kind)
    before = tr.flat.params.cpu().clone()
    data, jit, noise = _data(0, 2 * R_HALF, mode)
    _step(tr, data, jit, noise, mode)
    torch.cuda.synchronize()
    single = tr.flat.params.cpu()
    n = single.numel()   # (rs_ag pads the groups of the 2-rank buffer: compare group by group)
    if p0.numel() != n:
        p0 = torch.cat([p0[a:a + (d - c)] for (a, _), (c, d) in zip(r0["ranges"].values(), tr.flat.ranges.values())])
    # Adam's first steps move every touched parameter by ~lr: compare the UPDATES (fp32 reduction order differs)
    du, dv = p0 - before, single - before
    touched = dv.abs() > 0
    assert int(touched.sum()) > 100000
    # the update is lr * m / (sqrt(v) + eps): sign and size ~lr for any gradient, so entries whose tiny gradient flips sign
    # between the two summation orders are the only possible mismatches
    bad = (du - dv).abs() > 1e-3 * dv.abs().clamp_min(1e-12)
    assert int(bad.sum()) <= max(10, int(touched.sum()) // 2000), f"{int(bad.sum())} of {int(touched.sum())} updates differ"


def _capture_failure_worker(rank, port, out_dir):
    """Both ranks ask for hipGraph replay; rank 1's capture attempt fails.  The ranks must agree (Trainer.train_step -> agree_any) and BOTH
    continue with eager launches -- otherwise rank 0 would replay (one all-reduce after the graph) while rank 1 launches buckets."""
    import warnings
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    tr = _make(2, "static", "allreduce")
    tr.use_graph = True
    if rank == 1:
        def broken(data, prop_grad):
            raise RuntimeError("injected capture failure")
        tr._graphed_forward_backward = broken
    data, jit, noise = _data(rank * R_HALF, (rank + 1) * R_HALF, "pixel")
    import itertools
    draws = itertools.cycle(jit)   # (the capturing rank runs the forward several times -- warm-up, capture, replay, fallback: three draws each)
    tr.estimator.jitter_fn = lambda n, d: next(draws)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.train_step(data)
    assert tr.use_graph is False, f"rank {rank} kept replaying after a peer's capture failed"
    assert any("capture failed" in str(x.message) for x in w), [str(x.message) for x in w]
    torch.cuda.synchronize()
    torch.save({"params": tr.flat.params.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_capture_failure_on_one_rank_falls_back_everywhere(hip_lib, tmp_path):
    """the agreed eager fallback end to end -- a capture failure injected on ONE of two ranks; both fall back, the replicas stay
    identical and the step equals the one-rank step on the concatenated rays (as the all-eager two-rank run does)."""
    import torch.multiprocessing as mp
    mp.spawn(_capture_failure_worker, args=(_free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "rank0.pt")["params"], torch.load(tmp_path / "rank1.pt")["params"]
    assert torch.equal(p0, p1), "replicas diverged after the agreed fallback"
    tr = _make(1, "static")
    before = tr.flat.params.cpu().clone()
    data, jit, noise = _data(0, 2 * R_HALF, "pixel")
    _step(tr, data, jit, noise, "pixel")
    torch.cuda.synchronize()
    du, dv = p0 - before, tr.flat.params.cpu() - before
    touched = dv.abs() > 0
    bad = (du - dv).abs() > 1e-3 * dv.abs().clamp_min(1e-12)
    assert int(touched.sum()) > 100000 and int(bad.sum()) <= max(10, int(touched.sum()) // 2000), f"{int(bad.sum())} of {int(touched.sum())} updates differ"


def _rccl_worker(rank, port, out_dir, dp_mode):
    """One rank, backend "nccl" (= RCCL on ROCm), EMER_DP_FORCE=1: the trainer takes its data-parallel path and the REAL collectives
    run -- async all_reduce buckets, or the in-place reduce_scatter_tensor (output shard aliasing its input) and
    all_gather_into_tensor of the rs_ag mode -- trivially, on a communicator of size one."""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["EMER_DP_FORCE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl"
    tr = _make(1, "static", dp_mode)
    assert tr._dp_on
    seen = []
    for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: lambda *a, **k: (seen.append(n), o(*a, **k))[1])(orig, name))
    data, jit, noise = _data(0, 2 * R_HALF, "pixel")
    for _ in range(2):
        _step(tr, data, jit, noise, "pixel")
    torch.cuda.synchronize()
    want = {"allreduce": {"all_reduce"}, "single": {"all_reduce"}, "rs_ag": {"reduce_scatter_tensor", "all_gather_into_tensor"}}[dp_mode]
    assert set(seen) == want, seen
    if dp_mode == "single":
        assert len(seen) == 2, f"one all-reduce per step: {seen}"
    torch.save({"params": tr.flat.params.cpu(), "ranges": dict(tr.flat.ranges)}, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("dp_mode", ["allreduce", "rs_ag", "single"])
def test_real_rccl_collectives_execute_on_one_rank(hip_lib, tmp_path, dp_mode):
    """``reduce_scatter_tensor`` had never executed on any backend.  On this one-GPU box RCCL can only form a communicator
    of one rank, so the exchange is numerically the identity -- which makes the check exact: two steps through the real RCCL calls
    equal two steps of the plain single-GPU trainer (group padding of rs_ag aside), update for update."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path), dp_mode), nprocs=1, join=True)
    got = torch.load(tmp_path / "rccl.pt")
    tr = _make(1, "static")
    before = tr.flat.params.cpu().clone()
    data, jit, noise = _data(0, 2 * R_HALF, "pixel")
    for _ in range(2):
        _step(tr, data, jit, noise, "pixel")
    torch.cuda.synchronize()
    p = got["params"]
    if p.numel() != tr.flat.numel:
        p = torch.cat([p[a:a + (d - c)] for (a, _), (c, d) in zip(got["ranges"].values(), tr.flat.ranges.values())])
    # (the owner-computes grid backward is reproducible to an ulp, not bitwise: compare the UPDATES as the two-rank test does)
    du, dv = p - before, tr.flat.params.cpu() - before
    touched = dv.abs() > 0
    assert int(touched.sum()) > 100000
    bad = (du - dv).abs() > 1e-3 * dv.abs().clamp_min(1e-12)
    assert int(bad.sum()) <= max(10, int(touched.sum()) // 2000), f"{int(bad.sum())} of {int(touched.sum())} updates differ"


def test_bench_runs_with_two_ranks(hip_lib, tmp_path):
    """bench.py's N > 1 path end to end on this one-GPU box: two ranks share cuda:0 and exchange through gloo (test hooks;
    on a multi-GPU node the same code runs over RCCL).  Guards the contract that EVERY rank takes every step that contains
    a collective -- a rank-0-only training step after the timed region once hung the whole job."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMER_BENCH_SHARE_GPU="1", EMER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--init-steps", "7", "--rays", "1024", "--samples", "32"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["config"]["global_rays"] == 2048 and out["value"] > 0 and out["scaling"] == "weak"
