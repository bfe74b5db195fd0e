"""CPU tests (no GPU): the C-ABI library loads and exports exactly what include/emernerf_hip.h declares,
host-side logic (level tables, schedules, flat buffers, proposal-loss restatement) and the N>1 data-parallel
plumbing over gloo (world_size 2).  No kernel is launched here."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------- C ABI
def _declared_functions():
    src = open(os.path.join(ROOT, "include", "emernerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(emer_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol(hip_lib):
    from emernerf_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/emernerf_hip.h but not exported"
    bound = set(_lib.SIGNATURES) | {"emer_last_error", "emer_version"} | set(_lib.INT64_FUNCTIONS)
    assert bound <= declared, f"bound but undeclared: {bound - declared}"
    assert declared <= bound, f"declared but not bound by the host layer: {declared - bound}"


def test_error_reporting_without_gpu(hip_lib):
    from emernerf_amd import _lib
    with pytest.raises(_lib.EmerError, match="n_dims"):
        _lib.make_grid_desc(7, 4, 2, 19, 16, 1.5)
    with pytest.raises(_lib.EmerError, match="n_features"):
        _lib.make_grid_desc(3, 4, 3, 19, 16, 1.5)
    assert hip_lib.emer_version() >= 1


def test_product_rejects_cpu_tensors(hip_lib):
    """No CPU fallback anywhere on the product path."""
    from emernerf_amd import _lib, ops
    from emernerf_amd.encodings import HashEncoder
    enc = HashEncoder(3, 4, 16, 64, 12, 2, verbose=False)
    with pytest.raises(_lib.EmerError):
        enc(torch.rand(8, 3))
    with pytest.raises(_lib.EmerError):
        ops.render_weights(torch.rand(2, 4), torch.rand(2, 4), torch.rand(2, 4))


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from emernerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EmerError, match="no CPU/PyTorch fallback"):
        _lib.load()


@pytest.mark.parametrize("args", [(3, 16, 16, 2048, 19, 2), (3, 10, 16, 8192, 20, 4), (4, 10, 32, 8192, 18, 4),
                                  (4, 10, 16, 4096, 18, 4), (3, 8, 16, 512, 20, 1), (3, 8, 16, 2048, 20, 1), (2, 5, 8, 256, 12, 2)])
def test_level_table_host_equals_oracle(hip_lib, oracle, args):
    """emer_grid_desc_init (product, host C++) and orc_grid_init (oracle, C) agree bit for bit."""
    from emernerf_amd import _lib
    D, L, base, mx, T, F = args
    meta = oracle.grid_meta_from_encoder_args(*args)
    desc = _lib.make_grid_desc(D, L, F, T, base, meta.per_level_scale)
    assert desc.n_entries == meta.n_entries
    assert np.array_equal(np.array(desc.scale[:L], np.float32).view(np.uint32), meta.scale.view(np.uint32))
    for fld in ("res", "size", "offset", "hashed"):
        assert np.array_equal(np.array(getattr(desc, fld)[:L], np.uint32), getattr(meta, fld))


def test_encoder_state_dict_layout(hip_lib):
    from emernerf_amd.encodings import HashEncoder
    enc = HashEncoder(verbose=False)  # defaults = BASELINE configs[1] grid (encodings.py:110-118)
    assert list(enc.state_dict()) == ["tcnn_encoding.params"]
    assert enc.tcnn_encoding.params.shape == (12196240,) and enc.n_output_dims == 32
    assert float(enc.tcnn_encoding.params.abs().max()) <= 1e-4  # tcnn init range


# ------------------------------------------------------------------------------- host logic
def test_proposal_requires_grad_schedule():
    from emernerf_amd.prop_net import get_proposal_requires_grad_fn
    fn = get_proposal_requires_grad_fn()
    seq = [fn(s) for s in range(3000)]
    assert seq[0] is False and seq[1] is True
    tail = seq[2000:]
    assert abs(sum(tail) / len(tail) - 1 / 6) < 0.01  # steady state: every 6th call (target 5 steps between updates)


def test_lr_factor_matches_torch_schedulers():
    from emernerf_amd.trainer import lr_factor
    for num_iters in (25000, 2000):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1.0)
        ms = [num_iters // 2, num_iters * 3 // 4, num_iters * 9 // 10]
        if num_iters >= 10000:
            ms.insert(0, num_iters // 4)
        sched = torch.optim.lr_scheduler.ChainedScheduler([  # builders.py:75-88
            torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=num_iters // 10),
            torch.optim.lr_scheduler.MultiStepLR(opt, milestones=ms, gamma=0.33)])
        for step in range(num_iters):
            if step % 997 == 0 or step in ms or step == num_iters // 10:
                assert abs(opt.param_groups[0]["lr"] - lr_factor(step, num_iters)) < 1e-9, step
            opt.step(); sched.step()


def test_proposal_grad_schedule_matches_reference_rule():
    """get_proposal_requires_grad_fn (nerfacc_prop_net.py:280-296): the closure's rule, restated literally here."""
    from emernerf_amd.prop_net import get_proposal_requires_grad_fn
    for target, num_steps in ((5.0, 1000), (2.0, 10), (0.0, 5)):
        fn = get_proposal_requires_grad_fn(target, num_steps)
        since = 0
        for step in range(3000):
            want = since > min(step / num_steps, 1.0) * target
            if want:
                since = 0
            since += 1
            assert fn(step) == want, (target, num_steps, step)


def test_flat_params_views_and_grad_accumulation():
    from emernerf_amd.trainer import FlatParams
    torch.manual_seed(0)
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)
    ref = [p.detach().clone() for p in list(a.parameters()) + list(b.parameters())]
    flat = FlatParams({"main": [a], "prop": [b]}, "cpu")
    assert flat.numel == sum(p.numel() for p in ref) and flat.ranges == {"main": (0, 18), "prop": (18, 26)}
    for p, r in zip(list(a.parameters()) + list(b.parameters()), ref):
        assert torch.equal(p.detach(), r)
    y = b(a(torch.ones(4, 5))).sum()
    y.backward()
    assert flat.grads.abs().sum() > 0 and a.weight.grad.data_ptr() == flat.grads.data_ptr()  # grads land in the flat buffer
    flat.params.mul_(2.0)
    assert torch.allclose(a.weight.detach(), ref[0] * 2)  # parameters are views of the flat buffer
    flat.zero_grad()
    assert float(flat.grads.abs().sum()) == 0.0 and float(a.weight.grad.abs().sum()) == 0.0


# ------------------------------------------------------------------------------- data parallel (gloo, world 2)
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from emernerf_amd.trainer import FlatParams
    torch.manual_seed(0)  # identical parameters on every rank, as Trainer does
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    prop = torch.nn.Linear(6, 1)
    flat = FlatParams({"main": [net], "prop": [prop]}, "cpu")
    g = torch.Generator().manual_seed(100)
    xs, ys = torch.randn(world * 16, 6, generator=g), torch.randn(world * 16, 3, generator=g)
    x, y = xs[rank * 16:(rank + 1) * 16], ys[rank * 16:(rank + 1) * 16]  # each rank its own rays
    flat.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    (prop(x) ** 2).mean().backward()
    dist.all_reduce(flat.grads)  # THE one collective of a step: sum of the flat gradient buffer
    flat.grads.div_(world)       # (folded into the fused Adam's grad_scale on the GPU path)
    # single-process reference on the concatenated batch
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    prop2 = torch.nn.Linear(6, 1)
    ((net2(xs) - ys) ** 2).mean().backward()
    (prop2(xs) ** 2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for p in list(net2.parameters()) + list(prop2.parameters())])
    ret[rank] = float((flat.grads - want).abs().max())
    dist.destroy_process_group()


def _agree_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from emernerf_amd.trainer import agree_any
    ret[rank] = (agree_any(rank == 1, "cpu"), agree_any(False, "cpu"), agree_any(True, "cpu"))
    dist.destroy_process_group()


def test_ranks_agree_on_a_local_failure_gloo():
    """trainer.agree_any: a condition that is true on ONE rank (a failed hipGraph capture) becomes true on every rank, so all of them
    take the eager fallback together and issue the same sequence of gradient buckets; without a process group it is the local flag."""
    from emernerf_amd.trainer import agree_any
    assert agree_any(True, "cpu") is True and agree_any(False, "cpu") is False
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_agree_worker, args=(world, port, ret), nprocs=world, join=True)
        for r in range(world):
            assert tuple(ret[r]) == (True, False, True), dict(ret)


def test_data_parallel_flat_allreduce_gloo():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
        assert len(ret) == world and all(v < 1e-6 for v in ret.values()), dict(ret)


def test_subtract_ranges_partitions_the_span():
    """trainer._subtract_ranges: what is left of the exchange span after the ranges whose collectives started early -- together with those
    ranges it must cover every element of the span exactly once (random spans / holes, including holes that stick out or touch)."""
    sys.path.insert(0, ROOT)
    from emernerf_amd.trainer import _subtract_ranges
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = int(rng.integers(0, 50)); b = a + int(rng.integers(1, 200))
        cuts = sorted(set(int(v) for v in rng.integers(a - 20, b + 20, size=int(rng.integers(0, 9)))))
        holes = [(cuts[i], cuts[i + 1]) for i in range(0, len(cuts) - 1, 2)]   # disjoint, sorted, possibly outside [a, b)
        rng.shuffle(holes)
        late = _subtract_ranges((a, b), holes)
        count = np.zeros(b - a + 60, dtype=np.int64)
        for lo, hi in late:
            assert a <= lo < hi <= b
            count[lo - a + 30:hi - a + 30] += 1
        for lo, hi in holes:
            count[max(lo, a) - a + 30:max(min(hi, b), a) - a + 30] += 1
        assert (count[30:30 + b - a] == 1).all() and count[:30].sum() == 0 and count[30 + b - a:].sum() == 0
    assert _subtract_ranges((0, 10), []) == [(0, 10)] and _subtract_ranges((0, 10), [(0, 10)]) == []


class _FakeEncoding(torch.nn.Module):
    """A parameter named like a hash table (``...tcnn_encoding.params``): FlatParams files it under the tables."""

    def __init__(self, n):
        super().__init__()
        self.tcnn_encoding = torch.nn.Module()
        self.tcnn_encoding.params = torch.nn.Parameter(torch.zeros(n))


def _exchange_worker(rank, world, port, ret, dp_mode, early, table_cut, prop_grad):
    """The trainer's OWN exchange code (Trainer._exchange_grads / _launch_table_bucket / _launch_early_bucket / _launch_prop_bucket) on a
    CPU flat buffer over gloo: whatever subset of the buckets was started early, every gradient of the step ends up summed exactly once."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import types
    from emernerf_amd import trainer as T
    enc = _FakeEncoding(4096)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    xyzt = _FakeEncoding(1024)
    head = torch.nn.Linear(3, 2)
    prop = torch.nn.Sequential(_FakeEncoding(512), torch.nn.Linear(4, 1))
    flat = T.FlatParams({"main": [enc, mlp, xyzt, head], "prop": [prop]}, "cpu", align=4 * world)
    g = torch.Generator().manual_seed(7 + rank)
    flat.grads.copy_(torch.randn(flat.numel, generator=g))
    for p in flat._tables:
        p._emer_grad_fresh = False   # every table "was written" this step
    mine = flat.grads.clone()
    tr = types.SimpleNamespace(flat=flat, world_size=world, dp_mode=dp_mode, dp_debug=False, comm_events=None, _dp_on=True, _hold_buckets=False,
                               _rs_emulate=True, _early_done=False, _early_work=[], _prop_work=None, _table_work=[], _table_ranges=[],
                               _table_snapshots=[], model=types.SimpleNamespace(xyz_encoder=enc))
    a, b = flat.ranges["main"]
    tr._early_ranges = [(lo, hi) for lo, hi in flat._dense_ranges if a <= lo and hi <= b]
    for name in ("_exchange_grads", "_launch_table_bucket", "_launch_whole_table_bucket", "_launch_early_bucket", "_launch_prop_bucket"):
        setattr(tr, name, types.MethodType(getattr(T.Trainer, name), tr))
    orig_cap = torch.cuda.is_current_stream_capturing
    torch.cuda.is_current_stream_capturing = lambda: False   # (CPU build of the test: no stream to ask)
    T.fused.join_side_stream = lambda: None
    try:
        if prop_grad:
            tr._launch_prop_bucket()
        if early and dp_mode in ("allreduce", "single"):   # "single": the hooks fire as in an eager step and must start nothing
            tr._launch_whole_table_bucket(xyzt.tcnn_encoding.params)         # an xyzt table, from its own last backward
            tr._launch_early_bucket()                                        # the MLP ranges, when the static table's backward starts
            if table_cut:
                tr._launch_table_bucket(enc.tcnn_encoding.params, table_cut, 4096)   # the fine levels, between the two launches
        if dp_mode == "single":
            assert tr._early_work == [] and tr._table_work == [] and tr._prop_work is None, "single mode: nothing before the backward ends"
            n_coll, orig_ar = [0], dist.all_reduce
            dist.all_reduce = lambda *a_, **k_: (n_coll.__setitem__(0, n_coll[0] + 1), orig_ar(*a_, **k_))[1]
        tr._exchange_grads(prop_grad)
        if dp_mode == "single":
            dist.all_reduce = orig_ar
            assert n_coll[0] == 1, f"EMER_DP_SINGLE: {n_coll[0]} all-reduces in one step's exchange"
    finally:
        torch.cuda.is_current_stream_capturing = orig_cap
    # reference: the plain sum over ranks
    total = mine.clone()
    dist.all_reduce(total)
    pa, pb = flat.ranges["prop"]
    if dp_mode == "rs_ag":   # every rank holds the sum of ITS shard of each exchanged group
        err = 0.0
        for grp, (lo, hi) in tr._rs_shards.items():
            err = max(err, float((flat.grads[lo:hi] - total[lo:hi]).abs().max()))
        assert set(tr._rs_shards) == ({"main", "prop"} if prop_grad else {"main"})
    else:
        err = float((flat.grads[a:b] - total[a:b]).abs().max())
        if prop_grad:
            err = max(err, float((flat.grads[pa:pb] - total[pa:pb]).abs().max()))
        else:   # the idle proposal net's range is not exchanged
            assert torch.equal(flat.grads[pa:pb], mine[pa:pb])
    assert tr._early_work == [] and tr._table_work == [] and tr._table_ranges == [] and tr._prop_work is None
    ret[rank] = err
    dist.destroy_process_group()


@pytest.mark.parametrize("dp_mode,early,table_cut,prop_grad", [("allreduce", False, 0, False), ("allreduce", True, 0, False),
                                                                ("allreduce", True, 3000, False), ("allreduce", True, 3000, True),
                                                                ("rs_ag", False, 0, True), ("rs_ag", False, 0, False),
                                                                ("single", True, 3000, True), ("single", True, 0, False)])
def test_trainer_exchange_code_over_gloo(dp_mode, early, table_cut, prop_grad):
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_exchange_worker, args=(world, port, ret, dp_mode, early, table_cut, prop_grad), nprocs=world, join=True)
        assert len(ret) == world and all(v < 1e-6 for v in ret.values()), dict(ret)


@pytest.mark.parametrize("grid,rows", [((3, 16, 16, 2048, 19, 2), 64),     # BASELINE configs[1] main grid: 64 slices per hashed level
                                       ((3, 10, 16, 8192, 20, 4), 256),    # default static grid: 256 LDS slices -> 256 bitmap rows
                                       ((4, 10, 32, 8192, 18, 4), 64),     # dynamic / flow xyzt grids
                                       ((3, 8, 16, 2048, 20, 1), 64)])     # proposal net
def test_slice_plan_host_logic(hip_lib, grid, rows):
    """Host side of the owner-computes backward (no kernel launched): every shipped grid is supported and gets one bitmap
    row per LDS slice; the rmlp / neck dispatch predicates answer for the shipped head shapes."""
    import ctypes
    from emernerf_amd import _lib
    D, L, base, mx, T, F = grid
    growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
    desc = _lib.make_grid_desc(D, L, F, T, base, growth)
    assert hip_lib.emer_hashgrid_sliced_supported(ctypes.byref(desc)) == 1
    assert hip_lib.emer_hashgrid_mask_rows(ctypes.byref(desc)) == rows
    # the level at which a data-parallel trainer cuts the table backward into two launches: the finest levels that fill one round
    # of 256 resident owners (cfg 2: 4 hashed levels x 64 slices), never level 0 and never "everything"
    k = hip_lib.emer_hashgrid_sliced_split_level(ctypes.byref(desc))
    assert 0 < k < L and k == {(3, 16, 16, 2048, 19, 2): 12}.get(grid, k)
    assert hip_lib.emer_rmlp_supported(3, 40, 4, 64, 6) == 1 and hip_lib.emer_rmlp_supported(2, 64, 0, 64, 1) == 1
    assert hip_lib.emer_rmlp_supported(3, 64, 0, 64, 64) == 1 and hip_lib.emer_rmlp_supported(2, 43, 0, 32, 5) == 0
    assert hip_lib.emer_neck_supported(16, 2, 64, 64) == 1 and hip_lib.emer_neck_supported(10, 4, 64, 128) == 1


@pytest.mark.parametrize("grid,split", [((3, 16, 16, 2048, 19, 2), {}),              # cfg-2 main grid: 1900 dense items fill the tail, no hashed level is cut
                                        ((4, 10, 32, 8192, 18, 4), {8: 2, 9: 2}),   # dynamic xyzt table: 640 hashed items, remainder 128 -> two levels in halves
                                        ((4, 10, 16, 4096, 18, 4), {9: 4}),         # flow xyzt table: 576 hashed items + one dense level, remainder 64 -> quarters
                                        ((3, 10, 16, 8192, 20, 4), {}),             # default static table: 7 x 256 hashed items = whole rounds
                                        ((3, 8, 16, 2048, 20, 1), {})])             # proposal net: 4 x 64 hashed items = one round
def test_tail_items_of_the_owner_computes_backward(hip_lib, grid, split):
    """[r5] The hashed work items of the owner-computes grid backward are taken in rounds of 256 owners; where their count leaves a
    remainder that the dense levels' small items cannot fill, the finest levels are cut in R sample ranges (merged with atomics) so that the
    last round is full of 1/R-size items (emer_hashgrid_sliced_plan: host arithmetic).  Hashed levels otherwise have ONE range (every entry
    written once with plain stores); dense levels always have several."""
    import ctypes
    from emernerf_amd import _lib
    D, L, base, mx, T, F = grid
    growth = float(np.exp((np.log(mx) - np.log(base)) / (L - 1)))
    desc = _lib.make_grid_desc(D, L, F, T, base, growth)
    ns, nr = (ctypes.c_uint32 * L)(), (ctypes.c_uint32 * L)()
    total = hip_lib.emer_hashgrid_sliced_plan(ctypes.byref(desc), ns, nr)
    assert total == sum(ns[l] * nr[l] for l in range(L)) > 0
    hashed = [l for l in range(L) if desc.hashed[l]]
    assert {l: nr[l] for l in hashed if nr[l] != 1} == split, [(l, ns[l], nr[l]) for l in range(L)]
    assert all(nr[l] > 1 for l in range(L) if not desc.hashed[l] and ns[l] * nr[l] > 1) or not [l for l in range(L) if not desc.hashed[l]]
    n_hashed = sum(ns[l] for l in hashed)
    if split:   # the cut levels' items fill the owners' last round exactly or in whole multiples of 1/R rounds
        rem = n_hashed % 256
        cut_items = sum(ns[l] * nr[l] for l in split)
        assert rem != 0 and sum(ns[l] for l in split) >= rem and cut_items % 256 in (0, 128, 64, 192)


def test_inputs_are_detached_outside_autograd_recording():
    """A Function's forward sees needs_input_grad == requires_grad whatever the grad mode (and is_grad_enabled() is False inside
    every forward), so the public wrappers detach under torch.no_grad(): nothing is then saved for a backward that cannot run."""
    import torch
    from emernerf_amd import fused, ops
    w, x = torch.ones(3, requires_grad=True), torch.ones(3)
    for ng in (fused._ng, ops._ng):
        a = ng(x, w, 5, None)
        assert a[1] is w and a[0] is x
        with torch.no_grad():
            b = ng(x, w, 5, None)
        assert not b[1].requires_grad and b[1].data_ptr() == w.data_ptr() and b[2] == 5 and b[3] is None

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            Probe.seen = (ctx.needs_input_grad[0], torch.is_grad_enabled())
            return t * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    with torch.no_grad():
        Probe.apply(w)
    assert Probe.seen == (True, False), "the premise changed: revisit fused._ng"
    with torch.no_grad():
        Probe.apply(*fused._ng(w))
    assert Probe.seen == (False, False)


def test_ctypes_signatures_have_the_declared_arity():
    """Every entry point is bound with as many ctypes argument types as its declaration in include/emernerf_hip.h has
    parameters (a call with a stale argtypes list would pass the wrong registers to the kernel launch, silently)."""
    from emernerf_amd import _lib
    src = open(os.path.join(ROOT, "include", "emernerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(emer_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)}
    checked = 0
    for table in (_lib.SIGNATURES, _lib.INT64_FUNCTIONS):
        for name, argtypes in table.items():
            params = decl[name].strip()
            n = 0 if params in ("", "void") else params.count(",") + 1
            assert n == len(argtypes), f"{name}: header declares {n} parameters, _lib binds {len(argtypes)}"
            for i, (par, t) in enumerate(zip([q.strip() for q in params.split(",")], argtypes)):
                if "*" in par:
                    assert t is _lib._P or (isinstance(t, type) and issubclass(t, ctypes._Pointer)), \
                        f"{name}: parameter {i} is `{par}` in the header but bound as {t}"
                    continue
                else:
                    ctype = par.replace("const", "").split()[0] if par.replace("const", "").split()[0] != "unsigned" else "unsigned"
                    want = {"int64_t": (ctypes.c_int64,), "int32_t": (ctypes.c_int32, ctypes.c_int), "int": (ctypes.c_int, ctypes.c_int32),
                            "uint64_t": (ctypes.c_uint64,), "uint32_t": (ctypes.c_uint32,), "float": (ctypes.c_float,)}[ctype]
                assert t in want, f"{name}: parameter {i} is `{par}` in the header but bound as {t}"
            checked += 1
    assert checked >= 60
