"""Trainer-level checks on the GPU: hipGraph replay of a step == eager launches; gradient sinks == autograd accumulation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(use_graph):
    from emernerf_amd.trainer import Trainer, synthetic_rays
    dev = torch.device("cuda:0")
    tr = Trainer(kind="static", device=dev, num_samples=32, prop_samples=(32, 16), table_init=0.3, seed=3, use_graph=use_graph)
    jit = torch.full((512,), 0.37, device=dev)
    tr.estimator.jitter_fn = lambda n, d: jit  # deterministic stratified jitter: both runs see identical samples
    return tr, synthetic_rays(512, dev, seed=5)


def test_graph_replay_equals_eager(hip_lib):
    """Seven optimizer steps (both step types: with and without proposal-net training) with the forward+backward replayed
    from captured hipGraphs give the same parameters as eager launches (fp32 atomics order aside)."""
    eager, data = _make(False)
    graph, _ = _make(True)
    for tr in (eager, graph):
        tr.step_count = 0
    le, lg = [], []
    for _ in range(7):
        le.append(float(eager.train_step(data)["loss"]))
        lg.append(float(graph.train_step(data)["loss"]))
    assert graph.use_graph and len(graph._graphs) == 2, "both step types must have been captured"
    assert max(abs(a - b) for a, b in zip(le, lg)) < 1e-5 * max(abs(v) for v in le)
    pe, pg = eager.flat.params, graph.flat.params
    assert float((pe - pg).abs().max()) <= 2e-4 * float(pe.abs().max())
