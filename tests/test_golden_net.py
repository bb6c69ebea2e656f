"""Pins the graph plumbing of the CPU oracle net (oracle/golden_net.py) with the reference's own
integration check: forward finite differences against the analytic gradient, relative L2 error
below 0.02 per tensor (Kernel/src/convnet/convnet.cu:285-335, Kernel/include/util.cuh:46-47), on
a net modelled on example-layers/layers.gc.cfg and on a reduced NIN (cccp + dropout in its
test/grad-check form)."""
import os

import numpy as np
import pytest

import modelutil
from oracle.golden_net import GoldenNet


def _numeric_vs_analytic(net, data, tensor, step, picks, rng):
    costs = net.fprop(data, train=False, gc=True)
    base = net.cost_value()
    net.bprop(train=False, gc=True)
    N = data[0].shape[1]
    kind, key = tensor
    arr = net.W[key] if kind == "w" else net.B[key]
    grad = (net.wgrads[key].reshape(arr.shape) if kind == "w" else net.bgrads[key])
    idx = [np.unravel_index(i, arr.shape) for i in rng.choice(arr.size, size=min(picks, arr.size), replace=False)]
    num, ana = [], []
    for ij in idx:
        old = arr[ij]
        arr[ij] = old + np.float32(step)
        net.fprop(data, train=False, gc=True)
        num.append((net.cost_value() - base) / (N * step))
        arr[ij] = old
        ana.append(-float(grad[ij]) / N)
    num, ana = np.array(num), np.array(ana)
    return np.linalg.norm(num - ana) / max(np.linalg.norm(ana), 1e-30)


@pytest.mark.parametrize("name", ["gc-small", "nin-tiny", "image-layers", "gc-sparse", "example-small"])
def test_finite_differences(name):
    layers, dims, classes = modelutil.load_net(name, seed=7)
    data = modelutil.synthetic_batch(dims, classes, 8)
    if name == "nin-tiny":
        # initW=0.05 on these narrow layers lets the signal vanish (uniform softmax, gradients at the
        # fp32 noise floor of the cost); rescale layer by layer to unit output scale so it carries
        for l in layers:
            if "weights" in l:
                probe = GoldenNet(layers)
                probe.fprop(data, train=False, gc=True)
                for w in l["weights"]:
                    w *= np.float32(1.0 / max(float(probe.acts[l["name"]].std()), 1e-12))
    if name == "example-small":
        # forward differences want a smooth objective: relu -> logistic, abs -> square, max-pool -> avg-pool in this copy of the net
        # (with the kinks the error shrinks with the step -- 0.14 / 0.03 / 0.01 at 2e-3 / 5e-4 / 1e-4 -- until fp32 noise takes over);
        # what the case pins is the plumbing of cnorm, eltsum, cost.sum2, the grouped conv and the two-input fc
        for l in layers:
            if l["type"] == "neuron" and l["neuron"]["type"] in ("relu", "abs"):
                l["neuron"] = dict(l["neuron"], type="logistic" if l["neuron"]["type"] == "relu" else "square")
            if l["type"] == "pool":
                l["pool"] = "avg"
    net = GoldenNet(layers)
    rng = np.random.default_rng(3)
    for l in layers:
        if "weights" not in l:
            continue
        wstep = 0.1 if l["type"] in ("fc", "cccp") else 0.001
        bstep = 0.01 if l["type"] in ("fc", "cccp") else 0.002
        if name == "nin-tiny":
            wstep, bstep = 0.0005, 0.0005
        if name == "example-small":     # steps where neither the fp32 noise of the cost nor the curvature (square neuron) dominates
            wstep = 0.03 if l["name"] == "local8" else 0.01
            bstep = 0.001 if l["name"] == "conv8-3" else 0.01
        if name == "image-layers":      # fc inputs of O(1): the reference's 0.1 step leaves the softmax's linear range
            wstep, bstep = 0.002, 0.002
            if l["name"] == "conva":    # L*a*b* inputs of O(50) under a relu and a max: stay on one side of the kinks
                wstep = 0.0001
        for i in range(len(l["weights"])):
            rel = _numeric_vs_analytic(net, data, ("w", (l["name"], i)), wstep, 24, rng)
            assert rel < 0.02, (l["name"], i, rel)
        rel = _numeric_vs_analytic(net, data, ("b", l["name"]), bstep, 12, rng)
        assert rel < 0.02, (l["name"], "bias", rel)


def test_update_rule_and_costs():
    """One training step: costs are [-sum log p, errors]; weights move by inc = mom*inc + grad - wc*eps*W."""
    layers, dims, classes = modelutil.load_net("gc-small", seed=5)
    data = modelutil.synthetic_batch(dims, classes, 16)
    net = GoldenNet(layers)
    w0 = net.weight("conv32").copy()
    costs = net.train_step(data)
    assert set(costs) == {"logprob"} and len(costs["logprob"]) == 2
    assert costs["logprob"][0] > 0 and 0 <= costs["logprob"][1] <= 16
    l = [x for x in layers if x["name"] == "conv32"][0]
    expect = w0 + net.wgrads[("conv32", 0)].reshape(w0.shape) - np.float32(l["wc"][0] * l["epsW"][0]) * w0
    np.testing.assert_allclose(net.weight("conv32"), expect, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(net.Winc[("conv32", 0)], net.weight("conv32") - w0) if False else None


def test_data_parallel_equivalence_on_oracle():
    """2 shards x N cases with summed gradients == 1 x 2N cases (the identity the RCCL path relies on)."""
    layers, dims, classes = modelutil.load_net("gc-small", seed=11)
    data = modelutil.synthetic_batch(dims, classes, 16)
    full = GoldenNet(layers)
    full.fprop(data); full.bprop()
    shards = []
    for r in range(2):
        net = GoldenNet(layers, world=2)
        net.fprop([d[:, r * 8:(r + 1) * 8] for d in data]); net.bprop()
        shards.append(net)
    for key, gfull in full.wgrads.items():
        np.testing.assert_allclose(shards[0].wgrads[key] + shards[1].wgrads[key], gfull, rtol=2e-4, atol=1e-7)
    for key, gfull in full.bgrads.items():
        np.testing.assert_allclose(shards[0].bgrads[key] + shards[1].bgrads[key], gfull, rtol=2e-4, atol=1e-7)
