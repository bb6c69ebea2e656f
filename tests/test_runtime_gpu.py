"""Whole-step parity of the host runtime (libkernel: layer graph + worker thread + plugin +
kernels through the C-ABI) against the CPU oracle net, on identical synthetic inputs, weights and
dropout masks.  Tolerance: fp32 contraction tolerance rel 1e-4 (of the tensor's RMS) after a few
SGD steps, as BASELINE.json's north_star states; costs to 1e-4 relative."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import modelutil
from oracle.golden_net import GoldenNet

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_child(_env=None, **kw):
    out = os.path.join(tempfile.mkdtemp(prefix="rt_"), "res.npz")
    cmd = [sys.executable, "-u", os.path.join(HERE, "run_net_step.py"), "--out", out]
    for k, v in kw.items():
        cmd += ["--" + k.replace("_", "-"), str(v)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(_env or {})))
    assert r.returncode == 0, "child failed:\n" + r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return dict(np.load(out)), r.stdout


def rel_rms(got, exp):
    exp = exp.astype(np.float64)
    scale = max(float(np.sqrt((exp ** 2).mean())), 1e-30)
    return float(np.abs(got.astype(np.float64) - exp).max() / scale)


def oracle_run(net_name, cases, mini, seed=42, test=False, world=1, init_w=0.0):
    sys.path.insert(0, HERE)
    from run_net_step import make_masks
    layers, dims, classes = modelutil.load_net(net_name, seed=seed, init_w=init_w)
    data = modelutil.synthetic_batch(dims, classes, cases)
    net = GoldenNet(layers, world=world)
    masks = make_masks(layers, cases, mini)
    names = [nm for nm, _ in modelutil.dropout_layers(layers)]
    total = {}
    for b, s in enumerate(range(0, cases, mini)):
        mb = [d[:, s:s + mini] for d in data]
        if test:
            costs = net.fprop(mb, train=False)
        else:
            costs = net.train_step(mb, masks=dict(zip(names, masks[b])))
        for k, v in costs.items():
            total[k] = np.array(v) if k not in total else total[k] + np.array(v)
    return net, layers, total


def rel_l2(got, exp):
    exp = exp.astype(np.float64)
    return float(np.linalg.norm(got.astype(np.float64) - exp) / max(np.linalg.norm(exp), 1e-30))


# (net, cases, minibatch, metric): a full minibatch plus a ragged tail whose size is not a multiple
# of 4.  Metric "max": max |err| / RMS < 1e-4 on weights after the steps (north_star's tolerance).
# conv-local starts from initW = 1e-4 with zero biases, so its activations hover around zero and its
# first max-pool sees ~10^6 windows of nearly equal values: a 1e-6 rounding difference upstream flips
# a handful of argmax / relu decisions, each moving single gradient entries by O(1) (the routing
# itself is bit-exact given equal inputs: test_kernels_gpu.py::test_pool_and_undo).  For that net
# the comparison is one minibatch in the Frobenius norm.
CASES = [("gc-small", 128 + 70, 128, "max"), ("nin-tiny", 128 + 70, 128, "max"), ("80sec", 128 + 36, 128, "max"),
         ("18pct", 64, 64, "max"), ("conv-local", 34, 64, "l2"),
         # SURVEY 8f.3 layers: rgb2lab / rgb2yuv / blur / resize in front of two convs joined by eltmax, then blur + nailbed
         ("image-layers", 64 + 22, 64, "max"),
         # random sparse connectivity in a conv (groups 4, overSample 2) and a local layer (groups 2)
         ("gc-sparse", 128 + 70, 128, "max"),
         # the remaining config pairs the reference ships (NIN cifar-100 / mnist / svhn, layers-19pct), full size, small batches.
         # cifar-100 starts from initW = 0.001 and the mnist net sees a 1-channel U[0,1) image: both have near-tied max-pool
         # windows like conv-local above (bias gradients, sums with heavy cancellation, move by 2e-3 of their RMS) -> Frobenius norm
         ("nin-cifar100", 16 + 6, 16, "l2"), ("nin-mnist", 32 + 6, 32, "l2"), ("nin-svhn", 16 + 6, 16, "max"), ("19pct", 64 + 6, 64, "max"),
         # the same three nets at the STRICT metric with every weight matrix redrawn at a larger scale (fifth field; sixth: weight
         # seed): activations are no longer near zero and max-err 1e-4 holds -- the loosened metric above is about ties, not about
         # the kernels.  conv-local at the BASELINE minibatch of 128 plus the 16-case tail of a 10 000-case batch has 2 x 10^6
         # max-pool windows per step, so even well-scaled weights meet a window whose two largest entries differ by less than
         # the fp32 rounding of the conv every few draws: weight seed 43 (and 42 at 64 + 16 cases) has none and matches to 1e-5;
         # seeds 42 and 44 at 128 + 16 flip a routing decision.
         ("conv-local", 128 + 16, 128, "max", 0.05, 43), ("conv-local", 64 + 16, 64, "max", 0.05), ("nin-cifar100", 16 + 6, 16, "max", 0.05),
         ("nin-mnist", 32 + 6, 32, "max", 0.05),
         # BASELINE config 5 as a whole net: 224x224 input, 11x11 stride-4 conv, cmrnorm at 96x55^2 and 256x27^2, the 2x2 max pools,
         # fc 4608 -> 2048 -> 2048 -> 100 (an 8-case minibatch and a 3-case tail keep the oracle to seconds).  Its own initW
         # (1e-4 / 0.01, zero biases) leaves fc3's pre-activations at 1e-8: the sign the relu takes there is rounding noise,
         # so the weights are redrawn at 0.02 (unit gain per layer).  Weight seeds 43, 44 and 45 match to 3.5e-5; with seed 42
         # one fc1 unit of the second minibatch sits within rounding of zero and its relu takes the other side (measured
         # with tools/debug_parity.py: every activation agrees to 1e-5, neuron1's gradient differs in that unit's row).
         ("imagenet", 8 + 3, 8, "max", 0.02, 43),
         # cnorm, eltsum, cost.sum2, a grouped conv, a two-input fc and an fc straight from the data (shape of the reference's
         # example-layers/layers-example.cfg)
         ("example-small", 64 + 22, 64, "max"),
         # the reference's own gradient-check fixture (verbatim layers.gc.cfg / layer-params.gc.cfg) as a training step; weight seed 7:
         # with seed 42 one case's p[label] underflows to 0 and the cost is inf on both sides
         ("gc", 64 + 22, 64, "max", 0.0, 7),
         # THE net the headline number is quoted on, with its shipped parameters (NIN/cifar-10_def + cifar-10_params: epsW = 1,
         # wc = 1e-5, coeff = 0.1), full size, a 16-case minibatch and a 6-case tail.  Frobenius norm: conv1's relu has 3 x 10^6 units
         # per 16 cases and meets a pre-activation within fp32 rounding of zero about once per step, which moves single entries of
         # conv1's gradient only (every other tensor agrees to 6e-6 at max error); the strict, flip-counting comparison of this net
         # is test_one_step_strict_whatever_the_ties below
         ("nin", 16 + 6, 16, "l2")]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_train_step_matches_oracle(case):
    net_name, cases, mini, metric = case[:4]
    init_w = case[4] if len(case) > 4 else 0.0
    seed = case[5] if len(case) > 5 else 42
    got, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=init_w, seed=seed)
    net, layers, costs = oracle_run(net_name, cases, mini, init_w=init_w, seed=seed)
    assert int(got["numCases"]) == cases
    for k, v in costs.items():
        np.testing.assert_allclose(got["cost:" + k][0], v[0], rtol=2e-4, atol=1e-3)
        # the error count depends on exact ties between near-uniform class probabilities
        # (correctProb = 1/numMax, layer_kernels.cu:50-72): allow one case of slack
        if len(v) > 1:
            assert abs(got["cost:" + k][1] - v[1]) <= 1.0
    worst, table, bad = 0.0, [], []
    for l in layers:
        if "weights" not in l:
            continue
        for i in range(len(l["weights"])):
            key = (l["name"], i)
            if not isinstance(net.W[key], np.ndarray):
                continue
            err = rel_rms if metric == "max" else rel_l2
            e1 = err(got["W:%s:%d" % key], net.W[key])
            e2 = err(got["Winc:%s:%d" % key], net.Winc[key])
            table.append((l["name"], i, e1, e2))
            worst = max(worst, e1)
        eb = err(got["B:" + l["name"]], net.B[l["name"]])
        ebi = err(got["Binc:" + l["name"]], net.Binc[l["name"]])
        table.append((l["name"], "bias", eb, ebi))
    tolW, tolInc = (1e-4, 2e-3) if metric == "max" else (1e-3, 5e-3)
    for row in table:
        print("%-10s %-5s W %.2e  inc %.2e" % row)
        if not (row[2] < tolW and row[3] < tolInc):
            bad.append(row)
    assert not bad, bad
    print("worst weight rel err", worst)


# (net, minibatch, init_w, weight seed): one training step each, at the strict metric, for EVERY seed -- including the ones the
# multi-minibatch cases above avoid because a max-pool or relu decision flips (conv-local 42 / 44, imagenet 42) -- and for NIN
# cifar-10 with its shipped parameters, whose first relu has 3 x 10^6 units per 16 cases and meets a pre-activation within
# rounding of zero about once per step.
DECISION_CASES = [("nin", 16, 0.0, 42), ("nin", 16, 0.0, 43), ("conv-local", 128, 0.05, 42), ("conv-local", 128, 0.05, 43), ("conv-local", 128, 0.05, 44),
                  ("imagenet", 8, 0.02, 42), ("80sec", 128, 0.0, 42)]


@pytest.mark.parametrize("case", DECISION_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_one_step_strict_whatever_the_ties(case):
    """The flip-tolerant strict metric.  Back-propagation has two discontinuous decisions: a relu's gradient mask (output > 0)
    and a max-pool's equality routing.  When a pre-activation lies within fp32 rounding of zero, or a window's two largest
    entries within rounding of each other, two correct implementations may decide differently and single gradient entries
    move by O(1); the forward values agree to 1e-6 all the same.  So: the GPU's forward outputs of every relu and max-pool
    layer are read back, checked against the oracle's forward pass (max error 1e-4 of the RMS), and handed to the oracle's
    backward pass, which takes those two decisions from them (GoldenNet.bprop(decide=...)).  The decisions that differ are
    COUNTED (at most 10 per step, each at a relu input within 1e-5 of zero relative to the layer's RMS, or a routing change)
    and everything downstream is then compared at the strict tolerance: weights 1e-4, increments 2e-4 of their RMS."""
    net_name, mini, init_w, seed = case
    sys.path.insert(0, HERE)
    from run_net_step import make_masks
    layers, dims, classes = modelutil.load_net(net_name, seed=seed, init_w=init_w)
    relus = [l["name"] for l in layers if l["type"] == "neuron" and l["neuron"]["type"] == "relu"]
    pools = [l["name"] for l in layers if l["type"] == "pool" and l["pool"] == "max"]
    got, _ = run_child(net=net_name, cases=mini, mini=mini, init_w=init_w, seed=seed, acts=",".join(relus + pools))
    data = modelutil.synthetic_batch(dims, classes, mini)
    net = GoldenNet(layers)
    masks = make_masks(layers, mini, mini)
    names = [nm for nm, _ in modelutil.dropout_layers(layers)]
    costs = net.fprop(data, train=True, masks=dict(zip(names, masks[0])))
    np.testing.assert_allclose(got["cost:logprob"][0], costs["logprob"][0], rtol=2e-4, atol=1e-3)
    decide = {}
    for nm in relus + pools:
        assert rel_rms(got["acts:" + nm], net.acts[nm]) < 1e-4, nm          # the forward pass agrees everywhere
        decide[nm] = got["acts:" + nm]
    net.bprop(train=True, decide=decide)
    by = dict((l["name"], l) for l in layers)
    flips = 0
    for nm, idx in net.flips.items():
        flips += len(idx)
        if len(idx) and by[nm]["type"] == "neuron":      # a flipped relu sits at a pre-activation within rounding of zero
            pre = net.acts[layers[by[nm]["inputs"][0]]["name"]]
            assert np.abs(pre.reshape(-1)[idx]).max() < 1e-5 * float(np.sqrt((pre.astype(np.float64) ** 2).mean())), nm
        print("%-16s decisions that differ: %d" % (nm, len(idx)))
    assert flips <= 10, net.flips
    net.update()
    bad = []
    for l in layers:
        if "weights" not in l:
            continue
        rows = [("W:%s:%d" % (l["name"], i), net.W[(l["name"], i)], "Winc:%s:%d" % (l["name"], i), net.Winc.get((l["name"], i)))
                for i in range(len(l["weights"])) if isinstance(net.W[(l["name"], i)], np.ndarray)]
        rows.append(("B:" + l["name"], net.B[l["name"]], "Binc:" + l["name"], net.Binc[l["name"]]))
        for kw, w, ki, inc in rows:
            e1, e2 = rel_rms(got[kw], w), rel_rms(got[ki], inc)
            print("%-20s W %.2e  inc %.2e" % (kw, e1, e2))
            if not (e1 < 1e-4 and e2 < 2e-4):
                bad.append((kw, e1, e2))
    assert not bad, bad
    print("decisions that differed:", flips)


@pytest.mark.parametrize("net_name,cases,mini", [("80sec", 6 * 32 + 7, 32), ("example-small", 5 * 16 + 3, 16), ("conv-local", 5 * 16, 16)])
def test_captured_steps_equal_eager_steps(net_name, cases, mini):
    """Nets without random layers replay their training step as a HIP graph from the third minibatch of a shape on
    (ConvNet::trainStep).  Same kernels, same arguments: weights, increments and biases after the batch are bit-identical to
    an all-eager run (MC_NO_STEP_GRAPH=1), the costs agree (device-side double accumulation vs host-side), and the ragged tail
    -- a shape seen once -- runs eagerly in both.  Also against the oracle, like every other whole-step test."""
    got, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, _env={"MC_STEP_GRAPH": "1"})   # replay, no timing trial
    ref, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, _env={"MC_NO_STEP_GRAPH": "1"})
    full = cases // mini
    assert int(got["graphReplays"]) == full - 2 and int(ref["graphReplays"]) == 0
    assert int(got["numCases"]) == int(ref["numCases"]) == cases
    for k in ref:
        if k.startswith("cost:"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-12, atol=1e-9)
        elif k[0] in "WB":
            assert np.array_equal(got[k], ref[k]), k
    net, layers, costs = oracle_run(net_name, cases, mini, init_w=0.05)
    for k, v in costs.items():
        np.testing.assert_allclose(got["cost:" + k][0], v[0], rtol=2e-4, atol=1e-3)


def test_failed_capture_falls_back_to_a_clean_eager_step():
    """A step capture that fails half way (here: a test hook throws after the captured bprop, before the update) has walked the
    host side of part of a step without running anything: weight tensors count a queued gradient, layers hold arrival counters
    and per-pass flags.  ConvNet::captureStep puts all of that back, so the eager step that follows -- and the rest of the batch
    -- is bit-identical to a run that never tried to capture."""
    cases, mini = 5 * 32 + 7, 32
    got, _ = run_child(net="80sec", cases=cases, mini=mini, init_w=0.05, _env={"MC_STEP_GRAPH": "1", "MC_TEST_FAIL_CAPTURE": "1"})
    ref, _ = run_child(net="80sec", cases=cases, mini=mini, init_w=0.05, _env={"MC_NO_STEP_GRAPH": "1"})
    assert int(got["graphReplays"]) == 0 and int(got["numCases"]) == int(ref["numCases"]) == cases
    for k in ref:
        if k.startswith("cost:"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-12, atol=1e-9)
        elif k[0] in "WB":
            assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("net_name,cases,mini", [("nin-tiny", 3 * 32 + 7, 32), ("example-small", 2 * 16 + 3, 16)])
def test_warm_up_leaves_no_trace(net_name, cases, mini):
    """Before the first training step of a batch the runtime dry-runs one forward + backward pass per minibatch shape (full and
    ragged tail: ConvNet::warmUpShapes) so that matrices are sized and contraction tiles tuned outside the steps.  The dry run
    must not move the random stream (nin-tiny draws dropout masks from it), leave gradients or costs behind, or touch weights:
    the batch ends bit-identical to one without the dry run."""
    got, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, philox_seed=7)
    ref, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, philox_seed=7, _env={"MC_NO_WARMUP": "1"})
    assert int(got["numCases"]) == int(ref["numCases"]) == cases
    assert int(got["warmedShapes"]) == 2 and int(ref["warmedShapes"]) == 0     # the full shape and the tail
    for k in ref:
        if k.startswith("cost:") or k[0] in "WB":
            assert np.array_equal(got[k], ref[k]), k


def test_graph_or_eager_trial_changes_no_result():
    """By default the four steps after a capture alternate between the replayed graph and an eager (forking) step, timed, and the
    faster way is kept (ConvNet::trainStep).  Whichever way wins, and through the trial itself, the batch leaves behind exactly
    what an all-eager single-stream run leaves."""
    cases, mini = 9 * 32 + 5, 32
    got, _ = run_child(net="80sec", cases=cases, mini=mini, init_w=0.05)
    ref, _ = run_child(net="80sec", cases=cases, mini=mini, init_w=0.05, _env={"MC_NO_STEP_GRAPH": "1", "MC_NO_STEP_FORK": "1"})
    replays, forked = int(got["graphReplays"]), int(got["forkedSteps"])
    assert replays >= 2 and forked >= 2 and replays + forked == cases // mini - 2     # two trial steps each way, then one of the two
    assert int(got["numCases"]) == int(ref["numCases"]) == cases
    for k in ref:
        if k.startswith("cost:"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-12, atol=1e-9)
        elif k[0] in "WB":
            assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("net_name,cases,mini", [("nin-tiny", 6 * 32 + 7, 32), ("conv-local", 5 * 16, 16), ("example-small", 4 * 16 + 3, 16)])
def test_forked_steps_equal_single_stream_steps(net_name, cases, mini):
    """Eager training steps run their weight-gradient products on a second stream from the third minibatch of a shape on
    (ConvNet::trainStep / WeightLayer::bpropCommon).  Same kernels, same buffers, same order per buffer: everything the batch
    leaves behind is bit-identical to a run that keeps every step on one stream (MC_NO_STEP_FORK=1).  Graphs are off in both
    runs so that every step is an eager one (nin-tiny has dropout and would stay eager anyway)."""
    got, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, _env={"MC_NO_STEP_GRAPH": "1"})
    ref, _ = run_child(net=net_name, cases=cases, mini=mini, init_w=0.05, _env={"MC_NO_STEP_GRAPH": "1", "MC_NO_STEP_FORK": "1"})
    assert int(got["forkedSteps"]) == cases // mini - 2 and int(ref["forkedSteps"]) == 0
    for k in ref:
        if k.startswith("cost:") or k[0] in "WB":
            assert np.array_equal(got[k], ref[k]), k


def test_test_pass_costs_and_no_update():
    got, _ = run_child(net="nin-tiny", cases=150, mini=64, test=1, acts="probs")
    net, layers, costs = oracle_run("nin-tiny", 150, 64, test=True)
    np.testing.assert_allclose(got["cost:logprob"], costs["logprob"], rtol=2e-4, atol=1e-3)
    fresh, _, _ = modelutil.load_net("nin-tiny", seed=42)
    for l in fresh:
        if "weights" in l:
            np.testing.assert_array_equal(got["W:%s:0" % l["name"]], l["weights"][0])
    assert rel_rms(got["acts:probs"], net.acts["probs"]) < 1e-4     # last (tail) minibatch: 22 cases


@pytest.mark.parametrize("net_name", ["gc-small", "gc-sparse"])
def test_gradient_check_passes(net_name):
    """The reference's --check-grads on layers.gc.cfg-style nets (dense, and with randSparse layers as the
    reference's own layers.gc.cfg has): every tensor below 2% rel error."""
    got, out = run_child(net=net_name, cases=32, mini=32, check_grads=1)
    assert int(got["failures"]) == 0, out[-2000:]
    assert "TESTS PASSED" in out


def test_reference_gc_fixture_through_check_grads():
    """The reference's OWN integration fixture, verbatim (configs/layers.gc.cfg + layer-params.gc.cfg = example-layers/ of the
    reference: conv with linear[3,2.2], avg-pool with start=-2, local layer with groups=2 / randSparse / stride 3 / tanh, two-input
    fc), through --check-grads as Kernel/src/convnet/convnet.cu:262-335 runs it: every element of every tensor perturbed, forward
    differences, relative L2 error against 0.02.  On the fork's dummy data (U[0,1), Python/data.py:236 "changed to rand") the
    fixture is numerically marginal by construction, which the CPU oracle shows with the same rule (tests/test_golden_net.py's
    helper on this net): initW = 0.8 saturates the softmax (weight seed 42 has a case with p[label] == 0 in fp32: cost inf), and
    fc10's second weight matrix (initW 0.008, finite-difference step 0.1 >> the weights) sits at 0.03-0.036 relative error in
    float64 arithmetic -- a property of the fixture, not of an implementation.  So: weight seed 7 (finite cost), every tensor
    must pass except that one, which may land on either side of 0.02."""
    got, out = run_child(net="gc", cases=32, mini=32, check_grads=1, seed=7)
    verdicts = dict((m.group(2), m.group(1)) for m in re.finditer(r"\((PASS|\*\*\*\*FAIL\*\*\*\*)\) (.+) GRADIENT CHECK", out))
    assert len(verdicts) == 7, out[-3000:]        # conv32 / local32 weights + biases, fc10 weights[0], weights[1], biases
    failed = sorted(k for k, v in verdicts.items() if v != "PASS")
    marginal = [k for k in verdicts if "fc10" in k and "[1]" in k]
    assert len(marginal) == 1, verdicts
    assert set(failed) <= set(marginal), out[-3000:]
    assert int(got["failures"]) == len(failed)


def test_image_layers_forward_matches_oracle():
    """Test-mode pass of the image-layers net: the activations of every new layer type against the oracle net
    (last minibatch: a 22-case tail, the scalar code path of the kernels).  'lab' and 'emax' are not read back:
    their sole consumers are the blur layers, which -- as in the reference (layer.py:307-323) -- write in place."""
    names = ["yuv", "blur0", "resize", "blur1", "nails"]
    got, _ = run_child(net="image-layers", cases=86, mini=64, test=1, acts=",".join(names))
    net, layers, costs = oracle_run("image-layers", 86, 64, test=True)
    for nm in names:
        tol = 1e-5 if nm in ("yuv", "lab", "blur0", "resize") else 1e-4     # behind the convs: contraction tolerance
        assert rel_rms(got["acts:" + nm], net.acts[nm]) < tol, nm
    np.testing.assert_allclose(got["cost:logprob"], costs["logprob"], rtol=2e-4, atol=1e-3)


def test_multiview_test_matches_oracle():
    """startMultiviewTest (worker.cu:142-174): class probabilities averaged over the views of each case, then the
    logreg cost of the average.  3 views of 150 cases, minibatch 64 (two full real minibatches and a 22-case tail)."""
    sys.path.insert(0, HERE)
    from run_net_step import multiview_batch
    from oracle import golden
    cases, views, mini = 150, 3, 64
    got, _ = run_child(net="nin-tiny", cases=cases, mini=mini, multiview=views)
    layers, dims, classes = modelutil.load_net("nin-tiny", seed=42)
    data = multiview_batch(dims, classes, cases, views)
    net = GoldenNet(layers)
    exp = np.zeros(2)
    for s in range(0, cases, mini):
        e = min(cases, s + mini)
        probs = 0
        for v in range(views):
            net.fprop([d[:, v * cases + s:v * cases + e] for d in data], train=False)
            probs = probs + net.acts["probs"]
        probs = (probs * np.float32(1.0 / views)).astype(np.float32)
        exp += np.array(golden.logreg_cost(data[1][:, s:e], probs))
    assert int(got["numCases"]) == cases
    np.testing.assert_allclose(got["cost:logprob"][0], exp[0], rtol=2e-4)
    assert abs(got["cost:logprob"][1] - exp[1]) <= 1.0


def test_feature_writer_matches_oracle():
    """startFeatureWriter (worker.cu:191-214): the chosen layer's test-mode activations, transposed to
    (cases, outputs) C order and written into the caller's matrix, minibatch by minibatch (ragged tail)."""
    cases, mini = 150, 64
    got, _ = run_child(net="nin-tiny", cases=cases, mini=mini, features="pool2")
    layers, dims, classes = modelutil.load_net("nin-tiny", seed=42)
    data = modelutil.synthetic_batch(dims, classes, cases)
    net = GoldenNet(layers)
    rows = []
    for s in range(0, cases, mini):
        net.fprop([d[:, s:s + mini] for d in data], train=False)
        rows.append(net.acts["pool2"].T.copy())
    exp = np.concatenate(rows, axis=0)
    assert got["ftrs"].shape == exp.shape and not np.isnan(got["ftrs"]).any()
    assert rel_rms(got["ftrs"], exp) < 1e-4
    assert int(got["numCases"]) == cases


def test_counter_dropout_equals_stored_masks():
    """Without injected masks the dropout neuron regenerates its uniforms in the forward and backward kernels
    (mc_dropout_counter) instead of storing a mask: the trained weights equal, bit for bit, a run that is fed the
    same stream as stored masks (mc_rand_uniform at the same seed and offsets -> mc_dropout_apply)."""
    a, _ = run_child(net="nin-tiny", cases=128 + 70, mini=128, philox_seed=4321)
    b, _ = run_child(net="nin-tiny", cases=128 + 70, mini=128, inject_philox=4321)
    c, _ = run_child(net="nin-tiny", cases=128 + 70, mini=128, philox_seed=4322)
    for k in a:
        if k.startswith(("W:", "B:", "Winc:", "Binc:", "cost:")):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert not np.array_equal(a["W:conv2:0"], c["W:conv2:0"])        # another seed is another mask


@pytest.mark.parametrize("net_name,cases,mini", [("nin-tiny", 128 + 70, 128), ("nin", 2 * 16 + 6, 16)])
def test_dropout_fused_into_the_pools_changes_no_bit(net_name, cases, mini):
    """The max-pool layers below NIN's dropout layers apply the dropout in their own store and its backward step in their own
    max-undo (MaxPoolLayer::fpropActs / bpropActs, mc_conv_local_pool_dropout / mc_conv_local_max_undo_dropout): same uniforms
    from the same stream positions, so everything the batch leaves behind equals a run with the separate kernels
    (MC_NO_FUSE_DROPOUT=1) bit for bit -- full and ragged minibatches, and the forked steps from the third minibatch on."""
    a, _ = run_child(net=net_name, cases=cases, mini=mini, philox_seed=4321)
    b, _ = run_child(net=net_name, cases=cases, mini=mini, philox_seed=4321, _env={"MC_NO_FUSE_DROPOUT": "1"})
    for k in a:
        if k.startswith(("W:", "B:", "Winc:", "Binc:", "cost:")):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("net_name,cases,mini", [("80sec", 4 * 32 + 12, 32), ("18pct", 3 * 16, 16)])
def test_relu_fused_into_the_pool_changes_no_bit(net_name, cases, mini):
    """layers-80sec / 18pct put `neuron=relu` on pool1: the max-pool layer applies it in its own store and the relu's backward step in
    its max-undo (NeuronLayer::fuseReluIntoPool, mc_conv_local_pool_relu / mc_conv_local_max_undo_relu_out).  Everything the batch
    leaves behind equals a run with the separate kernels (MC_NO_FUSE_POOL_RELU=1) bit for bit."""
    a, _ = run_child(net=net_name, cases=cases, mini=mini)
    b, _ = run_child(net=net_name, cases=cases, mini=mini, _env={"MC_NO_FUSE_POOL_RELU": "1"})
    for k in a:
        if k.startswith(("W:", "B:", "Winc:", "Binc:", "cost:")):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("net_name,cases,mini", [("nin-tiny", 128, 64), ("nin-tiny", 6 * 32, 32), ("conv-local", 5 * 16, 16)])
def test_single_rank_rccl_path_is_identity(net_name, cases, mini):
    """world_size 1 through the real RCCL all-reduce / side-stream / event path gives the same
    weights as the path without a communicator.  The longer runs reach the steps that produce their weight gradients on the
    branch stream (third minibatch of a shape on), so the buckets' all-reduces are ordered behind that stream; conv-local's
    unshared layers are large enough for buckets to go out in the middle of back-propagation."""
    a, _ = run_child(net=net_name, cases=cases, mini=mini, _env={"MC_NO_STEP_GRAPH": "1"})
    b, _ = run_child(net=net_name, cases=cases, mini=mini, single_rank_comm=1)
    assert int(b["forkedSteps"]) == max(0, cases // mini - 2) and int(b["graphReplays"]) == 0
    for k in a:
        if k.startswith(("W:", "B:")):
            np.testing.assert_array_equal(a[k], b[k])
