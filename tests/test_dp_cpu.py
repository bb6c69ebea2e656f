"""Data-parallel path on CPU (world_size 2, gloo): rendezvous + sharding + summed gradients give the
same weights on both ranks, equal to one process training on the global minibatches."""
import os
import socket
import subprocess
import sys

import numpy as np

import dp
import modelutil
from oracle.golden_net import GoldenNet

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_columns_partition():
    cols = [dp.shard_columns(300, 64, r, 2) for r in range(2)]
    both = np.sort(np.concatenate(cols))
    assert list(both) == list(range(300))                   # nothing dropped, nothing twice
    # rank r's slice of global minibatch i is contiguous and of equal size
    assert list(cols[0][:64]) == list(range(0, 64)) and list(cols[1][:64]) == list(range(64, 128))
    assert list(cols[0][128:]) == list(range(256, 278)) and list(cols[1][128:]) == list(range(278, 300))


def test_shard_columns_ragged_tail_keeps_every_case():
    """10 000 cases, minibatch 128, 3 ranks: 26 global minibatches of 384 and a tail of 16 = 6 + 5 + 5 (the reference's own tail
    semantics: every case is trained on, Kernel/src/data/data.cu:67-104)."""
    cols = [dp.shard_columns(10000, 128, r, 3) for r in range(3)]
    assert sorted(np.concatenate(cols).tolist()) == list(range(10000))
    assert [len(c) - 26 * 128 for c in cols] == [6, 5, 5]
    assert dp.global_cases(10000, 128, 3) == 10000
    # a tail shorter than the world size cannot be dealt out (a rank would miss the all-reduce): skipped, and accounted for
    cols = [dp.shard_columns(2 * 8 * 3 + 2, 8, r, 3) for r in range(3)]
    assert sorted(np.concatenate(cols).tolist()) == list(range(48))
    assert dp.global_cases(50, 8, 3) == 48


def test_two_rank_gloo_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    r0, r1 = dict(np.load(out + ".0.npz")), dict(np.load(out + ".1.npz"))
    for k in r0:
        np.testing.assert_array_equal(r0[k], r1[k])       # replicas stay identical

    layers, dims, classes = modelutil.load_net("gc-small", seed=11)
    data = modelutil.synthetic_batch(dims, classes, 43)
    net = GoldenNet(layers)
    want = {}
    for s in range(0, 43, 16):
        for name, vals in net.train_step([d[:, s:s + 16] for d in data]).items():
            want[name] = [a + float(b) for a, b in zip(want.get(name, [0.0] * len(vals)), vals)]
    # what rank 0 prints and checkpoints on is the GLOBAL batch: costs and case counts summed over the ranks (dp.sum_costs)
    assert int(r0["cases"]) == int(r1["cases"]) == 43
    for name, vals in want.items():
        np.testing.assert_allclose(r0["cost:" + name], vals, rtol=1e-5)
        np.testing.assert_array_equal(r0["cost:" + name], r1["cost:" + name])
    for l in layers:
        if "weights" in l:
            np.testing.assert_allclose(r0["W:" + l["name"]], net.weight(l["name"], 0), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(r0["B:" + l["name"]], net.B[l["name"]], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# bucket planner of the native runtime (csrc/runtime/buckets.h), compiled for the host and driven over stdin
# ---------------------------------------------------------------------------------------------------------------------
def _arena(layers):
    """(names, ends, contributors, layer index -> [tensor indices in the order bpropCommon completes them]) -- the arena order
    ConvNet::copyToGPU lays out: last layer first, biases before weights, owned tensors only, each padded to 64 elements."""
    names, ends, contributors, owner_of = [], [], [], {}
    at = 0
    for li in reversed(range(len(layers))):
        l = layers[li]
        if "weights" not in l:
            continue
        tensors = [("biases", l["biases"], -1)] + [("weights[%d]" % i, w, l["weightSourceLayerIndices"][i]) for i, w in enumerate(l["weights"])]
        for i, (tag, arr, src) in enumerate(tensors):
            if src >= 0:
                continue
            at += (arr.size + 63) // 64 * 64
            owner_of[(li, tag)] = len(names)
            names.append("%s.%s" % (l["name"], tag))
            ends.append(at)
            contributors.append(0)
    arrivals = {}
    for li in reversed(range(len(layers))):
        l = layers[li]
        if "weights" not in l:
            continue
        seq = []
        if l["epsB"] > 0:
            seq.append(owner_of[(li, "biases")])
        for i in range(len(l["weights"])):
            if l["epsW"][i] <= 0:
                continue
            src = l["weightSourceLayerIndices"][i]
            seq.append(owner_of[(li, "weights[%d]" % i)] if src < 0 else owner_of[(src, "weights[%d]" % l["weightSourceMatrixIndices"][i])])
        for t in seq:
            contributors[t] += 1
        arrivals[li] = seq
    return names, ends, contributors, arrivals


def _plan(tmp_path, min_elems, ends, contributors, sequence):
    exe = str(tmp_path / "bucket_planner")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(HERE, "..", "cuda-convnet_amd", "csrc", "runtime"),
                               os.path.join(HERE, "bucket_planner_main.cpp"), "-o", exe])
    text = "%d\n%s\n%s\n%s\n" % (min_elems, " ".join(map(str, ends)), " ".join(map(str, contributors)), " ".join(map(str, sequence)))
    out = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("end\n")
    steps = [[tuple(int(x) for x in line.split()) for line in part.splitlines()] for part in out[:2]]
    assert steps[0] == steps[1]                                # the planner resets completely between steps
    return steps[0]


def _check_cover(buckets, total):
    assert buckets and buckets[0][0] == 0 and buckets[-1][1] == total
    for (lo, hi), (lo2, _) in zip(buckets, buckets[1:] + [(total, total)]):
        assert lo < hi and hi == lo2                            # disjoint, in order, no gap: every element reduced exactly once


def test_bucket_boundaries_imagenet(tmp_path):
    """The imagenet net's gradient arena (about 61 M floats) in back-propagation order: buckets start and end on tensor
    boundaries, every bucket but the last holds at least 256 Ki floats, the three fc layers' matrices go out as soon as they
    are complete (long before the conv stack has run), and the whole arena is covered exactly once."""
    layers, _, _ = modelutil.load_net("imagenet")
    names, ends, contributors, arrivals = _arena(layers)
    sequence = [t for li in sorted(arrivals, reverse=True) for t in arrivals[li]]
    min_elems = 256 * 1024
    buckets = _plan(tmp_path, min_elems, ends, contributors, sequence)
    _check_cover(buckets, ends[-1])
    assert all(hi in ends for _, hi in buckets)
    assert all(hi - lo >= min_elems for lo, hi in buckets[:-1])
    # the first bucket closes at the first tensor boundary with >= 256 Ki floats behind it: the top fc layers, whose
    # gradients exist long before the conv stack has run
    assert names[0].endswith("biases") and buckets[0] == (0, next(e for e in ends if e >= min_elems))
    assert names[ends.index(buckets[0][1])].startswith("fc") and len(buckets) >= 5
    # the same tensors arriving in any other order still cover the arena exactly once
    rng = np.random.default_rng(0)
    shuffled = list(rng.permutation(sequence))
    _check_cover(_plan(tmp_path, min_elems, ends, contributors, shuffled), ends[-1])


def test_buckets_with_shared_weights_wait_for_every_contributor(tmp_path):
    """Three 100-element tensors; tensor 0 is shared by two layers (2 contributors), tensor 2 does not learn.  Tensor 1 completes
    first but cannot go before tensor 0 (buckets are contiguous stretches from the frontier); once the second contribution to
    tensor 0 arrives, 0 and 1 go together, and tensor 2 (nobody contributes) rides along."""
    ends, contributors = [128, 256, 384], [2, 1, 0]
    assert _plan(tmp_path, 1, ends, contributors, [0, 1, 0]) == [(0, 384)]
    assert _plan(tmp_path, 1, ends, contributors, [0, 0, 1]) == [(0, 128), (128, 384)]
    assert _plan(tmp_path, 1, ends, contributors, [1]) == [(0, 384)]                 # tensor 0 never completes: the final flush sends all
    assert _plan(tmp_path, 200, ends, contributors, [0, 0, 1]) == [(0, 384)]         # 128 < 200: held until the frontier is far enough
    # a net that really shares matrices: the counts come out of the layer dicts
    layers, _, _ = modelutil.load_net("example-small")
    names, ends, contributors, arrivals = _arena(layers)
    sequence = [t for li in sorted(arrivals, reverse=True) for t in arrivals[li]]
    _check_cover(_plan(tmp_path, 64, ends, contributors, sequence), ends[-1])
