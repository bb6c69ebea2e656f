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
    assert len(both) == 300 - (300 % 128) % 2 and len(set(both)) == len(both)
    # rank r's slice of global minibatch i is contiguous and of equal size
    assert list(cols[0][:64]) == list(range(0, 64)) and list(cols[1][:64]) == list(range(64, 128))
    assert list(cols[0][128:]) == list(range(256, 278)) and list(cols[1][128:]) == list(range(278, 300))


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
    data = modelutil.synthetic_batch(dims, classes, 40)
    net = GoldenNet(layers)
    for s in range(0, 40, 16):
        net.train_step([d[:, s:s + 16] for d in data])
    for l in layers:
        if "weights" in l:
            np.testing.assert_allclose(r0["W:" + l["name"]], net.weight(l["name"], 0), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(r0["B:" + l["name"]], net.B[l["name"]], rtol=1e-4, atol=1e-6)
