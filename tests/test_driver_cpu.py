"""BASELINE config 1 (CPU plumbing): the driver's option parser + CIFAR data provider + layer parser on
layers-80sec, and the first conv(5x5, 3->32, pad 2) + max-pool(3/2) + relu block forward / backward on
the C/NumPy oracle.  No GPU."""
import os
import pickle

import numpy as np
import pytest

import modelutil
from oracle import golden as G


def make_cifar_dir(tmp_path, batches=2, cases=96):
    rng = np.random.default_rng(0)
    for b in range(1, batches + 1):
        d = {"data": rng.integers(0, 256, size=(3072, cases)).astype(np.float32), "labels": [int(x) for x in rng.integers(0, 10, size=cases)]}
        with open(tmp_path / ("data_batch_%d" % b), "wb") as f:
            pickle.dump(d, f)
    meta = {"data_mean": np.full((3072, 1), 120.0, np.float32), "label_names": [str(i) for i in range(10)], "num_vis": 3072, "num_cases_per_batch": cases}
    with open(tmp_path / "batches.meta", "wb") as f:
        pickle.dump(meta, f)
    return str(tmp_path)


def test_options_provider_and_parser(tmp_path):
    import convnet
    from gpumodel import IGPUModel
    from data import DataProvider
    data_dir = make_cifar_dir(tmp_path)
    op = convnet.ConvNet.get_options_parser()
    argv = ["--data-path=" + data_dir, "--save-path=" + str(tmp_path), "--test-range=2", "--train-range=1", "--data-provider=cifar",
            "--layer-def=" + os.path.join(modelutil.CONFIGS, "layers-80sec.cfg"), "--layer-params=" + os.path.join(modelutil.CONFIGS, "layer-params-80sec.cfg"),
            "--test-freq=13", "--mini=64"]
    op, load_dic = IGPUModel.parse_options(op, argv=argv)
    assert load_dic is None and op.get_value("minibatch_size") == 64 and op.get_value("train_batch_range") == [1]
    assert op.get_value("gpu") == [-1] and op.get_value("testing_freq") == 13

    dp = DataProvider.get_instance(data_dir, [1], type="cifar", dp_params={}, test=False)
    epoch, batchnum, (data, labels) = dp.get_next_batch()
    assert (epoch, batchnum) == (1, 1) and data.shape == (3072, 96) and data.dtype == np.float32 and labels.shape == (1, 96)
    assert data.flags["C_CONTIGUOUS"] and abs(float(data.mean())) < 10      # mean-subtracted
    assert dp.get_data_dims() == 3072 and dp.get_data_dims(1) == 1 and dp.get_num_classes() == 10
    assert dp.get_next_batch()[0] == 2                                        # single-batch range wraps: next epoch

    m = modelutil._Model()
    m.train_data_provider = dp
    import layer
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()) as out:
        layers = layer.LayerParser.parse_layers(op.get_value("layer_def"), op.get_value("layer_params"), m)
    assert "Initialized convolutional layer 'conv1', producing 32x32 32-channel output" in out.getvalue()
    assert [l["name"] for l in layers][:5] == ["data", "labels", "conv1", "pool1", "pool1_neuron"]


def test_conv_pool_relu_block_on_the_oracle():
    """fprop and bprop of layers-80sec's first block; the weight gradient is checked by finite differences."""
    rng = np.random.default_rng(1)
    N = 8
    x = rng.random((3072, N), dtype=np.float32)
    w = (0.05 * rng.standard_normal((75, 32))).astype(np.float32)
    b = np.zeros((32, 1), np.float32)
    geo = (N, 3, 32, 32, 32, 5, -2, 1, 32, 32, 1)

    def forward(w):
        c = G.filter_acts(x, w, *geo)
        c = (c.reshape(32, -1) + b).reshape(c.shape).astype(np.float32)
        p = G.pool(c, N, 32, 32, 3, 0, 2, 16, "max")
        return c, p, G.neuron_fwd("relu", {}, p)

    c, p, y = forward(w)
    assert c.shape == (32 * 1024, N) and p.shape == (32 * 256, N) and (y >= 0).all()
    g = rng.standard_normal(y.shape).astype(np.float32)
    gp = G.neuron_bwd("relu", {}, g, y, p)
    gc = G.max_undo(c, gp, p, N, 32, 32, 3, 0, 2, 16)
    dw = G.weight_acts(x, gc, *geo, partial_sum=0)
    dx = G.img_acts(gc, w, *geo)
    assert dw.shape == w.shape and dx.shape == x.shape
    loss = lambda w_: float((forward(w_)[2].astype(np.float64) * g).sum())
    base = loss(w)
    for idx in [(0, 0), (37, 5), (74, 31)]:
        w2 = w.copy()
        w2[idx] += 1e-3
        num = (loss(w2) - base) / 1e-3
        assert abs(num - dw[idx]) < 0.05 * max(1.0, abs(dw[idx])), (idx, num, dw[idx])
