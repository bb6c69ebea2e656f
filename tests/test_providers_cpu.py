"""Data providers on py2-pickled batches (the format real CIFAR / MNIST batch directories have): the `mnist` provider
(reference Python/convdata.py:60-87), the fork's `cifar-cropped` provider (Python/convdata.py:90-153 -- whose training branch
cannot run as written, so what is checked here is the geometry it sets out to produce: parity unpinned), the batch cursor,
sub-batch files and the provider registry."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cuda-convnet_amd"))
sys.path.insert(0, os.path.dirname(__file__))
import convdata  # noqa: E402
import convnet  # noqa: E402
import data as data_mod  # noqa: E402
import modelutil  # noqa: E402
from py2pickle import dumps  # noqa: E402


def write_batches(tmp_path, dims, cases_per_batch, classes=10, seed=0, maxval=256):
    rng = np.random.default_rng(seed)
    for i, cases in enumerate(cases_per_batch, start=1):
        batch = {"data": rng.integers(0, maxval, size=(dims, cases), dtype=np.uint8), "labels": [int(x) for x in rng.integers(0, classes, cases)]}
        (tmp_path / ("data_batch_%d" % i)).write_bytes(dumps(batch))
    mean = (rng.random((dims, 1)) * 255).astype(np.float32)
    meta = {"data_mean": mean, "label_names": ["c%d" % k for k in range(classes)], "num_vis": dims}
    (tmp_path / "batches.meta").write_bytes(dumps(meta))
    return mean


def raw_batch(tmp_path, i):
    import util
    return util.unpickle(str(tmp_path / ("data_batch_%d" % i)))


def test_mnist_provider_is_registered_and_mean_subtracts(tmp_path):
    mean = write_batches(tmp_path, 784, [30, 30])
    convnet.ConvNet.get_options_parser()             # registers the ConvNet providers
    dp = data_mod.DataProvider.get_instance(str(tmp_path), [1, 2], type="mnist", dp_params={}, test=False)
    assert isinstance(dp, convdata.MnistDataProvider)
    assert dp.get_data_dims() == 784 and dp.get_data_dims(1) == 1 and dp.get_num_classes() == 10
    epoch, batchnum, (x, y) = dp.get_next_batch()
    raw = raw_batch(tmp_path, 1)
    assert (epoch, batchnum) == (1, 1) and x.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and y.shape == (1, 30)
    np.testing.assert_array_equal(x, raw["data"].astype(np.float32) - mean)
    np.testing.assert_array_equal(y[0], np.asarray(raw["labels"], dtype=np.float32))
    pics = dp.get_plottable_data(x)
    assert pics.shape == (30, 28, 28, 1)
    np.testing.assert_allclose(pics[3, 5, 7, 0], raw["data"][5 * 28 + 7, 3] / 255.0, rtol=1e-5)
    # the data layer of the shipped MNIST NIN definition takes exactly this provider's dimensionality
    layers, dims, classes = modelutil.load_net("nin-mnist")
    assert layers[0]["type"] == "data" and layers[0]["outputs"] == dp.get_data_dims() == dims


def cropped(tmp_path, test, multiview=False, border=4, cases=(12, 12)):
    write_batches(tmp_path, 3072, list(cases))
    return convdata.CroppedCIFARDataProvider(str(tmp_path), list(range(1, len(cases) + 1)), dp_params={"crop_border": border, "multiview_test": multiview}, test=test)


def test_cropped_cifar_dims_match_the_nin_data_layer(tmp_path):
    dp = cropped(tmp_path, test=False)
    layers, dims, _ = modelutil.load_net("nin")
    assert dp.get_data_dims() == 3072 == layers[0]["outputs"] == dims and dp.get_data_dims(1) == 1


def test_cropped_cifar_test_batches_are_the_images_themselves(tmp_path):
    dp = cropped(tmp_path, test=True)
    _, _, (x, y) = dp.get_next_batch()
    raw = raw_batch(tmp_path, 1)
    assert x.dtype == np.float32 and x.shape == (3072, 12) and y.shape == (1, 12) and y.dtype == np.float32
    np.testing.assert_array_equal(x, raw["data"].astype(np.float32))        # nothing subtracted, nothing cut
    np.testing.assert_array_equal(y[0], np.asarray(raw["labels"], dtype=np.float32))


def test_cropped_cifar_training_windows(tmp_path):
    """Every training case is a 32x32 window of the zero-padded image, possibly mirrored: it must equal the padded image at
    one of the (2b+1)^2 offsets x 2 orientations, and over a batch more than one offset must occur."""
    b = 3
    dp = cropped(tmp_path, test=False, border=b)
    np.random.seed(11)
    _, _, (x, _) = dp.get_next_batch()
    raw = raw_batch(tmp_path, 1)["data"].astype(np.float32).reshape(3, 32, 32, 12)
    padded = np.pad(raw, ((0, 0), (b, b), (b, b), (0, 0)))
    seen = set()
    for c in range(12):
        pic = x[:, c].reshape(3, 32, 32)
        hits = [(sy, sx, flip) for sy in range(2 * b + 1) for sx in range(2 * b + 1) for flip in (0, 1)
                if np.array_equal(pic, padded[:, sy:sy + 32, sx:sx + 32, c][:, :, ::-1] if flip else padded[:, sy:sy + 32, sx:sx + 32, c])]
        assert hits, "case %d is not a window of its padded image" % c
        seen.add(hits[0])
    assert len(seen) > 3
    # two output buffers in turn: the previous batch's array is still intact after the next one is cut
    before = x.copy()
    _, _, (x2, _) = dp.get_next_batch()
    assert x2 is not x and np.array_equal(x, before)


def test_cropped_cifar_multiview_layout(tmp_path):
    """--multiview-test: 10 views side by side, view v of case c in column v*cases + c (what startMultiviewTest averages,
    Kernel/src/worker.cu:142-174); views 5..9 are the mirrors of 0..4; the centre view is the image."""
    b, cases = 4, 12
    dp = cropped(tmp_path, test=True, multiview=True, border=b)
    assert dp.num_views == 10
    _, _, (x, y) = dp.get_next_batch()
    raw = raw_batch(tmp_path, 1)
    img = raw["data"].astype(np.float32).reshape(3, 32, 32, cases)
    assert x.shape == (3072, 10 * cases) and y.shape == (1, 10 * cases)
    np.testing.assert_array_equal(y[0], np.tile(np.asarray(raw["labels"], dtype=np.float32), 10))
    views = x.reshape(3, 32, 32, 10, cases)
    np.testing.assert_array_equal(views[:, :, :, 2, :], img)                                   # centre
    np.testing.assert_array_equal(views[:, b:, b:, 0, :], img[:, :32 - b, :32 - b, :])          # top-left window of the padded image
    assert not views[:, :b, :, 0, :].any() and not views[:, :, :b, 0, :].any()
    for v in range(5):
        np.testing.assert_array_equal(views[:, :, :, 5 + v, :], views[:, :, ::-1, v, :])


def test_cursor_wraps_and_counts_epochs(tmp_path):
    write_batches(tmp_path, 48, [5, 5, 5])
    dp = data_mod.LabeledMemoryDataProvider(str(tmp_path), [1, 2, 3], init_epoch=4, init_batchnum=3)
    assert (dp.curr_epoch, dp.curr_batchnum, dp.batch_idx, dp.get_next_batch_num()) == (4, 3, 2, 1)
    assert dp.get_next_batch()[:2] == (4, 3) and dp.get_next_batch()[:2] == (5, 1) and dp.get_next_batch()[:2] == (5, 2)
    assert data_mod.DataProvider.get_batch_nums(str(tmp_path)) == [1, 2, 3] and data_mod.DataProvider.get_num_batches(str(tmp_path)) == 3
    dp = data_mod.DataProvider(str(tmp_path), None, init_batchnum=99)            # unknown start batch: the first one
    assert dp.curr_batchnum == 1 and dp.batch_range == [1, 2, 3]


def test_sub_batch_files_are_stacked(tmp_path):
    rng = np.random.default_rng(3)
    parts = [rng.random((4, 6)).astype(np.float32) for _ in range(3)]
    for k, p in enumerate(parts, start=1):
        (tmp_path / ("data_batch_7.%d" % k)).write_bytes(dumps({"data": p}))
    (tmp_path / "batches.meta").write_bytes(dumps({"num_vis": 6}))
    dp = data_mod.DataProvider(str(tmp_path), [7])
    assert data_mod.DataProvider.get_batch_nums(str(tmp_path)) == [7]
    np.testing.assert_array_equal(dp.get_next_batch()[2]["data"], np.concatenate(parts, axis=0))


def test_registry_and_dummy_names():
    convnet.ConvNet.get_options_parser()
    for name in ("cifar", "mnist", "cifar-cropped", "dummy-cn-n", "dummy-synth-n", "default", "memory", "labeled", "labeled-memory"):
        assert name in data_mod.dp_types and name in data_mod.dp_classes
    dp = data_mod.DataProvider.get_instance("", type="dummy-cn-192")
    assert isinstance(dp, convdata.DummyConvNetDataProvider) and dp.get_data_dims() == 192 and dp.get_num_classes() == 10
    np.random.seed(0)
    epoch, batchnum, (x, y) = dp.get_next_batch()
    assert x.shape == (192, 512) and y.shape == (1, 512) and x.dtype == y.dtype == np.float32 and (epoch, batchnum) == (2, 1)
    for bad in ("dummy-nothing-12", "nothere", "dummy-cn-n"):
        with pytest.raises(data_mod.DataProviderException):
            data_mod.DataProvider.get_instance("", type=bad)
    with pytest.raises(data_mod.DataProviderException):
        data_mod.DataProvider.register_data_provider("cifar", "again", data_mod.DataProvider)
