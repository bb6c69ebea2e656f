"""the reference's checkpoints and data batches are Python-2 cPickle files
(gpumodel.py:272-281 -- {"model_state": {...}, "op": OptionsParser}; data.py:254-273 -- {"data", "labels"};
batches.meta).  util.unpickle must read them (8-bit str payloads of NumPy arrays, old-style class
instances of options.Option / OptionsParser, optional gzip) and the driver must resume from one.
tests/py2pickle.py emits the cPickle byte stream; no file of the reference is involved."""
import gzip
import os

import numpy as np
import pytest

import modelutil
import options
import util
from py2pickle import Cls, Inst, dumps


def _opt(letter, name, parser, value, default=None):
    return Inst("options", "Option", {"letter": letter, "name": name, "desc": name, "parser": parser, "set_once": False,
                                      "default": default, "excuses": [], "requires": [], "save": True, "value": value,
                                      "value_given": True, "prefixed_letter": min(2, len(letter)) * "-" + letter})


def _checkpoint(layers):
    op = Inst("options", "OptionsParser", {"options": {
        "minibatch_size": _opt("mini", "minibatch_size", Cls("options", "IntegerOptionParser"), 128, default=128),
        "train_batch_range": _opt("train-range", "train_batch_range", Cls("options", "RangeOptionParser"), [1, 2, 3]),
        "layer_def": _opt("layer-def", "layer_def", Cls("options", "StringOptionParser"), "./example-layers/layers-80sec.cfg"),
        "scalar_mean": _opt("scalar-mean", "scalar_mean", Cls("options", "FloatOptionParser"), -1.0)}})
    return {"model_state": {"layers": layers, "epoch": 3, "batchnum": 2, "train_outputs": [({"logprob": [291.3, 110.0]}, 128)]}, "op": op}


def test_python2_checkpoint_loads(tmp_path):
    layers, _, _ = modelutil.load_net("gc-small", seed=5)
    slim = []
    for l in layers:            # the value kinds a layer dict holds: str, int, float, bool, lists, C- and F-ordered float32 arrays
        d = {k: v for k, v in l.items() if isinstance(v, (str, int, float, bool, np.ndarray)) or
             (isinstance(v, list) and all(isinstance(x, (str, int, float, bool, np.ndarray)) for x in v))}
        slim.append(d)
    blob = dumps(_checkpoint(slim))
    p = tmp_path / "3.2"
    p.write_bytes(blob)
    got = util.unpickle(str(p))
    assert set(got) == {"model_state", "op"} and got["model_state"]["epoch"] == 3
    assert got["model_state"]["train_outputs"][0][0]["logprob"] == [291.3, 110.0]
    for a, b in zip(slim, got["model_state"]["layers"]):
        assert sorted(a) == sorted(b)
        for k, v in a.items():
            if isinstance(v, np.ndarray):
                np.testing.assert_array_equal(v, b[k])
                assert b[k].dtype == np.float32 and b[k].flags["F_CONTIGUOUS"] == v.flags["F_CONTIGUOUS"], (a["name"], k)
            elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
                for x, y in zip(v, b[k]):
                    np.testing.assert_array_equal(x, y)
                    assert y.flags["F_CONTIGUOUS"] == x.flags["F_CONTIGUOUS"] and y.flags["WRITEABLE"]
            else:
                assert v == b[k], (a["name"], k)
    op = got["op"]
    assert isinstance(op, options.OptionsParser) and isinstance(op.options["minibatch_size"], options.Option)
    assert op.get_value("minibatch_size") == 128 and op.get_value("train_batch_range") == [1, 2, 3]
    assert op.options["minibatch_size"].parser is options.IntegerOptionParser
    assert op.options["layer_def"].get_str_value() == "./example-layers/layers-80sec.cfg"
    # weights read from such a file must be writable in place: syncWithHost copies the device state into them
    w = [l for l in got["model_state"]["layers"] if "weights" in l][0]["weights"][0]
    w += 1


def test_python2_checkpoint_gzip_and_directory(tmp_path):
    """--zip-save checkpoints (util.py:33-37 of the reference) and `-f <dir>` = newest file by natural order
    (gpumodel.py:283-289: 9.5 sorts before 10.1)."""
    import gpumodel
    d = tmp_path / "ConvNet__x"
    d.mkdir()
    for name, epoch in (("9.5", 9), ("10.1", 10)):
        ck = _checkpoint([])
        ck["model_state"]["epoch"] = epoch
        with gzip.open(str(d / name), "wb") as f:
            f.write(dumps(ck))
    assert util.unpickle(str(d / "9.5"))["model_state"]["epoch"] == 9
    assert gpumodel.IGPUModel.load_checkpoint(str(d))["model_state"]["epoch"] == 10
    with pytest.raises(util.UnpickleError):
        util.unpickle(str(d / "nothere"))


def test_python2_cifar_batch_and_meta(tmp_path):
    """data_batch_N = {"data": uint8 (3072, cases), "labels": [int]}, batches.meta = {"data_mean": float32 (3072, 1),
    "label_names": [...], "num_vis": 3072} (convdata.py:31-46): read through the CIFAR provider."""
    import convdata
    rng = np.random.default_rng(0)
    cases = 40
    for i in (1, 2):
        batch = {"data": rng.integers(0, 256, size=(3072, cases), dtype=np.uint8), "labels": [int(x) for x in rng.integers(0, 10, cases)],
                 "batch_label": "training batch %d of 5" % i, "filenames": ["img_%d.png" % k for k in range(cases)]}
        (tmp_path / ("data_batch_%d" % i)).write_bytes(dumps(batch))
    mean = (rng.random((3072, 1)) * 255).astype(np.float32)
    meta = {"data_mean": mean, "label_names": ["c%d" % k for k in range(10)], "num_vis": 3072, "num_cases_per_batch": cases}
    (tmp_path / "batches.meta").write_bytes(dumps(meta))
    dp = convdata.CIFARDataProvider(str(tmp_path), [1, 2], init_epoch=1, init_batchnum=1, dp_params={}, test=False)
    epoch, batchnum, (data, labels) = dp.get_next_batch()
    assert (epoch, batchnum) == (1, 1) and data.dtype == np.float32 and data.shape == (3072, cases) and labels.shape == (1, cases)
    raw = util.unpickle(str(tmp_path / "data_batch_1"))
    np.testing.assert_allclose(data, raw["data"].astype(np.float32) - mean, rtol=0, atol=0)
    np.testing.assert_array_equal(labels[0], np.array(raw["labels"], dtype=np.float32))
    assert dp.get_num_classes() == 10 and dp.get_data_dims() == 3072 and dp.get_data_dims(1) == 1
    assert dp.get_next_batch()[1] == 2 and dp.get_next_batch()[:2] == (2, 1)     # wraps into the next epoch
