"""Config front end: layer-def / layer-params files -> layer dicts (Python/layer.py of the reference).
The shipped reference files are parsed unchanged when the reference tree is present and must give
the same model as this repo's own copies of the BASELINE networks."""
import os

import numpy as np
import pytest

import modelutil

REF = "/root/reference"


def test_nin_shapes():
    layers, dims, classes = modelutil.load_net("nin")
    by = dict((l["name"], l) for l in layers)
    assert [l["name"] for l in layers][:6] == ["data", "labels", "conv1", "conv1_neuron", "cccp1", "cccp1_neuron"]
    assert by["conv1"]["outputs"] == 192 * 32 * 32 and by["conv1"]["modulesX"] == 32 and by["conv1"]["padding"] == [-2]
    assert by["conv1"]["weights"][0].shape == (75, 192) and by["conv1"]["weights"][0].flags["C_CONTIGUOUS"]
    assert by["cccp1"]["weights"][0].shape == (192, 160) and by["cccp1"]["weights"][0].flags["F_CONTIGUOUS"]
    assert by["cccp1"]["biases"].shape == (1, 160)
    assert by["pool1"]["outputsX"] == 16 and by["pool2"]["outputsX"] == 8 and by["poolg"]["outputsX"] == 1
    assert by["dropout1"]["neuron"] == {"type": "dropout", "params": {"prob": 0.5}, "usesActs": False, "usesInputs": False}
    # in-place relu after every weight layer; dropout shares only the gradient (pool "uses" its acts)
    assert by["conv1_neuron"]["actsTarget"] == 0 and by["conv1_neuron"]["actsGradTarget"] == 0
    assert by["dropout1"]["actsTarget"] == -1 and by["dropout1"]["actsGradTarget"] == 0
    assert by["logprob"]["inputs"] == [1, 25] and by["logprob"]["coeff"] == 0.1
    n = sum(w.size for l in layers if "weights" in l for w in l["weights"] + [l["biases"]])
    assert n == 966986
    assert by["cccp6"]["epsW"] == [0.1] and by["conv1"]["wc"] == [0.00001]


@pytest.mark.parametrize("name,params", [("80sec", 145578), ("conv-local", 1901386), ("imagenet", 50351076), ("18pct", 89578)])
def test_parameter_counts(name, params):
    layers, _, _ = modelutil.load_net(name)
    assert sum(w.size for l in layers if "weights" in l for w in l["weights"] + [l["biases"]]) == params


def test_norm_scale_and_default_k():
    layers, _, _ = modelutil.load_net("imagenet")
    r = [l for l in layers if l["name"] == "rnorm1"][0]
    assert r["type"] == "cmrnorm" and abs(r["scale"] - 0.0001 / 5) < 1e-12 and r["k"] == 1.0 and r["pow"] == 0.75
    layers, _, _ = modelutil.load_net("18pct")
    r = [l for l in layers if l["name"] == "rnorm1"][0]
    assert abs(r["scale"] - 0.00005 / 9) < 1e-15


def test_neuron_strings():
    import layer
    def parse(s):
        for p in layer.neuron_parsers:
            r = p.parse(s)
            if r:
                return r
    assert parse("tanh[1.79,-0.66]")["params"] == {"a": 1.79, "b": -0.66}
    assert parse("relu")["type"] == "relu" and parse("brelu[6]")["params"] == {"a": 6.0}
    assert parse("dropout[0.5]")["type"] == "dropout"
    assert parse("nosuch") is None and parse("tanh[1]") is None


def test_parse_errors_exit(tmp_path, capsys):
    import layer
    d = tmp_path / "bad.def"
    d.write_text("[data]\ntype=data\ndataIdx=0\n[c]\ntype=conv\ninputs=nothere\n")
    p = tmp_path / "bad.params"
    p.write_text("")
    m = modelutil._Model()
    m.train_data_provider = modelutil.ShapeProvider(3072, 10)
    with pytest.raises(SystemExit):
        layer.LayerParser.parse_layers(str(d), str(p), m)
    assert "input layer 'nothere' not defined" in capsys.readouterr().out


REF_PAIRS = [("nin", "NIN/cifar-10_def", "NIN/cifar-10_params"), ("80sec", "example-layers/layers-80sec.cfg", "example-layers/layer-params-80sec.cfg"),
             ("conv-local", "example-layers/layers-conv-local-13pct.cfg", "example-layers/layer-params-conv-local-13pct.cfg"),
             ("imagenet", "example-layers/imagenet", "example-layers/imagenet_params"),
             ("18pct", "example-layers/layers-18pct.cfg", "example-layers/layer-params-18pct.cfg"),
             ("19pct", "example-layers/layers-19pct.cfg", "example-layers/layer-params-19pct.cfg"),
             ("nin-cifar100", "NIN/cifar-100_def", "NIN/cifar-100_params"), ("nin-mnist", "NIN/mnist_def", "NIN/mnist_params"),
             ("nin-svhn", "NIN/svhn_def", "NIN/svhn_params")]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("name,rdef,rparams", REF_PAIRS)
def test_reference_config_files_parse_unchanged(name, rdef, rparams):
    mine, dims, classes = modelutil.load_net(name, seed=3)
    theirs = modelutil.parse_net(os.path.join(REF, rdef), os.path.join(REF, rparams), dims, classes, seed=3)
    assert [l["name"] for l in mine] == [l["name"] for l in theirs]
    for a, b in zip(mine, theirs):
        assert sorted(a.keys()) == sorted(b.keys()), a["name"]
        for k in a:
            if k in ("inputLayers",):
                continue
            if isinstance(a[k], np.ndarray):
                np.testing.assert_array_equal(a[k], b[k])
            elif k in ("weights", "weightsInc"):
                for x, y in zip(a[k], b[k]):
                    np.testing.assert_array_equal(x, y)
            elif k == "partialSum":
                continue      # a tuning knob of the reference's kernels; this runtime picks its own split
            else:
                assert a[k] == b[k], (a["name"], k, a[k], b[k])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_gc_config_parses():
    layers = modelutil.parse_net(os.path.join(REF, "example-layers/layers.gc.cfg"), os.path.join(REF, "example-layers/layer-params.gc.cfg"), 3072, 10)
    by = dict((l["name"], l) for l in layers)
    assert by["local32"]["randSparse"] == [True] and len(by["local32"]["filterConns"][0]) == by["local32"]["groups"][0] * by["local32"]["filterChannels"][0]
    assert by["avgpool"]["start"] == -2 and by["fc10"]["inputs"] == [6, 3] or by["fc10"]["numInputs"] == [by["local32"]["outputs"], by["conv32"]["outputs"]]


def test_reference_example_config_is_rejected_as_the_reference_rejects_it(tmp_path, capsys):
    """example-layers/layers-example.cfg + layer-params-example.cfg (verbatim copies under configs/) are stale in the fork: the
    norm sections say `sizeX` where NormLayerParser reads `size` (Python/layer.py:1089), and with that renamed the params file
    has no [rnorm1] / [cnorm1] section although norm layers take scale / pow from it (layer.py:276-284, 1078-1083).  Both are
    LayerParsingErrors in the reference (print + exit(1), layer.py:285-287); same messages, same order here."""
    d, p = os.path.join(modelutil.CONFIGS, "layers-example.cfg"), os.path.join(modelutil.CONFIGS, "layer-params-example.cfg")
    with pytest.raises(SystemExit):
        modelutil.parse_net(d, p, 3072, 10, quiet=False)
    assert "Layer 'rnorm1': required parameter 'size' missing" in capsys.readouterr().out
    fixed, section = [], ""
    for line in open(d).read().splitlines():
        section = line.strip() if line.startswith("[") else section
        fixed.append("size=" + line[6:] if section in ("[rnorm1]", "[cnorm1]") and line.startswith("sizeX=") else line)
    d2 = tmp_path / "layers-example-size.cfg"
    d2.write_text("\n".join(fixed) + "\n")
    with pytest.raises(SystemExit):
        modelutil.parse_net(str(d2), p, 3072, 10, quiet=False)
    assert "Layer 'rnorm1' of type 'rnorm' requires extra parameters, but none given in file" in capsys.readouterr().out


def test_shipped_gc_fixture_is_the_reference_file():
    """configs/layers.gc.cfg and layer-params.gc.cfg are byte copies of the reference's fixture (checked where the reference
    tree is present); the net they describe runs through --check-grads and as a whole-step case on the GPU."""
    layers, dims, classes = modelutil.load_net("gc")
    by = dict((l["name"], l) for l in layers)
    assert by["local32"]["randSparse"] == [True] and by["local32"]["groups"] == [2] and by["avgpool"]["start"] == -2
    assert by["conv32_neuron"]["neuron"]["params"] == {"a": 3.0, "b": 2.2} and by["local32_neuron"]["neuron"]["type"] == "tanh"
    assert len(by["fc10"]["weights"]) == 2
    if os.path.isdir(REF):
        for ours, theirs in (("layers.gc.cfg", "example-layers/layers.gc.cfg"), ("layer-params.gc.cfg", "example-layers/layer-params.gc.cfg"),
                             ("layers-example.cfg", "example-layers/layers-example.cfg"), ("layer-params-example.cfg", "example-layers/layer-params-example.cfg")):
            assert open(os.path.join(modelutil.CONFIGS, ours), "rb").read() == open(os.path.join(REF, theirs), "rb").read()


def test_conv_to_local_is_the_same_function():
    """--conv-to-local (layer.py:816-844): the filter bank tiled once per module.  The converted net computes the
    same forward pass, and after one training step the per-module weight increments of the local layer add up
    to the shared layer's increment (the shared gradient is the sum over modules, weight_acts.cu:608-681)."""
    import layer
    from oracle.golden_net import GoldenNet
    conv, dims, classes = modelutil.load_net("80sec", seed=11)
    loc, _, _ = modelutil.load_net("80sec", seed=11)
    idx = [l["name"] for l in loc].index("conv2")
    layer.LocalLayerParser.conv_to_local(loc, idx)
    c, l = conv[idx], loc[idx]
    assert l["type"] == "local" and l["weights"][0].shape == (c["modules"] * c["weights"][0].shape[0], c["filters"])
    assert l["biases"].shape == (c["modules"] * c["filters"], 1) and l["weights"][0].flags["C_CONTIGUOUS"]
    np.testing.assert_array_equal(l["weights"][0].reshape(c["modules"], -1, c["filters"])[7], c["weights"][0])
    for d in (c, l):
        d["wc"], d["momW"] = [0.0], [0.0]          # increments = eps * gradient: comparable between the two
    data = modelutil.synthetic_batch(dims, classes, 16)
    a, b = GoldenNet(conv), GoldenNet(loc)
    ca, cb = a.train_step(data), b.train_step(data)
    np.testing.assert_allclose(ca["logprob"], cb["logprob"], rtol=1e-6)
    summed = b.Winc[("conv2", 0)].reshape(c["modules"], -1, c["filters"]).astype(np.float64).sum(axis=0)
    ref = a.Winc[("conv2", 0)]
    assert np.abs(summed - ref).max() < 1e-4 * np.sqrt((ref.astype(np.float64) ** 2).mean())
    bsum = b.Binc["conv2"].reshape(c["filters"], c["modules"]).astype(np.float64).sum(axis=1)
    assert np.abs(bsum - a.Binc["conv2"].reshape(-1)).max() < 1e-4 * np.abs(a.Binc["conv2"]).max()


SHARED = """[data]
type=data
dataIdx=0
[labels]
type=data
dataIdx=1
[fca]
type=fc
inputs=data
outputs=12
initW=0.1
[fcb]
type=fc
inputs=data,fca
outputs=12
initW=0.1,0.1
weightSource=fca,
[probs]
type=softmax
inputs=fcb
[logprob]
type=cost.logreg
inputs=labels,probs
"""
SHARED_PARAMS = "".join("[%s]\nepsW=%s\nepsB=0.01\nmomW=%s\nmomB=0\nwc=%s\n" % (nm, v, v0, v0)
                        for nm, v, v0 in (("fca", "0.01", "0"), ("fcb", "0.01,0.01", "0,0"))) + "[logprob]\ncoeff=1\n"


def test_weight_sharing_and_unshare(tmp_path, capsys):
    """weightSource makes two layers hold the SAME ndarray (layer.py:693-703); --unshare-weights gives the
    borrower its own copy and clears the source indices (layer.py:653-669)."""
    import layer
    (tmp_path / "s.def").write_text(SHARED)
    (tmp_path / "s.params").write_text(SHARED_PARAMS)
    layers = modelutil.parse_net(str(tmp_path / "s.def"), str(tmp_path / "s.params"), 20, 12)
    by = dict((l["name"], l) for l in layers)
    assert by["fcb"]["weights"][0] is by["fca"]["weights"][0] and by["fcb"]["weightsInc"][0] is by["fca"]["weightsInc"][0]
    assert by["fcb"]["weightSourceLayerIndices"] == [2, -1] and by["fcb"]["weightSourceMatrixIndices"] == [0, -1]
    layer.WeightLayerParser.unshare_weights(by["fcb"], layers, matrix_idx=0)
    assert by["fcb"]["weights"][0] is not by["fca"]["weights"][0]
    np.testing.assert_array_equal(by["fcb"]["weights"][0], by["fca"]["weights"][0])
    assert by["fcb"]["weightSourceLayerIndices"] == [-1, -1] and not by["fcb"]["weightsInc"][0].any()
    layer.WeightLayerParser.unshare_weights(by["fcb"], layers)
    out = capsys.readouterr().out
    assert "Unshared weight matrix fcb[0] from fca[0]." in out and "Weight matrix fcb[0] already unshared." in out
