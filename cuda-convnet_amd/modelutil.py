"""Helpers shared by bench.py, __graft_entry__.py and the tests: parse a layer-def / layer-params
pair into the layer-dict list without going through the command-line driver."""
import contextlib
import io
import os

import numpy as n
import numpy.random as nr

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = os.path.join(HERE, "configs")

NETS = {   # name -> (layer-def, layer-params, data dims, classes)
    "nin": ("nin-cifar10.def", "nin-cifar10.params", 3072, 10),
    "nin-tiny": ("nin-tiny.def", "nin-tiny.params", 768, 10),
    "nin-cifar100": ("nin-cifar100.def", "nin-cifar100.params", 3072, 100),
    "nin-mnist": ("nin-mnist.def", "nin-mnist.params", 784, 10),
    "nin-svhn": ("nin-svhn.def", "nin-svhn.params", 3072, 10),
    "19pct": ("layers-19pct.cfg", "layer-params-19pct.cfg", 3072, 10),
    "80sec": ("layers-80sec.cfg", "layer-params-80sec.cfg", 3072, 10),
    "18pct": ("layers-18pct.cfg", "layer-params-18pct.cfg", 3072, 10),
    "conv-local": ("layers-conv-local-13pct.cfg", "layer-params-conv-local-13pct.cfg", 3072, 10),
    "imagenet": ("imagenet.def", "imagenet.params", 150528, 100),
    "gc-small": ("gc-small.def", "gc-small.params", 432, 10),
    "image-layers": ("image-layers.def", "image-layers.params", 768, 10),
    "gc-sparse": ("gc-sparse.def", "gc-sparse.params", 432, 10),
    "example-small": ("example-small.def", "example-small.params", 768, 10),
    # the reference's own --check-grads fixture, a verbatim copy of example-layers/layers.gc.cfg + layer-params.gc.cfg (run there
    # on `--data-provider dummy-cn-<dims>`; any 3 * S * S works, 432 = 12 x 12 images).  The other fixture it ships there,
    # layers-example.cfg + layer-params-example.cfg, is rejected by the fork's own parser (and by ours, the same way: see
    # tests/test_layer_parser.py::test_reference_example_config_is_rejected_as_the_reference_rejects_it); "example-small" above
    # is its runnable analogue.
    "gc": ("layers.gc.cfg", "layer-params.gc.cfg", 432, 10),
}


class ShapeProvider(object):
    """The two questions layer.py asks a data provider."""

    def __init__(self, dims, classes):
        self.dims, self.classes = dims, classes

    def get_data_dims(self, idx=0):
        return self.dims if idx == 0 else 1

    def get_num_classes(self):
        return self.classes


class _Model(object):
    op = None


def parse_net(layer_def, layer_params, dims, classes, seed=42, quiet=True):
    """Layer dicts with weights drawn from numpy's global RNG seeded with `seed`
    (initW * N(0,1), Python/layer.py:726 of the reference)."""
    import layer as lay
    m = _Model()
    m.train_data_provider = ShapeProvider(dims, classes)
    nr.seed(seed)
    ctx = contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()
    with ctx:
        layers = lay.LayerParser.parse_layers(layer_def, layer_params, m)
    return layers


def load_net(name, seed=42, init_w=0.0):
    """init_w > 0 redraws every weight matrix as init_w * N(0,1) (seeded), overriding the def file's initW: nets whose own
    initW leaves the activations near zero get well-separated max-pool windows that way."""
    d, p, dims, classes = NETS[name]
    layers = parse_net(os.path.join(CONFIGS, d), os.path.join(CONFIGS, p), dims, classes, seed=seed)
    if init_w > 0:
        rng = n.random.default_rng(seed + 1000)
        for l in layers:
            for w in l.get("weights", []):
                w[...] = (init_w * rng.standard_normal(w.shape)).astype(n.float32)   # in place: keeps the C / Fortran order
    return layers, dims, classes


def synthetic_batch(dims, classes, cases, seed_data=1234, seed_labels=5678):
    """BASELINE.md section 2 inputs: data U[0,1) float32 (dims, cases), labels uniform ints as float32 (1, cases)."""
    data = n.random.default_rng(seed_data).random((dims, cases), dtype=n.float32)
    labels = n.random.default_rng(seed_labels).integers(0, classes, size=(1, cases)).astype(n.float32)
    return [data, labels]


def dropout_layers(layers):
    """(name, outputs) of every dropout neuron layer, in fprop order."""
    return [(l["name"], l["outputs"]) for l in layers if l["type"] == "neuron" and l["neuron"]["type"] == "dropout"]
