"""Layer-definition / layer-parameter front end (Python 3): two INI files in, the list of layer dicts that IS the model
specification of `libkernel.initModel` out (same keys, value types and derived sizes as the reference's Python/layer.py, whose
file formats and stdout lines this keeps so that existing configs run unchanged).

Design: a layer type is a *schema* -- a table of its config keys (name, value kind, default, per-input or not, allowed
range) plus one `build` function that derives sizes, validates and allocates -- and ONE engine (`LayerParser`) walks schemas.
The engine reads a key table the same way for every type, so a new layer type is a table and a function, not a class
hierarchy; the named `*Parser` classes below are the public handles config / plugin authors already use (`layer_parsers`,
`neuron_parsers`, `parse` / `add_params` / `optimize`, `LayerParser.parse_layers`, `LayerParser.register_layer_parser`,
`WeightLayerParser.make_weights` ...) and each just names its schema.

Deliberate deviations from the reference (both make shipped configs usable, neither changes a valid config's result):
  * norm layers default `k` to 1 when the params file has none (the reference requires a key that no shipped params file
    provides and only cmrnorm reads, Kernel/src/layers/layer.cu:904);
  * the data provider is any object with `get_data_dims(idx)` / `get_num_classes()`.
"""
import configparser
import os
import re
import sys
from math import ceil, exp, floor

import numpy as n
import numpy.random as nr


class LayerParsingError(Exception):
    pass


# ==========================================================================================
# value kinds: how a raw INI string becomes a value, and how to say what was expected
# ==========================================================================================
_TRUE, _FALSE = ("1", "yes", "true", "on"), ("0", "no", "false", "off")


def _to_bool(s):
    s = s.strip().lower()
    if s in _TRUE:
        return True
    if s in _FALSE:
        return False
    raise ValueError(s)


KINDS = {   # kind -> (converter of one item, is a ','-list, words for the error message)
    "str": (str, False, "a string"), "int": (int, False, "int"), "float": (float, False, "float"), "bool": (_to_bool, False, "bool"),
    "strs": (str, True, "strings"), "ints": (int, True, "ints"), "floats": (float, True, "floats"),
    "bools": (lambda x: x.lower() in ("true", "1"), True, "bools"),
}
REQUIRED = object()


class MyConfigParser(configparser.RawConfigParser):
    """INI access whose failures name the layer and the key (the `safe_get*` family is public: plugins call it)."""

    def __init__(self, *a, **kw):
        kw.setdefault("strict", False)
        kw.pop("dict_type", None)
        configparser.RawConfigParser.__init__(self, *a, **kw)

    def fetch(self, section, option, kind, default=None):
        conv, is_list, words = KINDS[kind]
        try:
            raw = configparser.RawConfigParser.get(self, section, option)
        except (configparser.NoOptionError, configparser.NoSectionError):
            if default is None:
                raise LayerParsingError("Layer '%s': required parameter '%s' missing" % (section, option))
            return default
        try:
            return [conv(x.strip()) for x in raw.split(",")] if is_list else conv(raw)
        except Exception:
            if is_list:
                raise LayerParsingError("Layer '%s': parameter '%s' must be ','-delimited list of %s" % (section, option, words))
            raise LayerParsingError("Layer '%s': parameter '%s' must be %s" % (section, option, words))

    def safe_get(self, section, option, f=None, typestr=None, default=None):
        try:
            raw = configparser.RawConfigParser.get(self, section, option)
        except (configparser.NoOptionError, configparser.NoSectionError):
            if default is None:
                raise LayerParsingError("Layer '%s': required parameter '%s' missing" % (section, option))
            return default
        if f is None:
            return raw
        try:
            return f(raw)
        except (ValueError, KeyError):
            if typestr is None:
                raise
            raise LayerParsingError("Layer '%s': parameter '%s' must be %s" % (section, option, typestr))

    def safe_get_list(self, section, option, f=str, typestr="strings", default=None):
        v = self.safe_get(section, option, default=default)
        if isinstance(v, list):
            return v
        try:
            return [f(x.strip()) for x in str(v).split(",")]
        except Exception:
            raise LayerParsingError("Layer '%s': parameter '%s' must be ','-delimited list of %s" % (section, option, typestr))

    def safe_get_int(self, s, o, default=None):
        return self.fetch(s, o, "int", default)

    def safe_get_float(self, s, o, default=None):
        return self.fetch(s, o, "float", default)

    def safe_get_bool(self, s, o, default=None):
        return self.fetch(s, o, "bool", default)

    def safe_get_float_list(self, s, o, default=None):
        return self.fetch(s, o, "floats", default)

    def safe_get_int_list(self, s, o, default=None):
        return self.fetch(s, o, "ints", default)

    def safe_get_bool_list(self, s, o, default=None):
        return self.fetch(s, o, "bools", default)


class _DictConfig(object):
    """A dict answering for an INI section (a neuron layer split off its parent has no section of its own)."""

    def __init__(self, values):
        self.values = values

    def fetch(self, section, option, kind, default=None):
        return self.values[option]

    def safe_get(self, section, option, default=None, **kw):
        return self.values[option]


class Key(object):
    """One config key of a layer type.  per_input: a ','-list with one entry per input (a missing key takes `default` for
    every input); lo / hi: inclusive range checked on every entry (hi may be the name of another key of the layer)."""

    def __init__(self, name, kind, default=REQUIRED, per_input=False, lo=None, hi=None):
        self.name, self.kind, self.default, self.per_input, self.lo, self.hi = name, kind, default, per_input, lo, hi


# ==========================================================================================
# neuron strings: "relu", "tanh[1.79,-0.66]", "dropout[0.5]"
# ==========================================================================================
class NeuronParser(object):
    """`type` may carry parameter names, "tanh[a,b]": then "tanh[1.79,-0.66]" parses to params {"a": 1.79, "b": -0.66}."""
    _DECL = re.compile(r"^\s*(\w+)\s*(?:\[\s*(\w+(?:\s*,\w+)*)\s*\])?\s*$")

    def __init__(self, type, func_str, uses_acts=True, uses_inputs=True):
        self.type, self.func_str, self.uses_acts, self.uses_inputs = type, func_str, uses_acts, uses_inputs
        m = self._DECL.match(type)
        self.base_type = m.group(1)
        self.param_names = m.group(2).split(",") if m.group(2) else []
        if len(set(self.param_names)) != len(self.param_names):
            raise ValueError("neuron '%s' repeats a parameter name" % type)

    def parse(self, text):
        if not self.param_names:
            values = [] if text == self.type else None
        else:
            m = re.match(r"^%s\s*\[([\d,\.\s\-e]*)\]\s*$" % self.base_type, text)
            try:
                values = [float(v.strip()) for v in m.group(1).split(",")] if m else None
            except ValueError:
                values = None
        if values is None or len(values) != len(self.param_names):
            return None
        return {"type": self.base_type, "params": self.adjust(dict(zip(self.param_names, values))), "usesActs": self.uses_acts,
                "usesInputs": self.uses_inputs}

    def adjust(self, params):
        return params


ParamNeuronParser = NeuronParser   # the reference has two classes; one covers both here


class AbsTanhNeuronParser(NeuronParser):
    def __init__(self):
        NeuronParser.__init__(self, "abstanh[a,b]", "f(x) = a * |tanh(b * x)|")

    def adjust(self, params):
        params["b"] = abs(params["b"])
        return params


# ==========================================================================================
# the engine
# ==========================================================================================
class LayerParser(object):
    """Walks a schema: `keys` (def file), `param_keys` (params file), `num_inputs`, the sharing flags, and `build`."""
    keys = ()
    param_keys = ()
    num_inputs = 0          # 0: no `inputs` key at all (data layers); -1: any number; k: exactly k
    flags = {}              # overrides of the defaults in set_defaults()

    def __init__(self):
        self.dic = {}
        self.set_defaults()

    def init(self, d):
        self.dic = d
        return self

    def set_defaults(self):
        self.dic.update(outputs=0, parser=self, requiresParams=bool(self.param_keys),
                        usesActs=True,        # the layer reads its own output again (for its gradient)
                        usesInputs=True,      # ... and its inputs
                        forceOwnActs=True,    # its gradient map is not element-wise: it needs a gradient matrix of its own
                        gradConsumer=False)   # only layers with parameters consume gradient themselves
        self.dic.update(self.flags)

    # ---------------------------------------------------------------- def file
    def parse(self, name, mcp, prev_layers, model=None):
        d = self.dic
        self.prev_layers, self.model = prev_layers, model
        d["name"] = name
        d["type"] = mcp.safe_get(name, "type")
        if self.num_inputs != 0:
            self._wire_inputs(mcp)
        for key in self.keys:
            d[key.name] = self._read(mcp, key)
        for key in self.keys:
            self._check_range(key)
        self.prepare()
        self.build(mcp)
        return d

    def _wire_inputs(self, mcp):
        d, name, prev = self.dic, self.dic["name"], self.prev_layers
        where = dict((p["name"], i) for i, p in enumerate(prev))
        wanted = [s.strip() for s in mcp.safe_get(name, "inputs").split(",")]
        for w in wanted:
            if w not in where:
                raise LayerParsingError("Layer '%s': input layer '%s' not defined" % (name, w))
            if prev[where[w]]["outputs"] == 0:
                raise LayerParsingError("Layer '%s': input layer '%s' does not produce any output" % (name, w))
        d["inputs"] = [where[w] for w in wanted]
        d["inputLayers"] = [prev[i] for i in d["inputs"]]
        d["numInputs"] = [prev[i]["outputs"] for i in d["inputs"]]
        d["neuron"] = mcp.safe_get(name, "neuron", default="")
        if self.num_inputs > 0 and len(wanted) != self.num_inputs:
            raise LayerParsingError("Layer '%s': number of inputs must be %d" % (name, self.num_inputs))

    def _read(self, mcp, key):
        d = self.dic
        default = key.default
        if callable(default):
            default = default(d)
        if key.per_input and default is not REQUIRED and not isinstance(default, list):
            default = [default] * len(d["inputs"])
        value = mcp.fetch(d["name"], key.name, key.kind, None if default is REQUIRED else default)
        if key.per_input and len(value) != len(d["inputs"]):
            raise LayerParsingError("Layer '%s': %s list length does not match number of inputs" % (d["name"], key.name))
        return value

    def _check_range(self, key):
        if key.lo is None and key.hi is None:
            return
        hi = self.dic[key.hi] if isinstance(key.hi, str) else key.hi
        fmt = (lambda x: "%.3f" % x) if key.kind in ("float", "floats") else (lambda x: "%d" % x)
        self.verify_num_range(self.dic[key.name], key.name, key.lo, hi, strconv=fmt)

    def prepare(self):
        """Family-specific part that has to precede build() (weight layers resolve their sharing sources here)."""

    def build(self, mcp):
        """Type-specific part: derive sizes, validate, allocate, print the banner."""

    # ---------------------------------------------------------------- params file
    def add_params(self, mcp):
        for key in self.param_keys:
            self.dic[key.name] = self._read(mcp, key)
        self.finish_params()

    def finish_params(self):
        pass

    # ---------------------------------------------------------------- activity / gradient matrix sharing
    def optimize(self, layers):
        """A layer whose map is element-wise may write into its input's matrices when it is that input's only consumer: the
        gradient matrix always, the activity matrix when neither side needs the overwritten values again."""
        d = self.dic
        d["actsTarget"] = d["actsGradTarget"] = -1
        if d["forceOwnActs"] or "inputs" not in d:
            return
        for slot, src in enumerate(d["inputs"]):
            consumers = sum(1 for other in layers if src in other.get("inputs", ()))
            if consumers == 1 and layers[src]["outputs"] == d["outputs"]:
                if not layers[src]["usesActs"] and not d["usesInputs"]:
                    d["actsTarget"] = slot
                d["actsGradTarget"] = slot

    # ---------------------------------------------------------------- validation helpers (public: plugins use them)
    def _who(self, input_idx=-1):
        return self.dic["name"] if input_idx < 0 else "%s[%d]" % (self.dic["name"], input_idx)

    def verify_num_range(self, v, param_name, _min, _max, strconv=lambda x: "%d" % x):
        for idx, item in (enumerate(v) if isinstance(v, list) else [(-1, v)]):
            below, above = _min is not None and item < _min, _max is not None and item > _max
            if not (below or above):
                continue
            if _min is not None and _max is not None:
                what = "in the range %s-%s" % (strconv(_min), strconv(_max))
            elif below:
                what = "greater than or equal to %s" % strconv(_min)
            else:
                what = "smaller than or equal to %s" % strconv(_max)
            raise LayerParsingError("Layer '%s': parameter '%s' must be %s" % (self._who(idx), param_name, what))

    def verify_float_range(self, v, param_name, _min, _max):
        self.verify_num_range(v, param_name, _min, _max, strconv=lambda x: "%.3f" % x)

    def verify_divisible(self, value, div, value_name, div_name=None, input_idx=0):
        if value % div:
            who = self._who(input_idx if self.dic.get("inputs") else -1)
            raise LayerParsingError("Layer '%s': parameter '%s' must be divisible by %s" % (who, value_name, str(div) if div_name is None else "'%s'" % div_name))

    def verify_one_of(self, value, allowed, fmt="'%s'"):
        if value not in allowed:
            raise LayerParsingError("Layer '%s': parameter '%s' must be one of %s" % (self.dic["name"], value, ", ".join(fmt % a for a in allowed)))

    verify_str_in = verify_one_of

    def verify_int_in(self, value, allowed):
        self.verify_one_of(value, allowed, fmt="'%d'")

    def verify_num_params(self, params):
        for p in params:
            if len(self.dic[p]) != len(self.dic["inputs"]):
                raise LayerParsingError("Layer '%s': %s list length does not match number of inputs" % (self.dic["name"], p))

    def square_image(self, channels):
        """Reads input 0 as (channels, size, size): sets imgPixels / imgSize, complains if it is not that."""
        d = self.dic
        d["imgPixels"] = d["numInputs"][0] // channels
        d["imgSize"] = int(n.sqrt(d["imgPixels"]))
        return d["imgSize"]

    def verify_img_size(self):
        d = self.dic
        if d["numInputs"][0] % d["imgPixels"] or d["imgSize"] ** 2 != d["imgPixels"]:
            raise LayerParsingError("Layer '%s': has %-d dimensional input, not interpretable as %d-channel images" % (d["name"], d["numInputs"][0], d["channels"]))

    def verify_no_grads(self):
        def consumes(d):
            return d["gradConsumer"] or any(consumes(x) for x in d.get("inputLayers", ()))
        if consumes(self.dic):
            raise LayerParsingError("Layer '%s': layers of type '%s' cannot propagate gradient and must not be placed over layers with parameters." % (self.dic["name"], self.dic["type"]))

    def banner(self, what, shape):
        print("Initialized %s '%s', producing %s" % (what, self.dic["name"], shape))

    def image_banner(self, what, size, channels=None):
        self.banner(what, "%dx%d %d-channel output" % (size, size, self.dic["channels"] if channels is None else channels))

    # ---------------------------------------------------------------- whole-file driver
    @staticmethod
    def detach_neuron_layers(layers):
        """`neuron=x` on a layer becomes a neuron layer of its own right behind it; indices further down shift."""
        out = []
        for pos, l in enumerate(layers):
            out.append(l)
            if l["type"] != "neuron" and l.get("neuron"):
                NeuronLayerParser().detach_neuron_layer(pos, layers, out)
        return out

    @staticmethod
    def parse_layers(layer_cfg_path, param_cfg_path, model, layers=None):
        """First run: topology from the def file, then hyper-parameters from the params file.  With `layers` given (a
        checkpoint) only the params file is read again, so learning rates can change between runs."""
        layers = [] if layers is None else layers
        try:
            for path, what in ((layer_cfg_path, "definition"), (param_cfg_path, "parameter")):
                if not os.path.exists(path):
                    raise LayerParsingError("Layer %s file '%s' does not exist" % (what, path))
            if not layers:
                layers = LayerParser._read_topology(layer_cfg_path, model)
            LayerParser._read_hyper_parameters(param_cfg_path, layers, model)
        except LayerParsingError as e:
            print(e)
            sys.exit(1)
        return layers

    @staticmethod
    def _read_topology(path, model):
        mcp = MyConfigParser()
        mcp.read([path])
        layers = []
        for name in mcp.sections():
            if not mcp.has_option(name, "type"):
                raise LayerParsingError("Layer '%s': no type given" % name)
            ltype = mcp.safe_get(name, "type")
            if ltype not in layer_parsers:
                raise LayerParsingError("Layer '%s': Unknown layer type: '%s'" % (name, ltype))
            layers.append(layer_parsers[ltype]().parse(name, mcp, layers, model))
        layers = LayerParser.detach_neuron_layers(layers)
        for l in layers:
            l.pop("parser").optimize(layers)
        fed = set(layers[i]["name"] for l in layers for i in l.get("inputs", ()))
        for l in layers:
            if l["name"] not in fed and not l["type"].startswith("cost."):
                raise LayerParsingError("Layer '%s' of type '%s' is unused" % (l["name"], l["type"]))
        return layers

    @staticmethod
    def _read_hyper_parameters(path, layers, model):
        mcp = MyConfigParser()
        mcp.read([path])
        op = getattr(model, "op", None)
        conserve = op.get_value("conserve_mem") if op is not None else 0
        for l in layers:
            if l["requiresParams"] and not mcp.has_section(l["name"]):
                raise LayerParsingError("Layer '%s' of type '%s' requires extra parameters, but none given in file '%s'." % (l["name"], l["type"], path))
            layer_parsers[l["type"]]().init(l).add_params(mcp)
            l["conserveMem"] = conserve

    @staticmethod
    def register_layer_parser(ltype, cls):
        if ltype in layer_parsers:
            raise LayerParsingError("Layer type '%s' already registered" % ltype)
        layer_parsers[ltype] = cls


class LayerWithInputParser(LayerParser):
    num_inputs = -1

    def __init__(self, num_inputs=None):
        if num_inputs is not None:
            self.num_inputs = num_inputs
        LayerParser.__init__(self)


ELEMENTWISE = dict(forceOwnActs=False, usesActs=False, usesInputs=False)


# ==========================================================================================
# layers without parameters
# ==========================================================================================
class DataLayerParser(LayerParser):
    keys = (Key("dataIdx", "int"),)

    def build(self, mcp):
        self.dic["outputs"] = self.model.train_data_provider.get_data_dims(idx=self.dic["dataIdx"])
        self.banner("data layer", "%d outputs" % self.dic["outputs"])


class NeuronLayerParser(LayerWithInputParser):
    num_inputs = 1
    flags = dict(forceOwnActs=False)

    def build(self, mcp):
        d = self.dic
        d["outputs"] = d["numInputs"][0]
        text = d["neuron"]
        for p in neuron_parsers:
            found = p.parse(text)
            if found:
                d["neuron"], d["usesActs"], d["usesInputs"] = found, found["usesActs"], found["usesInputs"]
                break
        else:
            width = max(len("Neuron type"), max(len(p.type) for p in neuron_parsers)) + 2
            rows = ["Neuron type".ljust(width) + "Function"] + [p.type.ljust(width) + p.func_str for p in neuron_parsers]
            raise LayerParsingError("Layer '%s': unable to parse neuron type '%s'. Valid neuron types: %s%s%sWhere neurons have parameters, they must be floats."
                                    % (d["name"], text, os.linesep, os.linesep.join(rows), os.linesep))
        self.banner("neuron layer", "%d outputs" % d["outputs"])

    def detach_neuron_layer(self, pos, layers, out):
        parent = layers[pos]
        taken = set(l["name"] for l in layers)
        name = next(c for c in ["%s_neuron" % parent["name"]] + ["%s_neuron.%d" % (parent["name"], k) for k in range(1, 100)] if c not in taken)
        born = self.parse(name, _DictConfig({"type": "neuron", "inputs": parent["name"], "neuron": parent["neuron"]}), out)
        slot = len(out)          # where the new layer lands; its parent sits at slot - 1 and later layers move down by one
        for l in layers[pos + 1:]:
            if "inputs" in l:                        # readers of the parent now read the neuron layer
                l["inputs"] = [i + (i >= slot - 1) for i in l["inputs"]]
            if "weightSourceLayerIndices" in l:      # ... but shared weights stay with the parent
                l["weightSourceLayerIndices"] = [i + (i >= slot) for i in l["weightSourceLayerIndices"]]
        out.append(born)


class SameSizeInputsMixin(object):
    def same_size_inputs(self):
        d = self.dic
        if len(set(d["numInputs"])) != 1:
            raise LayerParsingError("Layer '%s': all inputs must have the same dimensionality. Got dimensionalities: %s" % (d["name"], ", ".join(str(s) for s in d["numInputs"])))
        d["outputs"] = d["numInputs"][0]


class EltwiseSumLayerParser(LayerWithInputParser, SameSizeInputsMixin):
    flags = ELEMENTWISE
    keys = (Key("coeffs", "floats", default=lambda d: [1.0] * len(d["inputs"])),)

    def build(self, mcp):
        self.same_size_inputs()
        self.banner("elementwise sum layer", "%d outputs" % self.dic["outputs"])


class EltwiseMaxLayerParser(LayerWithInputParser, SameSizeInputsMixin):
    def build(self, mcp):
        if len(self.dic["inputs"]) < 2:
            raise LayerParsingError("Layer '%s': elementwise max layer must have at least 2 inputs, got %d." % (self.dic["name"], len(self.dic["inputs"])))
        self.same_size_inputs()
        self.banner("elementwise max layer", "%d outputs" % self.dic["outputs"])


class SoftmaxLayerParser(LayerWithInputParser):
    num_inputs = 1

    def build(self, mcp):
        self.dic["outputs"] = self.dic["numInputs"][0]
        self.banner("softmax layer", "%d outputs" % self.dic["outputs"])


class PoolLayerParser(LayerWithInputParser):
    num_inputs = 1
    keys = (Key("channels", "int", lo=1), Key("sizeX", "int"), Key("start", "int", default=0), Key("stride", "int"),
            Key("outputsX", "int", default=0, lo=0), Key("pool", "str"))
    # (the reference derives usesActs / usesInputs from `'pool' != 'avg'`, i.e. always True: the defaults, kept, because they
    # decide which matrices the runtime may share)

    def build(self, mcp):
        d = self.dic
        size = self.square_image(d["channels"])
        self.verify_num_range(d["sizeX"], "sizeX", 1, size)
        self.verify_num_range(d["stride"], "stride", 1, d["sizeX"])
        self.verify_str_in(d["pool"], ["max", "avg"])
        self.verify_img_size()
        if d["outputsX"] <= 0:
            d["outputsX"] = int(ceil((size - d["start"] - d["sizeX"]) / float(d["stride"]))) + 1
        d["outputs"] = d["outputsX"] ** 2 * d["channels"]
        self.image_banner("%s-pooling layer" % d["pool"], d["outputsX"])


class NormLayerParser(LayerWithInputParser):
    RESPONSE_NORM, CONTRAST_NORM, CROSSMAP_RESPONSE_NORM = "response", "contrast", "cross-map response"
    num_inputs = 1
    keys = (Key("channels", "int", lo=1), Key("size", "int"), Key("blocked", "bool", default=False))
    param_keys = (Key("scale", "float"), Key("pow", "float"), Key("k", "float", default=1.0))

    def __init__(self, norm_type):
        self.norm_type = norm_type
        LayerWithInputParser.__init__(self)

    def build(self, mcp):
        d, across = self.dic, self.norm_type == self.CROSSMAP_RESPONSE_NORM
        size = self.square_image(d["channels"])
        d["usesInputs"] = self.norm_type != self.CONTRAST_NORM
        self.verify_num_range(d["size"], "size", 2 if across else 1, d["channels"] if across else size)
        if across and d["channels"] % 16:
            raise LayerParsingError("Layer '%s': number of channels must be divisible by 16 when using crossMap" % d["name"])
        if not across and d["channels"] > 3 and d["channels"] % 4:
            raise LayerParsingError("Layer '%s': number of channels must be smaller than 4 or divisible by 4" % d["name"])
        self.verify_img_size()
        d["outputs"] = d["imgPixels"] * d["channels"]
        self.image_banner("%s-normalization layer" % self.norm_type, size)

    def finish_params(self):
        # the file gives the scale of the whole window; the kernels want it per window element
        window = self.dic["size"] if self.norm_type == self.CROSSMAP_RESPONSE_NORM else self.dic["size"] ** 2
        self.dic["scale"] /= window


class ImageOpLayerParser(LayerWithInputParser):
    """One-input layers that map (channels, size, size) images to images and have no parameters."""
    num_inputs = 1
    flags = ELEMENTWISE


class NailbedLayerParser(ImageOpLayerParser):
    keys = (Key("channels", "int", lo=1), Key("stride", "int"))

    def build(self, mcp):
        d = self.dic
        size = self.square_image(d["channels"])
        d["outputsX"] = (size + d["stride"] - 1) // d["stride"]
        d["start"] = (size - d["stride"] * (d["outputsX"] - 1)) // 2      # nails centred in the image
        d["outputs"] = d["channels"] * d["outputsX"] ** 2
        self.verify_num_range(d["outputsX"], "outputsX", 0, None)
        self.verify_img_size()
        self.image_banner("bed-of-nails layer", d["outputsX"])


class GaussianBlurLayerParser(ImageOpLayerParser):
    keys = (Key("channels", "int", lo=1), Key("filterSize", "int"), Key("stdev", "float"))

    def build(self, mcp):
        d = self.dic
        d["outputs"] = d["numInputs"][0]
        self.verify_int_in(d["filterSize"], [3, 5, 7, 9])
        size = self.square_image(d["channels"])
        taps, half = d["filterSize"], d["filterSize"] // 2
        bell = n.array([exp(-(half - i) ** 2 / float(2 * d["stdev"] ** 2)) for i in range(taps)], dtype=n.float32).reshape(1, taps)
        d["filter"] = bell / bell.sum()
        self.verify_img_size()
        if taps > size:
            raise LayerParsingError("Layer '%s': filter size (%d) must be smaller than image size (%d)." % (d["name"], taps, size))
        self.image_banner("Gaussian blur layer", size)


class ResizeLayerParser(ImageOpLayerParser):
    keys = (Key("channels", "int", lo=1), Key("scale", "float", lo=0.5, hi=2))
    factor, what = "scale", "resize layer"

    def build(self, mcp):
        d = self.dic
        size = self.square_image(d["channels"])
        d["tgtSize"] = int(floor(size / d[self.factor]))
        d["tgtPixels"] = d["tgtSize"] ** 2
        d["outputs"] = d["channels"] * d["tgtPixels"]
        self.verify_img_size()
        self.verify_no_grads()
        self.image_banner(self.what, d["tgtSize"])


class RandomScaleLayerParser(ResizeLayerParser):
    keys = (Key("channels", "int", lo=1), Key("maxScale", "float", lo=1, hi=2))
    factor, what = "maxScale", "random scale layer"


class ColorTransformLayerParser(ImageOpLayerParser):
    what = "colour transform layer"

    def build(self, mcp):
        d = self.dic
        size = self.square_image(3)
        d["channels"] = 3
        d["outputs"] = d["numInputs"][0]
        self.verify_img_size()
        self.verify_no_grads()
        self.image_banner(self.what, size)


class RGBToYUVLayerParser(ColorTransformLayerParser):
    what = "RGB --> YUV layer"


class RGBToLABLayerParser(ColorTransformLayerParser):
    what = "RGB --> LAB layer"
    keys = (Key("center", "bool", default=False),)


# ==========================================================================================
# layers with weights
# ==========================================================================================
class WeightLayerParser(LayerWithInputParser):
    flags = dict(gradConsumer=True)
    keys = (Key("initW", "floats", default=0.01, per_input=True), Key("initB", "float", default=0), Key("initWFunc", "str", default=""),
            Key("initBFunc", "str", default=""), Key("weightSource", "strs", default="", per_input=True))
    param_keys = (Key("epsW", "floats", per_input=True), Key("epsB", "float"), Key("momW", "floats", per_input=True),
                  Key("momB", "float"), Key("wc", "floats", per_input=True))
    _REF = re.compile(r"^\s*([^\s\[]+)(?:\[(\d+)\])?\s*$")     # "layer" or "layer[3]"

    @staticmethod
    def get_layer_name(name_str):
        m = WeightLayerParser._REF.match(name_str)
        return (m.group(1), m.group(2)) if m else None

    def finish_params(self):
        self.dic["gradConsumer"] = self.dic["epsB"] > 0 or any(e > 0 for e in self.dic["epsW"])

    def prepare(self):
        """weightSource = layer[idx] per input -> indices of the source layer / matrix (-1: own weights).  A layer may share
        with an earlier matrix of itself; its index is then len(prev_layers), the slot it is about to take."""
        d, name, prev = self.dic, self.dic["name"], self.prev_layers
        where = dict((p["name"], i) for i, p in enumerate(prev))
        d["requiresParams"] = True
        d["weightSourceLayerIndices"], d["weightSourceMatrixIndices"] = [], []
        for slot, ref in enumerate(d["weightSource"]):
            layer_idx = matrix_idx = -1
            if ref != "":
                parsed = self.get_layer_name(ref)
                if parsed is None:
                    raise LayerParsingError("Layer '%s': unable to parse weight sharing source '%s'. Format is layer[idx] or just layer, in which case idx=0 is used." % (name, ref))
                src_name, matrix_idx = parsed[0], int(parsed[1] or 0)
                own = src_name == name
                if not own and src_name not in where:
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' does not exist." % (name, src_name))
                layer_idx = len(prev) if own else where[src_name]
                src = d if own else prev[layer_idx]
                if src["type"] != d["type"]:
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' is of type '%s'; should be '%s'." % (name, src_name, src["type"], d["type"]))
                if own and matrix_idx >= slot:
                    raise LayerParsingError("Layer '%s': weight sharing source '%s[%d]' not defined yet." % (name, name, matrix_idx))
                if not own and matrix_idx >= len(src["weights"]):
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' has %d weight matrices, but '%s[%d]' requested." % (name, src_name, len(src["weights"]), ref, matrix_idx))
            d["weightSourceLayerIndices"].append(layer_idx)
            d["weightSourceMatrixIndices"].append(matrix_idx)

    def _source_layer(self, slot):
        idx = self.dic["weightSourceLayerIndices"][slot]
        return self.prev_layers[idx] if idx < len(self.prev_layers) else self.dic

    # -- user-supplied initialisers: initWFunc / initBFunc / initCFunc = module.function(p1,p2,...)
    def call_init_func(self, param_name, shapes, input_idx=-1):
        d = self.dic
        m = re.match(r"^([^\.]+)\.([^\(\)]+)\s*(?:\(([^,]+(?:,[^,]+)*)\))?$", d[param_name])
        if not m:
            raise LayerParsingError("Layer '%s': '%s' parameter must have format 'moduleName.functionName(param1,param2,...)'; got: %s." % (d["name"], param_name, d[param_name]))
        args = m.group(3).split(",") if m.group(3) is not None else []
        try:
            fn = getattr(__import__(m.group(1)), m.group(2))
            return fn(d["name"], input_idx, shapes, params=args) if input_idx >= 0 else fn(d["name"], shapes, params=args)
        except (ImportError, AttributeError, TypeError) as e:
            raise LayerParsingError("Layer '%s': %s." % (d["name"], e))

    def _vet(self, what, arr, shape, fname, slot=-1):
        who = self._who(slot)
        if not isinstance(arr, n.ndarray):
            raise LayerParsingError("Layer '%s': %s initialization function %s must return numpy.ndarray object. Got: %s." % (who, what, fname, type(arr)))
        if arr.dtype != n.float32:
            raise LayerParsingError("Layer '%s': %s initialization function %s must return single-precision floats. Got: %s." % (who, what, fname, arr.dtype))
        if arr.shape != shape:
            raise LayerParsingError("Layer '%s': %s returned by initialization function %s has wrong shape. Should be: %s; got: %s." % (who, what, fname, shape, arr.shape))

    def make_weights(self, initW, rows, cols, order="C"):
        """One (rows[i], cols[i]) float32 matrix per input: from initWFunc, shared with weightSource (the SAME ndarray object,
        which is how the runtime recognises sharing), or initW[i] * N(0,1).  weightsInc alongside, zero."""
        d = self.dic
        d["weights"], d["weightsInc"] = [], []
        for slot in range(len(d["inputs"])):
            shape = (rows[slot], cols[slot])
            if d["initWFunc"]:
                w = self.call_init_func("initWFunc", shape, input_idx=slot)
                self._vet("weight matrix", w, shape, d["initWFunc"], slot)
                w, inc = n.require(w, requirements=order), None
                print("Layer '%s[%d]' initialized weight matrices from function %s" % (d["name"], slot, d["initWFunc"]))
            elif d["weightSourceLayerIndices"][slot] >= 0:
                src, mi = self._source_layer(slot), d["weightSourceMatrixIndices"][slot]
                w, inc = src["weights"][mi], src["weightsInc"][mi]
                if w.shape != shape:
                    raise LayerParsingError("Layer '%s': weight sharing source matrix '%s' has shape %dx%d; should be %dx%d." % ((d["name"], d["weightSource"][slot]) + w.shape + shape))
                print("Layer '%s' initialized weight matrix %d from %s" % (d["name"], slot, d["weightSource"][slot]))
            else:
                w, inc = n.array(initW[slot] * nr.randn(*shape), dtype=n.single, order=order), None
            d["weights"].append(w)
            d["weightsInc"].append(n.zeros_like(w) if inc is None else inc)

    def make_biases(self, rows, cols, order="C"):
        d = self.dic
        if d["initBFunc"]:
            b = self.call_init_func("initBFunc", (rows, cols))
            self._vet("bias vector", b, (rows, cols), d["initBFunc"])
            d["biases"] = n.require(b, requirements=order)
            print("Layer '%s' initialized bias vector from function %s" % (d["name"], d["initBFunc"]))
        else:
            d["biases"] = d["initB"] * n.ones((rows, cols), order="C", dtype=n.single)
        d["biasesInc"] = n.zeros_like(d["biases"])

    @staticmethod
    def unshare_weights(layer, layers, matrix_idx=None):
        """--unshare-weights: give `layer` private copies of the matrices it shares (all of them, or one)."""
        if "weightSourceLayerIndices" not in layer:
            return
        for slot in (range(len(layer["inputs"])) if matrix_idx is None else [int(matrix_idx)]):
            src = layer["weightSourceLayerIndices"][slot]
            if src < 0:
                print("Weight matrix %s[%d] already unshared." % (layer["name"], slot))
                continue
            was = (layers[src]["name"], layer["weightSourceMatrixIndices"][slot])
            layer["weightSourceLayerIndices"][slot] = layer["weightSourceMatrixIndices"][slot] = -1
            layer["weights"][slot] = layer["weights"][slot].copy()
            layer["weightsInc"][slot] = n.zeros_like(layer["weights"][slot])
            print("Unshared weight matrix %s[%d] from %s[%d]." % ((layer["name"], slot) + was))


class FCLayerParser(WeightLayerParser):
    flags = dict(gradConsumer=True, usesActs=False)
    keys = WeightLayerParser.keys[:-1] + (Key("outputs", "int", lo=1),) + WeightLayerParser.keys[-1:]

    def build(self, mcp):
        d = self.dic
        # (inputs, outputs) Fortran-ordered = (outputs, inputs) row-major on the device
        self.make_weights(d["initW"], d["numInputs"], [d["outputs"]] * len(d["inputs"]), order="F")
        self.make_biases(1, d["outputs"], order="F")
        self.banner("fully-connected layer", "%d outputs" % d["outputs"])


class LocalLayerParser(WeightLayerParser):
    """Shared core of `conv` (one filter bank) and `local` (a filter bank per module)."""
    flags = dict(gradConsumer=True, usesActs=False)
    keys = (Key("initW", "floats", per_input=True), Key("initB", "float", default=0), Key("initWFunc", "str", default=""), Key("initBFunc", "str", default=""),
            Key("initCFunc", "str", default=""),
            Key("channels", "ints", per_input=True, lo=1), Key("padding", "ints", default=0, per_input=True, lo=0),
            Key("stride", "ints", default=1, per_input=True, lo=1), Key("filterSize", "ints", per_input=True, lo=1),
            Key("filters", "ints", per_input=True), Key("groups", "ints", default=1, per_input=True, lo=1),
            Key("randSparse", "bools", default=False, per_input=True),
            Key("weightSource", "strs", default="", per_input=True))

    def geometry(self, mcp):
        d, name, ni = self.dic, self.dic["name"], len(self.dic["inputs"])
        d["imgPixels"] = [d // c for d, c in zip(d["numInputs"], d["channels"])]
        d["imgSize"] = [int(n.sqrt(p)) for p in d["imgPixels"]]
        self.verify_num_range(d["imgSize"], "imgSize", 1, None)
        d["filters"] = [f * g for f, g in zip(d["filters"], d["groups"])]          # the file gives filters per group
        d["filterPixels"] = [fs * fs for fs in d["filterSize"]]
        per_side = [1 + int(ceil((2 * p + s - fs) / float(st))) for p, s, fs, st in zip(d["padding"], d["imgSize"], d["filterSize"], d["stride"])]
        d["modulesX"] = per_side
        d["filterChannels"] = [c // g for c, g in zip(d["channels"], d["groups"])]
        if any(d["randSparse"]):
            d["filterChannels"] = mcp.fetch(name, "filterChannels", "ints", d["filterChannels"])
            self.verify_num_params(["filterChannels"])
        if len(set(per_side)) != 1 or len(set(d["filters"])) != 1:
            raise LayerParsingError("Layer '%s': all inputs must produce equally-dimensioned output. Dimensions are: %s."
                                    % (name, ", ".join("%dx%dx%d" % (f, m, m) for f, m in zip(d["filters"], per_side))))
        d["modulesX"], d["filters"] = per_side[0], d["filters"][0]
        d["modules"] = d["modulesX"] ** 2
        d["outputs"] = d["modules"] * d["filters"]
        d["filterConns"] = [[]] * ni
        for i in range(ni):
            ch, g, fc, fs, pad, size = (d[k][i] for k in ("channels", "groups", "filterChannels", "filterSize", "padding", "imgSize"))
            if d["numInputs"][i] % d["imgPixels"][i] or size * size != d["imgPixels"][i]:
                raise LayerParsingError("Layer '%s[%d]': has %-d dimensional input, not interpretable as square %d-channel images" % (name, i, d["numInputs"][i], ch))
            if ch > 3 and ch % 4:
                raise LayerParsingError("Layer '%s[%d]': number of channels must be smaller than 4 or divisible by 4" % (name, i))
            if fs > 2 * pad + size:
                raise LayerParsingError("Layer '%s[%d]': filter size (%d) greater than image size + 2 * padding (%d)" % (name, i, fs, 2 * pad + size))
            if d["randSparse"][i]:
                if g == 1:
                    raise LayerParsingError("Layer '%s[%d]': number of groups must be greater than 1 when using random sparse connectivity" % (name, i))
                self.verify_divisible(ch, fc, "channels", "filterChannels", input_idx=i)
                self.verify_divisible(fc, 4, "filterChannels", input_idx=i)
                self.verify_divisible(g * fc, ch, "groups * filterChannels", "channels", input_idx=i)
                d["filterConns"][i] = self.gen_rand_conns(g, ch, fc, i)
            else:
                if g > 1:
                    self.verify_divisible(ch, 4 * g, "channels", "4 * groups", input_idx=i)
                self.verify_divisible(ch, g, "channels", "groups", input_idx=i)
            self.verify_divisible(d["filters"], g, "filters * groups", input_idx=i)
            d["padding"][i] = -pad          # the kernels take paddingStart <= 0
        d["overSample"] = [g * fc // c for g, fc, c in zip(d["groups"], d["filterChannels"], d["channels"])]
        return [fp * fc for fp, fc in zip(d["filterPixels"], d["filterChannels"])]       # rows of one filter bank, per input

    def gen_rand_conns(self, groups, channels, filterChannels, inputIdx):
        """randSparse: which channel each of the groups*filterChannels filter-channel slots reads -- overSample permutations of
        the channels back to back (drawn even when a function or a sharing source then replaces them: the draw is part of
        the seeded sequence the weights come from)."""
        d, over = self.dic, groups * filterChannels // channels
        conns = [int(c) for _ in range(over) for c in nr.permutation(range(channels))]
        who = "%s[%d]" % (d["name"], inputIdx)
        if d["initCFunc"]:
            conns = self.call_init_func("initCFunc", (groups, channels, filterChannels), input_idx=inputIdx)
            if len(conns) != over * channels:
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return list of length <groups> * <filterChannels> = %d; got: %d" % (who, d["initCFunc"], over * channels, len(conns)))
            if any(c not in range(channels) for c in conns):
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return list of channel indices in the range 0-<channels-1> = 0-%d." % (who, d["initCFunc"], channels - 1))
            if any(len(set(conns[o * channels:(o + 1) * channels])) != channels for o in range(over)):
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return a permutation of 0-%d in every block of %d elements." % (who, d["initCFunc"], channels - 1, channels))
        elif d["weightSourceLayerIndices"][inputIdx] >= 0:
            src, si = self._source_layer(inputIdx), d["weightSourceMatrixIndices"][inputIdx]
            if not src.get("randSparse") or not src["randSparse"][si]:
                raise LayerParsingError("Layer '%s': randSparse is true in this layer but false in weight sharing source layer '%s[%d]'." % (who, src["name"], si))
            mine, theirs = (groups, channels, filterChannels), (src["groups"][si], src["channels"][si], src["filterChannels"][si])
            if mine != theirs:
                raise LayerParsingError("Layer '%s': groups, channels, filterChannels set to %d, %d, %d, respectively. Does not match setting in weight sharing source layer '%s[%d]': %d, %d, %d." % ((who,) + mine + (src["name"], si) + theirs))
            conns = src["filterConns"][si]
        return conns

    @staticmethod
    def conv_to_local(layers, idx):
        """--conv-to-local: a conv layer (and every layer sharing its weights) becomes an unshared local layer whose per-module
        filter banks all start as the shared one."""
        layer = layers[idx]
        if layer["type"] != "conv":
            return layer
        layer["type"] = "local"
        for slot, src in enumerate(layer["weightSourceLayerIndices"]):
            if src >= 0:
                LocalLayerParser.conv_to_local(layers, src)
                mi = layer["weightSourceMatrixIndices"][slot]
                layer["weights"][slot], layer["weightsInc"][slot] = layers[src]["weights"][mi], layers[src]["weightsInc"][mi]
            else:
                bank = layer["weights"][slot]
                layer["weights"][slot] = n.require(n.tile(bank.reshape(1, bank.size), (layer["modules"], 1)).reshape(layer["modules"] * bank.shape[0], layer["filters"]), requirements="C")
                layer["weightsInc"][slot] = n.zeros_like(layer["weights"][slot])
        if layer["sharedBiases"]:
            layer["biases"] = n.require(n.repeat(layer["biases"], layer["modules"], axis=0), requirements="C")
            layer["biasesInc"] = n.zeros_like(layer["biases"])
        print("Converted layer '%s' from convolutional to unshared, locally-connected" % layer["name"])
        for other, l in enumerate(layers):
            if idx in l.get("weightSourceLayerIndices", ()):
                LocalLayerParser.conv_to_local(layers, other)
        return layer


class ConvLayerParser(LocalLayerParser):
    keys = LocalLayerParser.keys[:-1] + (Key("partialSum", "int"), Key("sharedBiases", "bool", default=True)) + LocalLayerParser.keys[-1:]

    def build(self, mcp):
        d = self.dic
        bank_rows = self.geometry(mcp)
        if d["partialSum"] and d["modules"] % d["partialSum"]:
            raise LayerParsingError("Layer '%s': convolutional layer produces %dx%d=%d outputs per filter, but given partialSum parameter (%d) does not divide this number"
                                    % (d["name"], d["modulesX"], d["modulesX"], d["modules"], d["partialSum"]))
        self.make_weights(d["initW"], bank_rows, [d["filters"]] * len(d["inputs"]))
        self.make_biases(d["filters"] if d["sharedBiases"] else d["modules"] * d["filters"], 1)
        self.image_banner("convolutional layer", d["modulesX"], d["filters"])


class LocalUnsharedLayerParser(LocalLayerParser):
    def build(self, mcp):
        d = self.dic
        bank_rows = self.geometry(mcp)
        self.make_weights(d["initW"], [d["modules"] * r for r in bank_rows], [d["filters"]] * len(d["inputs"]))
        self.make_biases(d["modules"] * d["filters"], 1)
        self.image_banner("locally-connected layer", d["modulesX"], d["filters"])


# ==========================================================================================
# costs
# ==========================================================================================
class CostParser(LayerWithInputParser):
    param_keys = (Key("coeff", "float"),)

    def parse(self, name, mcp, prev_layers, model=None):
        d = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        d["requiresParams"] = True
        d.pop("neuron", None)      # costs have no neuron option
        return d

    def add_params(self, mcp):
        self.dic["coeff"] = mcp.safe_get_float(self.dic["name"], "coeff")


class LogregCostParser(CostParser):
    num_inputs = 2

    def build(self, mcp):
        d, prev = self.dic, self.prev_layers
        probs = prev[d["inputs"][1]]
        if d["numInputs"][0] != 1:
            raise LayerParsingError("Layer '%s': dimensionality of first input must be 1" % d["name"])
        if probs["type"] != "softmax":
            raise LayerParsingError("Layer '%s': second input must be softmax layer" % d["name"])
        classes = self.model.train_data_provider.get_num_classes()
        if d["numInputs"][1] != classes:
            raise LayerParsingError("Layer '%s': softmax input '%s' must produce %d outputs, because that is the number of classes in the dataset" % (d["name"], probs["name"], classes))
        print("Initialized logistic regression cost '%s'" % d["name"])


class SumOfSquaresCostParser(CostParser):
    num_inputs = 1

    def build(self, mcp):
        print("Initialized sum-of-squares cost '%s'" % self.dic["name"])


# ==========================================================================================
# registries
# ==========================================================================================
layer_parsers = {
    "data": DataLayerParser, "fc": FCLayerParser, "conv": ConvLayerParser, "local": LocalUnsharedLayerParser,
    "softmax": SoftmaxLayerParser, "eltsum": EltwiseSumLayerParser, "eltmax": EltwiseMaxLayerParser, "neuron": NeuronLayerParser,
    "pool": PoolLayerParser,
    "rnorm": lambda: NormLayerParser(NormLayerParser.RESPONSE_NORM),
    "cnorm": lambda: NormLayerParser(NormLayerParser.CONTRAST_NORM),
    "cmrnorm": lambda: NormLayerParser(NormLayerParser.CROSSMAP_RESPONSE_NORM),
    "nailbed": NailbedLayerParser, "blur": GaussianBlurLayerParser, "resize": ResizeLayerParser, "rscale": RandomScaleLayerParser,
    "rgb2yuv": RGBToYUVLayerParser, "rgb2lab": RGBToLABLayerParser,
    "cost.logreg": LogregCostParser, "cost.sum2": SumOfSquaresCostParser,
}

_NEURONS = [   # (type string, formula, reads its output in bprop, reads its input in bprop)
    ("ident", "f(x) = x", False, False), ("logistic", "f(x) = 1 / (1 + e^-x)", True, False), ("abs", "f(x) = |x|", False, True),
    ("relu", "f(x) = max(0, x)", True, False), ("softrelu", "f(x) = log(1 + e^x)", True, False), ("square", "f(x) = x^2", False, True),
    ("sqrt", "f(x) = sqrt(x)", True, False), ("tanh[a,b]", "f(x) = a * tanh(b * x)", True, False),
    ("brelu[a]", "f(x) = min(a, max(0, x))", True, False), ("linear[a,b]", "f(x) = a * x + b", True, False),
]
neuron_parsers = [NeuronParser(t, f, uses_acts=a, uses_inputs=i) for t, f, a, i in _NEURONS]

import plugin  # noqa: E402  (the plugin module imports names from this one)

layer_parsers.update(plugin.extra_layer_parsers)
neuron_parsers = sorted(neuron_parsers + plugin.extra_neuron_parsers, key=lambda p: p.type)
