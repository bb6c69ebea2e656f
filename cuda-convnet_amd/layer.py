"""Layer-definition / layer-parameter config parser (Python 3).

Reads the reference's two INI files unchanged and produces the list of layer dicts that IS the
model specification handed to `libkernel.initModel` (same keys, same value types, same derived
sizes).  Behaviour follows Python/layer.py of the reference (cited per class); the code is new.

Public surface kept for config / plugin authors (Python/layer.py:37-131, 142-294, 1155-1195):
`LayerParsingError`, `MyConfigParser.safe_get*`, the `*Parser` classes with `parse` /
`add_params` / `optimize`, `layer_parsers`, `neuron_parsers`, `LayerParser.parse_layers`,
`LayerParser.register_layer_parser`.

Deliberate deviations (both make shipped configs usable, neither changes a valid config's result):
  * norm layers default `k` to 1 when the params file has none: the reference requires it
    (layer.py:1083) although no shipped params file provides it and only cmrnorm reads it
    (Kernel/src/layers/layer.cu:904);
  * the data provider is consulted through `model.train_data_provider` exactly as before, but a
    plain object with `get_data_dims(idx)` / `get_num_classes()` is enough.
"""
import configparser as cfg
import os
import re
import sys
from math import ceil, exp, floor
from os import linesep as NL

import numpy as n
import numpy.random as nr


class LayerParsingError(Exception):
    pass


# ------------------------------------------------------------------------------------------
# neuron strings: "relu", "tanh[1.79,-0.66]", "dropout[0.5]" ...        (layer.py:40-90)
# ------------------------------------------------------------------------------------------
class NeuronParser(object):
    def __init__(self, type, func_str, uses_acts=True, uses_inputs=True):
        self.type, self.func_str = type, func_str
        self.uses_acts, self.uses_inputs = uses_acts, uses_inputs

    def _result(self, base, params):
        return {"type": base, "params": params, "usesActs": self.uses_acts, "usesInputs": self.uses_inputs}

    def parse(self, type):
        return self._result(self.type, {}) if type == self.type else None


class ParamNeuronParser(NeuronParser):
    neuron_regex = re.compile(r"^\s*(\w+)\s*\[\s*(\w+(\s*,\w+)*)\s*\]\s*$")

    def __init__(self, type, func_str, uses_acts=True, uses_inputs=True):
        NeuronParser.__init__(self, type, func_str, uses_acts, uses_inputs)
        m = self.neuron_regex.match(type)
        self.base_type = m.group(1)
        self.param_names = m.group(2).split(",")
        assert len(set(self.param_names)) == len(self.param_names)

    def parse(self, type):
        m = re.match(r"^%s\s*\[([\d,\.\s\-e]*)\]\s*$" % self.base_type, type)
        if not m:
            return None
        try:
            vals = [float(v.strip()) for v in m.group(1).split(",")]
        except (TypeError, ValueError):
            return None
        if len(vals) != len(self.param_names):
            return None
        return self._result(self.base_type, dict(zip(self.param_names, vals)))


class AbsTanhNeuronParser(ParamNeuronParser):
    def __init__(self):
        ParamNeuronParser.__init__(self, "abstanh[a,b]", "f(x) = a * |tanh(b * x)|")

    def parse(self, type):
        dic = ParamNeuronParser.parse(self, type)
        if dic:
            dic["params"]["b"] = abs(dic["params"]["b"])
        return dic


# ------------------------------------------------------------------------------------------
# INI access with convnet-flavoured errors                                 (layer.py:93-139)
# ------------------------------------------------------------------------------------------
_BOOL = {"1": True, "yes": True, "true": True, "on": True, "0": False, "no": False, "false": False, "off": False}


class MyConfigParser(cfg.RawConfigParser):
    def __init__(self, *a, **kw):
        kw.setdefault("strict", False)
        kw.pop("dict_type", None)
        cfg.RawConfigParser.__init__(self, *a, **kw)

    def safe_get(self, section, option, f=None, typestr=None, default=None):
        try:
            raw = cfg.RawConfigParser.get(self, section, option)
        except (cfg.NoOptionError, cfg.NoSectionError):
            if default is not None:
                return default
            raise LayerParsingError("Layer '%s': required parameter '%s' missing" % (section, option))
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
        if type(v) == list:
            return v
        try:
            return [f(x.strip()) for x in str(v).split(",")]
        except Exception:
            raise LayerParsingError("Layer '%s': parameter '%s' must be ','-delimited list of %s" % (section, option, typestr))

    def safe_get_int(self, section, option, default=None):
        return self.safe_get(section, option, f=int, typestr="int", default=default)

    def safe_get_float(self, section, option, default=None):
        return self.safe_get(section, option, f=float, typestr="float", default=default)

    def safe_get_bool(self, section, option, default=None):
        return self.safe_get(section, option, f=lambda s: _BOOL[s.strip().lower()], typestr="bool", default=default)

    def safe_get_float_list(self, section, option, default=None):
        return self.safe_get_list(section, option, float, typestr="floats", default=default)

    def safe_get_int_list(self, section, option, default=None):
        return self.safe_get_list(section, option, int, typestr="ints", default=default)

    def safe_get_bool_list(self, section, option, default=None):
        return self.safe_get_list(section, option, lambda x: x.lower() in ("true", "1"), typestr="bools", default=default)


class FakeConfigParser(object):
    """Lets a dict stand in for the INI file when a neuron layer is split off its parent."""

    def __init__(self, dic):
        self.dic = dic

    def safe_get(self, section, option, default=None):
        return self.dic[option]


# ------------------------------------------------------------------------------------------
# base parsers                                                             (layer.py:142-365)
# ------------------------------------------------------------------------------------------
class LayerParser(object):
    def __init__(self):
        self.dic = {}
        self.set_defaults()

    def init(self, dic):
        self.dic = dic
        return self

    def set_defaults(self):
        d = self.dic
        d["outputs"] = 0
        d["parser"] = self
        d["requiresParams"] = False
        d["usesActs"] = True        # layer reads its own output again (e.g. for its gradient)
        d["usesInputs"] = True      # layer reads its inputs again in bprop
        d["forceOwnActs"] = True    # gradient map is not element-wise: needs its own actsGrad
        d["gradConsumer"] = False   # only layers with parameters consume gradient themselves

    def optimize(self, layers):
        self.dic["actsTarget"] = -1
        self.dic["actsGradTarget"] = -1

    def add_params(self, mcp):
        pass

    def parse(self, name, mcp, prev_layers, model=None):
        self.prev_layers = prev_layers
        self.dic["name"] = name
        self.dic["type"] = mcp.safe_get(name, "type")
        return self.dic

    # -- validation helpers ------------------------------------------------------------
    def verify_float_range(self, v, param_name, _min, _max):
        self.verify_num_range(v, param_name, _min, _max, strconv=lambda x: "%.3f" % x)

    def verify_num_range(self, v, param_name, _min, _max, strconv=lambda x: "%d" % x):
        if type(v) == list:
            for i, vv in enumerate(v):
                self._verify_num_range(vv, param_name, _min, _max, i, strconv=strconv)
        else:
            self._verify_num_range(v, param_name, _min, _max, strconv=strconv)

    def _verify_num_range(self, v, param_name, _min, _max, input=-1, strconv=lambda x: "%d" % x):
        who = self.dic["name"] if input < 0 else "%s[%d]" % (self.dic["name"], input)
        if _min is not None and _max is not None and (v < _min or v > _max):
            raise LayerParsingError("Layer '%s': parameter '%s' must be in the range %s-%s" % (who, param_name, strconv(_min), strconv(_max)))
        if _min is not None and v < _min:
            raise LayerParsingError("Layer '%s': parameter '%s' must be greater than or equal to %s" % (who, param_name, strconv(_min)))
        if _max is not None and v > _max:
            raise LayerParsingError("Layer '%s': parameter '%s' must be smaller than or equal to %s" % (who, param_name, strconv(_max)))

    def verify_divisible(self, value, div, value_name, div_name=None, input_idx=0):
        who = self.dic["name"] if len(self.dic["inputs"]) == 0 else "%s[%d]" % (self.dic["name"], input_idx)
        if value % div != 0:
            raise LayerParsingError("Layer '%s': parameter '%s' must be divisible by %s" % (who, value_name, str(div) if div_name is None else "'%s'" % div_name))

    def verify_str_in(self, value, lst):
        if value not in lst:
            raise LayerParsingError("Layer '%s': parameter '%s' must be one of %s" % (self.dic["name"], value, ", ".join("'%s'" % s for s in lst)))

    def verify_int_in(self, value, lst):
        if value not in lst:
            raise LayerParsingError("Layer '%s': parameter '%s' must be one of %s" % (self.dic["name"], value, ", ".join("'%d'" % s for s in lst)))

    # -- whole-file driver ------------------------------------------------------------------
    @staticmethod
    def detach_neuron_layers(layers):
        """`neuron=x` on a layer becomes a separate neuron layer right after it (layer.py:238-244)."""
        out = []
        for i, l in enumerate(layers):
            out.append(l)
            if l["type"] != "neuron" and l.get("neuron"):
                NeuronLayerParser().detach_neuron_layer(i, layers, out)
        return out

    @staticmethod
    def parse_layers(layer_cfg_path, param_cfg_path, model, layers=None):
        """First run: read the topology, then the params.  With `layers` given (a checkpoint), only
        the params file is re-read so learning rates can change between runs (layer.py:247-288)."""
        layers = [] if layers is None else layers
        try:
            if not os.path.exists(layer_cfg_path):
                raise LayerParsingError("Layer definition file '%s' does not exist" % layer_cfg_path)
            if not os.path.exists(param_cfg_path):
                raise LayerParsingError("Layer parameter file '%s' does not exist" % param_cfg_path)
            if len(layers) == 0:
                mcp = MyConfigParser()
                mcp.read([layer_cfg_path])
                for name in mcp.sections():
                    if not mcp.has_option(name, "type"):
                        raise LayerParsingError("Layer '%s': no type given" % name)
                    ltype = mcp.safe_get(name, "type")
                    if ltype not in layer_parsers:
                        raise LayerParsingError("Layer '%s': Unknown layer type: '%s'" % (name, ltype))
                    layers.append(layer_parsers[ltype]().parse(name, mcp, layers, model))

                layers = LayerParser.detach_neuron_layers(layers)
                for l in layers:
                    l["parser"].optimize(layers)
                    del l["parser"]

                for l in layers:
                    if not l["type"].startswith("cost."):
                        used = any(l["name"] in [layers[i]["name"] for i in l2["inputs"]] for l2 in layers if "inputs" in l2)
                        if not used:
                            raise LayerParsingError("Layer '%s' of type '%s' is unused" % (l["name"], l["type"]))

            mcp = MyConfigParser()
            mcp.read([param_cfg_path])
            conserve = model.op.get_value("conserve_mem") if getattr(model, "op", None) is not None else 0
            for l in layers:
                if not mcp.has_section(l["name"]) and l["requiresParams"]:
                    raise LayerParsingError("Layer '%s' of type '%s' requires extra parameters, but none given in file '%s'." % (l["name"], l["type"], param_cfg_path))
                lp = layer_parsers[l["type"]]().init(l)
                lp.add_params(mcp)
                lp.dic["conserveMem"] = conserve
        except LayerParsingError as e:
            print(e)
            sys.exit(1)
        return layers

    @staticmethod
    def register_layer_parser(ltype, cls):
        if ltype in layer_parsers:
            raise LayerParsingError("Layer type '%s' already registered" % ltype)
        layer_parsers[ltype] = cls


class LayerWithInputParser(LayerParser):
    def __init__(self, num_inputs=-1):
        LayerParser.__init__(self)
        self.num_inputs = num_inputs

    def verify_num_params(self, params):
        for param in params:
            if len(self.dic[param]) != len(self.dic["inputs"]):
                raise LayerParsingError("Layer '%s': %s list length does not match number of inputs" % (self.dic["name"], param))

    def optimize(self, layers):
        """Activity / gradient matrix sharing with a sole-consumer input (layer.py:307-323)."""
        LayerParser.optimize(self, layers)
        dic = self.dic
        if dic["forceOwnActs"]:
            return
        for i, inp in enumerate(dic["inputs"]):
            l = layers[inp]
            consumers = sum("inputs" in ll and inp in ll["inputs"] for ll in layers)
            if l["outputs"] == dic["outputs"] and consumers == 1:
                if not l["usesActs"] and not dic["usesInputs"]:
                    dic["actsTarget"] = i
                dic["actsGradTarget"] = i

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerParser.parse(self, name, mcp, prev_layers, model)
        names = [inp.strip() for inp in mcp.safe_get(name, "inputs").split(",")]
        prev_names = [p["name"] for p in prev_layers]
        for inp in names:
            if inp not in prev_names:
                raise LayerParsingError("Layer '%s': input layer '%s' not defined" % (name, inp))
        dic["inputs"] = [prev_names.index(inp) for inp in names]
        dic["inputLayers"] = [prev_layers[i] for i in dic["inputs"]]
        for i in dic["inputs"]:
            if prev_layers[i]["outputs"] == 0:
                raise LayerParsingError("Layer '%s': input layer '%s' does not produce any output" % (name, prev_names[i]))
        dic["numInputs"] = [prev_layers[i]["outputs"] for i in dic["inputs"]]
        dic["neuron"] = mcp.safe_get(name, "neuron", default="")
        if self.num_inputs > 0 and len(dic["numInputs"]) != self.num_inputs:
            raise LayerParsingError("Layer '%s': number of inputs must be %d" % (name, self.num_inputs))
        return dic

    def _square_image(self, channels):
        """imgPixels / imgSize of a (channels, size, size) input."""
        dic = self.dic
        dic["imgPixels"] = dic["numInputs"][0] // channels
        dic["imgSize"] = int(n.sqrt(dic["imgPixels"]))

    def verify_img_size(self):
        dic = self.dic
        if dic["numInputs"][0] % dic["imgPixels"] != 0 or dic["imgSize"] * dic["imgSize"] != dic["imgPixels"]:
            raise LayerParsingError("Layer '%s': has %-d dimensional input, not interpretable as %d-channel images" % (dic["name"], dic["numInputs"][0], dic["channels"]))

    @staticmethod
    def grad_consumers_below(dic):
        if dic["gradConsumer"]:
            return True
        if "inputLayers" in dic:
            return any(LayerWithInputParser.grad_consumers_below(l) for l in dic["inputLayers"])
        return False

    def verify_no_grads(self):
        if LayerWithInputParser.grad_consumers_below(self.dic):
            raise LayerParsingError("Layer '%s': layers of type '%s' cannot propagate gradient and must not be placed over layers with parameters." % (self.dic["name"], self.dic["type"]))

    def _elementwise(self):
        self.dic["forceOwnActs"] = False
        self.dic["usesActs"] = False
        self.dic["usesInputs"] = False


# ------------------------------------------------------------------------------------------
# image-op layers without parameters                                       (layer.py:367-530)
# ------------------------------------------------------------------------------------------
class NailbedLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        self._elementwise()
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["stride"] = mcp.safe_get_int(name, "stride")
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self._square_image(dic["channels"])
        dic["outputsX"] = (dic["imgSize"] + dic["stride"] - 1) // dic["stride"]
        dic["start"] = (dic["imgSize"] - dic["stride"] * (dic["outputsX"] - 1)) // 2
        dic["outputs"] = dic["channels"] * dic["outputsX"] ** 2
        self.verify_num_range(dic["outputsX"], "outputsX", 0, None)
        self.verify_img_size()
        print("Initialized bed-of-nails layer '%s', producing %dx%d %d-channel output" % (name, dic["outputsX"], dic["outputsX"], dic["channels"]))
        return dic


class GaussianBlurLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        self._elementwise()
        dic["outputs"] = dic["numInputs"][0]
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["filterSize"] = mcp.safe_get_int(name, "filterSize")
        dic["stdev"] = mcp.safe_get_float(name, "stdev")
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self.verify_int_in(dic["filterSize"], [3, 5, 7, 9])
        self._square_image(dic["channels"])
        half = dic["filterSize"] // 2
        taps = [exp(-(half - i) ** 2 / float(2 * dic["stdev"] ** 2)) for i in range(dic["filterSize"])]
        dic["filter"] = n.array(taps, dtype=n.float32).reshape(1, dic["filterSize"])
        dic["filter"] /= dic["filter"].sum()
        self.verify_img_size()
        if dic["filterSize"] > dic["imgSize"]:
            raise LayerParsingError("Later '%s': filter size (%d) must be smaller than image size (%d)." % (dic["name"], dic["filterSize"], dic["imgSize"]))
        print("Initialized Gaussian blur layer '%s', producing %dx%d %d-channel output" % (name, dic["imgSize"], dic["imgSize"], dic["channels"]))
        return dic


class ResizeLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        self._elementwise()
        dic["channels"] = mcp.safe_get_int(name, "channels")
        self._square_image(dic["channels"])
        dic["scale"] = mcp.safe_get_float(name, "scale")
        dic["tgtSize"] = int(floor(dic["imgSize"] / dic["scale"]))
        dic["tgtPixels"] = dic["tgtSize"] ** 2
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self.verify_float_range(dic["scale"], "scale", 0.5, 2)
        dic["outputs"] = dic["channels"] * dic["tgtPixels"]
        self.verify_img_size()
        self.verify_no_grads()
        print("Initialized resize layer '%s', producing %dx%d %d-channel output" % (name, dic["tgtSize"], dic["tgtSize"], dic["channels"]))
        return dic


class RandomScaleLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        self._elementwise()
        dic["channels"] = mcp.safe_get_int(name, "channels")
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self._square_image(dic["channels"])
        dic["maxScale"] = mcp.safe_get_float(name, "maxScale")
        dic["tgtSize"] = int(floor(dic["imgSize"] / dic["maxScale"]))
        dic["tgtPixels"] = dic["tgtSize"] ** 2
        self.verify_float_range(dic["maxScale"], "maxScale", 1, 2)
        dic["outputs"] = dic["channels"] * dic["tgtPixels"]
        self.verify_img_size()
        self.verify_no_grads()
        print("Initialized random scale layer '%s', producing %dx%d %d-channel output" % (name, dic["tgtSize"], dic["tgtSize"], dic["channels"]))
        return dic


class ColorTransformLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        self._elementwise()
        self._square_image(3)
        dic["channels"] = 3
        dic["outputs"] = dic["numInputs"][0]
        self.verify_img_size()
        self.verify_no_grads()
        return dic


class RGBToYUVLayerParser(ColorTransformLayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = ColorTransformLayerParser.parse(self, name, mcp, prev_layers, model)
        print("Initialized RGB --> YUV layer '%s', producing %dx%d %d-channel output" % (name, dic["imgSize"], dic["imgSize"], dic["channels"]))
        return dic


class RGBToLABLayerParser(ColorTransformLayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = ColorTransformLayerParser.parse(self, name, mcp, prev_layers, model)
        dic["center"] = mcp.safe_get_bool(name, "center", default=False)
        print("Initialized RGB --> LAB layer '%s', producing %dx%d %d-channel output" % (name, dic["imgSize"], dic["imgSize"], dic["channels"]))
        return dic


# ------------------------------------------------------------------------------------------
# neuron / element-wise layers                                             (layer.py:532-626)
# ------------------------------------------------------------------------------------------
class NeuronLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    @staticmethod
    def get_unused_layer_name(layers, wish):
        taken = set(l["name"] for l in layers)
        if wish not in taken:
            return wish
        for i in range(1, 100):
            name = "%s.%d" % (wish, i)
            if name not in taken:
                return name
        raise LayerParsingError("This is insane.")

    def parse_neuron(self, neuron_str):
        for p in neuron_parsers:
            res = p.parse(neuron_str)
            if res:
                self.dic["neuron"] = res
                self.dic["usesActs"] = res["usesActs"]
                self.dic["usesInputs"] = res["usesInputs"]
                return
        width = max(len("Neuron type"), max(len(p.type) for p in neuron_parsers)) + 2
        table = NL.join(p.type.ljust(width) + p.func_str for p in neuron_parsers)
        raise LayerParsingError("Layer '%s': unable to parse neuron type '%s'. Valid neuron types: %s%s%s%s%sWhere neurons have parameters, they must be floats."
                                % (self.dic["name"], neuron_str, NL, "Neuron type".ljust(width) + "Function", NL, table, NL))

    def detach_neuron_layer(self, idx, layers, layers_new):
        dic = self.dic
        self.set_defaults()
        dic["name"] = NeuronLayerParser.get_unused_layer_name(layers, "%s_neuron" % layers[idx]["name"])
        dic["type"] = "neuron"
        dic["inputs"] = layers[idx]["name"]
        dic["neuron"] = layers[idx]["neuron"]
        dic = self.parse(dic["name"], FakeConfigParser(dic), layers_new)
        # layers defined after the parent now sit one slot further down
        for l in layers[idx + 1:]:
            if "inputs" in l:
                l["inputs"] = [i + (i >= len(layers_new) - 1) for i in l["inputs"]]
            if "weightSourceLayerIndices" in l:
                l["weightSourceLayerIndices"] = [i + (i >= len(layers_new)) for i in l["weightSourceLayerIndices"]]
        layers_new.append(dic)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["outputs"] = dic["numInputs"][0]
        self.parse_neuron(dic["neuron"])
        dic["forceOwnActs"] = False
        print("Initialized neuron layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


class EltwiseSumLayerParser(LayerWithInputParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        if len(set(dic["numInputs"])) != 1:
            raise LayerParsingError("Layer '%s': all inputs must have the same dimensionality. Got dimensionalities: %s" % (name, ", ".join(str(s) for s in dic["numInputs"])))
        dic["outputs"] = dic["numInputs"][0]
        self._elementwise()
        dic["coeffs"] = mcp.safe_get_float_list(name, "coeffs", default=[1.0] * len(dic["inputs"]))
        print("Initialized elementwise sum layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


class EltwiseMaxLayerParser(LayerWithInputParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        if len(dic["inputs"]) < 2:
            raise LayerParsingError("Layer '%s': elementwise max layer must have at least 2 inputs, got %d." % (name, len(dic["inputs"])))
        if len(set(dic["numInputs"])) != 1:
            raise LayerParsingError("Layer '%s': all inputs must have the same dimensionality. Got dimensionalities: %s" % (name, ", ".join(str(s) for s in dic["numInputs"])))
        dic["outputs"] = dic["numInputs"][0]
        print("Initialized elementwise max layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


# ------------------------------------------------------------------------------------------
# layers with weights                                                      (layer.py:628-1007)
# ------------------------------------------------------------------------------------------
class WeightLayerParser(LayerWithInputParser):
    LAYER_PAT = re.compile(r"^\s*([^\s\[]+)(?:\[(\d+)\])?\s*$")   # layername or layername[3]

    @staticmethod
    def get_layer_name(name_str):
        m = WeightLayerParser.LAYER_PAT.match(name_str)
        return (m.group(1), m.group(2)) if m else None

    def add_params(self, mcp):
        dic, name = self.dic, self.dic["name"]
        dic["epsW"] = mcp.safe_get_float_list(name, "epsW")
        dic["epsB"] = mcp.safe_get_float(name, "epsB")
        dic["momW"] = mcp.safe_get_float_list(name, "momW")
        dic["momB"] = mcp.safe_get_float(name, "momB")
        dic["wc"] = mcp.safe_get_float_list(name, "wc")
        self.verify_num_params(["epsW", "momW", "wc"])
        dic["gradConsumer"] = dic["epsB"] > 0 or any(w > 0 for w in dic["epsW"])

    @staticmethod
    def unshare_weights(layer, layers, matrix_idx=None):
        if "weightSourceLayerIndices" not in layer:
            return
        for i in (range(len(layer["inputs"])) if matrix_idx is None else [int(matrix_idx)]):
            src = layer["weightSourceLayerIndices"][i]
            if src >= 0:
                src_name, src_idx = layers[src]["name"], layer["weightSourceMatrixIndices"][i]
                layer["weightSourceLayerIndices"][i] = -1
                layer["weightSourceMatrixIndices"][i] = -1
                layer["weights"][i] = layer["weights"][i].copy()
                layer["weightsInc"][i] = n.zeros_like(layer["weights"][i])
                print("Unshared weight matrix %s[%d] from %s[%d]." % (layer["name"], i, src_name, src_idx))
            else:
                print("Weight matrix %s[%d] already unshared." % (layer["name"], i))

    def call_init_func(self, param_name, shapes, input_idx=-1):
        """initWFunc / initBFunc / initCFunc = module.function(p1,p2,...) (layer.py:672-684)."""
        dic = self.dic
        m = re.match(r"^([^\.]+)\.([^\(\)]+)\s*(?:\(([^,]+(?:,[^,]+)*)\))?$", dic[param_name])
        if not m:
            raise LayerParsingError("Layer '%s': '%s' parameter must have format 'moduleName.functionName(param1,param2,...)'; got: %s." % (dic["name"], param_name, dic[param_name]))
        module, func = m.group(1), m.group(2)
        params = m.group(3).split(",") if m.group(3) is not None else []
        try:
            fn = getattr(__import__(module), func)
            if input_idx >= 0:
                return fn(dic["name"], input_idx, shapes, params=params)
            return fn(dic["name"], shapes, params=params)
        except (ImportError, AttributeError, TypeError) as e:
            raise LayerParsingError("Layer '%s': %s." % (dic["name"], e))

    def _checked_init(self, what, arr, shape, fname, idx=None):
        who = "%s[%d]" % (self.dic["name"], idx) if idx is not None else self.dic["name"]
        if type(arr) != n.ndarray:
            raise LayerParsingError("Layer '%s': %s initialization function %s must return numpy.ndarray object. Got: %s." % (who, what, fname, type(arr)))
        if arr.dtype != n.float32:
            raise LayerParsingError("Layer '%s': %s initialization function %s must return single-precision floats. Got: %s." % (who, what, fname, arr.dtype))
        if arr.shape != shape:
            raise LayerParsingError("Layer '%s': %s returned by initialization function %s has wrong shape. Should be: %s; got: %s." % (who, what, fname, shape, arr.shape))

    def _source_layer(self, i):
        idx = self.dic["weightSourceLayerIndices"][i]
        return self.prev_layers[idx] if idx < len(self.prev_layers) else self.dic

    def make_weights(self, initW, rows, cols, order="C"):
        dic = self.dic
        dic["weights"], dic["weightsInc"] = [], []
        for i in range(len(dic["inputs"])):
            shape = (rows[i], cols[i])
            if dic["initWFunc"]:
                w = self.call_init_func("initWFunc", shape, input_idx=i)
                self._checked_init("weight matrix", w, shape, dic["initWFunc"], i)
                dic["weights"].append(n.require(w, requirements=order))
                dic["weightsInc"].append(n.zeros_like(dic["weights"][i]))
                print("Layer '%s[%d]' initialized weight matrices from function %s" % (dic["name"], i, dic["initWFunc"]))
            elif dic["weightSourceLayerIndices"][i] >= 0:    # same ndarray objects as the source: shared
                src, mi = self._source_layer(i), dic["weightSourceMatrixIndices"][i]
                dic["weights"].append(src["weights"][mi])
                dic["weightsInc"].append(src["weightsInc"][mi])
                if dic["weights"][i].shape != shape:
                    raise LayerParsingError("Layer '%s': weight sharing source matrix '%s' has shape %dx%d; should be %dx%d."
                                            % (dic["name"], dic["weightSource"][i], dic["weights"][i].shape[0], dic["weights"][i].shape[1], rows[i], cols[i]))
                print("Layer '%s' initialized weight matrix %d from %s" % (dic["name"], i, dic["weightSource"][i]))
            else:
                dic["weights"].append(n.array(initW[i] * nr.randn(rows[i], cols[i]), dtype=n.single, order=order))
                dic["weightsInc"].append(n.zeros_like(dic["weights"][i]))

    def make_biases(self, rows, cols, order="C"):
        dic = self.dic
        if dic["initBFunc"]:
            b = self.call_init_func("initBFunc", (rows, cols))
            self._checked_init("bias vector", b, (rows, cols), dic["initBFunc"])
            dic["biases"] = n.require(b, requirements=order)
            print("Layer '%s' initialized bias vector from function %s" % (dic["name"], dic["initBFunc"]))
        else:
            dic["biases"] = dic["initB"] * n.ones((rows, cols), order="C", dtype=n.single)
        dic["biasesInc"] = n.zeros_like(dic["biases"])

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["requiresParams"] = True
        dic["gradConsumer"] = True
        dic["initW"] = mcp.safe_get_float_list(name, "initW", default=0.01)
        dic["initB"] = mcp.safe_get_float(name, "initB", default=0)
        dic["initWFunc"] = mcp.safe_get(name, "initWFunc", default="")
        dic["initBFunc"] = mcp.safe_get(name, "initBFunc", default="")
        dic["weightSource"] = mcp.safe_get_list(name, "weightSource", default=[""] * len(dic["inputs"]))
        if type(dic["initW"]) != list:
            dic["initW"] = [dic["initW"]] * len(dic["inputs"])
        self.verify_num_params(["initW", "weightSource"])

        prev_names = [p["name"] for p in prev_layers]
        dic["weightSourceLayerIndices"], dic["weightSourceMatrixIndices"] = [], []
        for i, src_name in enumerate(dic["weightSource"]):
            li = mi = -1
            if src_name != "":
                parsed = WeightLayerParser.get_layer_name(src_name)
                if parsed is None:
                    raise LayerParsingError("Layer '%s': unable to parse weight sharing source '%s'. Format is layer[idx] or just layer, in which case idx=0 is used." % (name, src_name))
                lname = parsed[0]
                mi = int(parsed[1]) if parsed[1] is not None else 0
                if lname != name and lname not in prev_names:
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' does not exist." % (name, lname))
                li = prev_names.index(lname) if lname != name else len(prev_names)
                src = prev_layers[li] if lname != name else dic
                if src["type"] != dic["type"]:
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' is of type '%s'; should be '%s'." % (name, lname, src["type"], dic["type"]))
                if lname != name and len(src["weights"]) <= mi:
                    raise LayerParsingError("Layer '%s': weight sharing source layer '%s' has %d weight matrices, but '%s[%d]' requested." % (name, lname, len(src["weights"]), src_name, mi))
                if lname == name and mi >= i:
                    raise LayerParsingError("Layer '%s': weight sharing source '%s[%d]' not defined yet." % (name, name, mi))
            dic["weightSourceLayerIndices"].append(li)
            dic["weightSourceMatrixIndices"].append(mi)
        return dic


class FCLayerParser(WeightLayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = WeightLayerParser.parse(self, name, mcp, prev_layers, model)
        dic["usesActs"] = False
        dic["outputs"] = mcp.safe_get_int(name, "outputs")
        self.verify_num_range(dic["outputs"], "outputs", 1, None)
        self.make_weights(dic["initW"], dic["numInputs"], [dic["outputs"]] * len(dic["numInputs"]), order="F")
        self.make_biases(1, dic["outputs"], order="F")
        print("Initialized fully-connected layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


class LocalLayerParser(WeightLayerParser):
    @staticmethod
    def conv_to_local(layers, idx):
        """Turn a conv layer (and everything sharing its weights) into an unshared local layer by
        tiling the filter bank once per module (layer.py:816-844)."""
        layer = layers[idx]
        if layer["type"] != "conv":
            return layer
        layer["type"] = "local"
        for inp in range(len(layer["inputs"])):
            src_idx = layer["weightSourceLayerIndices"][inp]
            if src_idx >= 0:
                LocalLayerParser.conv_to_local(layers, src_idx)
                for w in ("weights", "weightsInc"):
                    layer[w][inp] = layers[src_idx][w][layer["weightSourceMatrixIndices"][inp]]
            else:
                w = layer["weights"][inp]
                tiled = n.tile(w.reshape(1, w.size), (layer["modules"], 1))
                layer["weights"][inp] = n.require(tiled.reshape(layer["modules"] * layer["filterChannels"][inp] * layer["filterPixels"][inp], layer["filters"]), requirements="C")
                layer["weightsInc"][inp] = n.zeros_like(layer["weights"][inp])
        if layer["sharedBiases"]:
            layer["biases"] = n.require(n.repeat(layer["biases"], layer["modules"], axis=0), requirements="C")
            layer["biasesInc"] = n.zeros_like(layer["biases"])
        print("Converted layer '%s' from convolutional to unshared, locally-connected" % layer["name"])
        for i, l in enumerate(layers):
            if "weightSourceLayerIndices" in l and idx in l["weightSourceLayerIndices"]:
                LocalLayerParser.conv_to_local(layers, i)
        return layer

    def gen_rand_conns(self, groups, channels, filterChannels, inputIdx):
        """(groups*filterChannels) channel indices: overSample concatenated permutations (layer.py:848-871)."""
        dic = self.dic
        overSample = groups * filterChannels // channels
        conns = [int(x) for _ in range(overSample) for x in nr.permutation(range(channels))]
        if dic["initCFunc"]:
            conns = self.call_init_func("initCFunc", (groups, channels, filterChannels), input_idx=inputIdx)
            who = "%s[%d]" % (dic["name"], inputIdx)
            if len(conns) != overSample * channels:
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return list of length <groups> * <filterChannels> = %d; got: %d" % (who, dic["initCFunc"], overSample * channels, len(conns)))
            if any(c not in range(channels) for c in conns):
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return list of channel indices in the range 0-<channels-1> = 0-%d." % (who, dic["initCFunc"], channels - 1))
            if any(len(set(c)) != len(c) for c in [conns[o * channels:(o + 1) * channels] for o in range(overSample)]):
                raise LayerParsingError("Layer '%s': random connectivity initialization function %s must return a permutation of 0-%d in every block of %d elements." % (who, dic["initCFunc"], channels - 1, channels))
        elif dic["weightSourceLayerIndices"][inputIdx] >= 0:
            src, src_inp = self._source_layer(inputIdx), dic["weightSourceMatrixIndices"][inputIdx]
            if not src.get("randSparse") or not src["randSparse"][src_inp]:
                raise LayerParsingError("Layer '%s[%d]': randSparse is true in this layer but false in weight sharing source layer '%s[%d]'." % (dic["name"], inputIdx, src["name"], src_inp))
            mine = (groups, channels, filterChannels)
            theirs = (src["groups"][src_inp], src["channels"][src_inp], src["filterChannels"][src_inp])
            if mine != theirs:
                raise LayerParsingError("Layer '%s[%d]': groups, channels, filterChannels set to %d, %d, %d, respectively. Does not match setting in weight sharing source layer '%s[%d]': %d, %d, %d." % ((dic["name"], inputIdx) + mine + (src["name"], src_inp) + theirs))
            conns = src["filterConns"][src_inp]
        return conns

    def parse(self, name, mcp, prev_layers, model=None):
        dic = WeightLayerParser.parse(self, name, mcp, prev_layers, model)
        dic["requiresParams"] = True
        dic["usesActs"] = False
        ni = len(dic["inputs"])
        dic["channels"] = mcp.safe_get_int_list(name, "channels")
        dic["padding"] = mcp.safe_get_int_list(name, "padding", default=[0] * ni)
        dic["stride"] = mcp.safe_get_int_list(name, "stride", default=[1] * ni)
        dic["filterSize"] = mcp.safe_get_int_list(name, "filterSize")
        dic["filters"] = mcp.safe_get_int_list(name, "filters")
        dic["groups"] = mcp.safe_get_int_list(name, "groups", default=[1] * ni)
        dic["randSparse"] = mcp.safe_get_bool_list(name, "randSparse", default=[False] * ni)
        dic["initW"] = mcp.safe_get_float_list(name, "initW")
        dic["initCFunc"] = mcp.safe_get(name, "initCFunc", default="")
        self.verify_num_params(["channels", "padding", "stride", "filterSize", "filters", "groups", "randSparse", "initW"])
        self.verify_num_range(dic["stride"], "stride", 1, None)
        self.verify_num_range(dic["filterSize"], "filterSize", 1, None)
        self.verify_num_range(dic["padding"], "padding", 0, None)
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self.verify_num_range(dic["groups"], "groups", 1, None)

        dic["imgPixels"] = [ni_ // c for ni_, c in zip(dic["numInputs"], dic["channels"])]
        dic["imgSize"] = [int(n.sqrt(p)) for p in dic["imgPixels"]]
        self.verify_num_range(dic["imgSize"], "imgSize", 1, None)
        dic["filters"] = [f * g for f, g in zip(dic["filters"], dic["groups"])]       # `filters` is per group
        dic["filterPixels"] = [fs ** 2 for fs in dic["filterSize"]]
        dic["modulesX"] = [1 + int(ceil((2 * p + s - fs) / float(st)))
                           for p, s, fs, st in zip(dic["padding"], dic["imgSize"], dic["filterSize"], dic["stride"])]
        dic["filterChannels"] = [c // g for c, g in zip(dic["channels"], dic["groups"])]
        if max(dic["randSparse"]):
            dic["filterChannels"] = mcp.safe_get_int_list(name, "filterChannels", default=dic["filterChannels"])
            self.verify_num_params(["filterChannels"])
        if len(set(dic["modulesX"])) != 1 or len(set(dic["filters"])) != 1:
            raise LayerParsingError("Layer '%s': all inputs must produce equally-dimensioned output. Dimensions are: %s."
                                    % (name, ", ".join("%dx%dx%d" % (f, mx, mx) for f, mx in zip(dic["filters"], dic["modulesX"]))))
        dic["modulesX"] = dic["modulesX"][0]
        dic["modules"] = dic["modulesX"] ** 2
        dic["filters"] = dic["filters"][0]
        dic["outputs"] = dic["modules"] * dic["filters"]
        dic["filterConns"] = [[]] * ni

        for i in range(ni):
            if dic["numInputs"][i] % dic["imgPixels"][i] != 0 or dic["imgSize"][i] ** 2 != dic["imgPixels"][i]:
                raise LayerParsingError("Layer '%s[%d]': has %-d dimensional input, not interpretable as square %d-channel images" % (name, i, dic["numInputs"][i], dic["channels"][i]))
            if dic["channels"][i] > 3 and dic["channels"][i] % 4 != 0:
                raise LayerParsingError("Layer '%s[%d]': number of channels must be smaller than 4 or divisible by 4" % (name, i))
            if dic["filterSize"][i] > 2 * dic["padding"][i] + dic["imgSize"][i]:
                raise LayerParsingError("Layer '%s[%d]': filter size (%d) greater than image size + 2 * padding (%d)" % (name, i, dic["filterSize"][i], 2 * dic["padding"][i] + dic["imgSize"][i]))
            if dic["randSparse"][i]:
                if dic["groups"][i] == 1:
                    raise LayerParsingError("Layer '%s[%d]': number of groups must be greater than 1 when using random sparse connectivity" % (name, i))
                self.verify_divisible(dic["channels"][i], dic["filterChannels"][i], "channels", "filterChannels", input_idx=i)
                self.verify_divisible(dic["filterChannels"][i], 4, "filterChannels", input_idx=i)
                self.verify_divisible(dic["groups"][i] * dic["filterChannels"][i], dic["channels"][i], "groups * filterChannels", "channels", input_idx=i)
                dic["filterConns"][i] = self.gen_rand_conns(dic["groups"][i], dic["channels"][i], dic["filterChannels"][i], i)
            else:
                if dic["groups"][i] > 1:
                    self.verify_divisible(dic["channels"][i], 4 * dic["groups"][i], "channels", "4 * groups", input_idx=i)
                self.verify_divisible(dic["channels"][i], dic["groups"][i], "channels", "groups", input_idx=i)
            self.verify_divisible(dic["filters"], dic["groups"][i], "filters * groups", input_idx=i)
            dic["padding"][i] = -dic["padding"][i]     # the kernels take paddingStart <= 0
        dic["overSample"] = [g * fc // c for g, fc, c in zip(dic["groups"], dic["filterChannels"], dic["channels"])]
        return dic

    def _patch_sizes(self):
        return [fp * fc for fp, fc in zip(self.dic["filterPixels"], self.dic["filterChannels"])]


class ConvLayerParser(LocalLayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = LocalLayerParser.parse(self, name, mcp, prev_layers, model)
        dic["partialSum"] = mcp.safe_get_int(name, "partialSum")
        dic["sharedBiases"] = mcp.safe_get_bool(name, "sharedBiases", default=True)
        if dic["partialSum"] != 0 and dic["modules"] % dic["partialSum"] != 0:
            raise LayerParsingError("Layer '%s': convolutional layer produces %dx%d=%d outputs per filter, but given partialSum parameter (%d) does not divide this number"
                                    % (name, dic["modulesX"], dic["modulesX"], dic["modules"], dic["partialSum"]))
        num_biases = dic["filters"] if dic["sharedBiases"] else dic["modules"] * dic["filters"]
        self.make_weights(dic["initW"], self._patch_sizes(), [dic["filters"]] * len(dic["inputs"]), order="C")
        self.make_biases(num_biases, 1, order="C")
        print("Initialized convolutional layer '%s', producing %dx%d %d-channel output" % (name, dic["modulesX"], dic["modulesX"], dic["filters"]))
        return dic


class LocalUnsharedLayerParser(LocalLayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = LocalLayerParser.parse(self, name, mcp, prev_layers, model)
        rows = [dic["modules"] * p for p in self._patch_sizes()]      # one filter bank per module
        self.make_weights(dic["initW"], rows, [dic["filters"]] * len(dic["inputs"]), order="C")
        self.make_biases(dic["modules"] * dic["filters"], 1, order="C")
        print("Initialized locally-connected layer '%s', producing %dx%d %d-channel output" % (name, dic["modulesX"], dic["modulesX"], dic["filters"]))
        return dic


# ------------------------------------------------------------------------------------------
# data / softmax / pool / norm / cost                                      (layer.py:1009-1153)
# ------------------------------------------------------------------------------------------
class DataLayerParser(LayerParser):
    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerParser.parse(self, name, mcp, prev_layers, model)
        dic["dataIdx"] = mcp.safe_get_int(name, "dataIdx")
        dic["outputs"] = model.train_data_provider.get_data_dims(idx=dic["dataIdx"])
        print("Initialized data layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


class SoftmaxLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["outputs"] = prev_layers[dic["inputs"][0]]["outputs"]
        print("Initialized softmax layer '%s', producing %d outputs" % (name, dic["outputs"]))
        return dic


class PoolLayerParser(LayerWithInputParser):
    def __init__(self):
        LayerWithInputParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["sizeX"] = mcp.safe_get_int(name, "sizeX")
        dic["start"] = mcp.safe_get_int(name, "start", default=0)
        dic["stride"] = mcp.safe_get_int(name, "stride")
        dic["outputsX"] = mcp.safe_get_int(name, "outputsX", default=0)
        dic["pool"] = mcp.safe_get(name, "pool")
        # The reference compares the literal 'pool' with 'avg' here (layer.py:1045-1046), so both
        # flags are always True; kept, because they decide the actsTarget sharing the runtime sees.
        dic["usesActs"] = True
        dic["usesInputs"] = True
        self._square_image(dic["channels"])
        self.verify_num_range(dic["sizeX"], "sizeX", 1, dic["imgSize"])
        self.verify_num_range(dic["stride"], "stride", 1, dic["sizeX"])
        self.verify_num_range(dic["outputsX"], "outputsX", 0, None)
        self.verify_num_range(dic["channels"], "channels", 1, None)
        self.verify_str_in(dic["pool"], ["max", "avg"])
        self.verify_img_size()
        if dic["outputsX"] <= 0:
            dic["outputsX"] = int(ceil((dic["imgSize"] - dic["start"] - dic["sizeX"]) / float(dic["stride"]))) + 1
        dic["outputs"] = dic["outputsX"] ** 2 * dic["channels"]
        print("Initialized %s-pooling layer '%s', producing %dx%d %d-channel output" % (dic["pool"], name, dic["outputsX"], dic["outputsX"], dic["channels"]))
        return dic


class NormLayerParser(LayerWithInputParser):
    RESPONSE_NORM = "response"
    CONTRAST_NORM = "contrast"
    CROSSMAP_RESPONSE_NORM = "cross-map response"

    def __init__(self, norm_type):
        LayerWithInputParser.__init__(self, num_inputs=1)
        self.norm_type = norm_type

    def add_params(self, mcp):
        dic, name = self.dic, self.dic["name"]
        dic["scale"] = mcp.safe_get_float(name, "scale")
        dic["scale"] /= dic["size"] if self.norm_type == self.CROSSMAP_RESPONSE_NORM else dic["size"] ** 2
        dic["pow"] = mcp.safe_get_float(name, "pow")
        dic["k"] = mcp.safe_get_float(name, "k", default=1.0)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["requiresParams"] = True
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["size"] = mcp.safe_get_int(name, "size")
        dic["blocked"] = mcp.safe_get_bool(name, "blocked", default=False)
        self._square_image(dic["channels"])
        dic["usesInputs"] = self.norm_type != self.CONTRAST_NORM
        self.verify_num_range(dic["channels"], "channels", 1, None)
        if self.norm_type == self.CROSSMAP_RESPONSE_NORM:
            self.verify_num_range(dic["size"], "size", 2, dic["channels"])
            if dic["channels"] % 16 != 0:
                raise LayerParsingError("Layer '%s': number of channels must be divisible by 16 when using crossMap" % name)
        else:
            self.verify_num_range(dic["size"], "size", 1, dic["imgSize"])
            if dic["channels"] > 3 and dic["channels"] % 4 != 0:
                raise LayerParsingError("Layer '%s': number of channels must be smaller than 4 or divisible by 4" % name)
        self.verify_img_size()
        dic["outputs"] = dic["imgPixels"] * dic["channels"]
        print("Initialized %s-normalization layer '%s', producing %dx%d %d-channel output" % (self.norm_type, name, dic["imgSize"], dic["imgSize"], dic["channels"]))
        return dic


class CostParser(LayerWithInputParser):
    def __init__(self, num_inputs=-1):
        LayerWithInputParser.__init__(self, num_inputs=num_inputs)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = LayerWithInputParser.parse(self, name, mcp, prev_layers, model)
        dic["requiresParams"] = True
        del dic["neuron"]
        return dic

    def add_params(self, mcp):
        self.dic["coeff"] = mcp.safe_get_float(self.dic["name"], "coeff")


class LogregCostParser(CostParser):
    def __init__(self):
        CostParser.__init__(self, num_inputs=2)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = CostParser.parse(self, name, mcp, prev_layers, model)
        if dic["numInputs"][0] != 1:
            raise LayerParsingError("Layer '%s': dimensionality of first input must be 1" % name)
        if prev_layers[dic["inputs"][1]]["type"] != "softmax":
            raise LayerParsingError("Layer '%s': second input must be softmax layer" % name)
        classes = model.train_data_provider.get_num_classes()
        if dic["numInputs"][1] != classes:
            raise LayerParsingError("Layer '%s': softmax input '%s' must produce %d outputs, because that is the number of classes in the dataset"
                                    % (name, prev_layers[dic["inputs"][1]]["name"], classes))
        print("Initialized logistic regression cost '%s'" % name)
        return dic


class SumOfSquaresCostParser(CostParser):
    def __init__(self):
        CostParser.__init__(self, num_inputs=1)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = CostParser.parse(self, name, mcp, prev_layers, model)
        print("Initialized sum-of-squares cost '%s'" % name)
        return dic


# ------------------------------------------------------------------------------------------
# registries                                                               (layer.py:1155-1195)
# ------------------------------------------------------------------------------------------
layer_parsers = {"data": lambda: DataLayerParser(),
                 "fc": lambda: FCLayerParser(),
                 "conv": lambda: ConvLayerParser(),
                 "local": lambda: LocalUnsharedLayerParser(),
                 "softmax": lambda: SoftmaxLayerParser(),
                 "eltsum": lambda: EltwiseSumLayerParser(),
                 "eltmax": lambda: EltwiseMaxLayerParser(),
                 "neuron": lambda: NeuronLayerParser(),
                 "pool": lambda: PoolLayerParser(),
                 "rnorm": lambda: NormLayerParser(NormLayerParser.RESPONSE_NORM),
                 "cnorm": lambda: NormLayerParser(NormLayerParser.CONTRAST_NORM),
                 "cmrnorm": lambda: NormLayerParser(NormLayerParser.CROSSMAP_RESPONSE_NORM),
                 "nailbed": lambda: NailbedLayerParser(),
                 "blur": lambda: GaussianBlurLayerParser(),
                 "resize": lambda: ResizeLayerParser(),
                 "rgb2yuv": lambda: RGBToYUVLayerParser(),
                 "rgb2lab": lambda: RGBToLABLayerParser(),
                 "rscale": lambda: RandomScaleLayerParser(),
                 "cost.logreg": lambda: LogregCostParser(),
                 "cost.sum2": lambda: SumOfSquaresCostParser()}

neuron_parsers = sorted([NeuronParser("ident", "f(x) = x", uses_acts=False, uses_inputs=False),
                         NeuronParser("logistic", "f(x) = 1 / (1 + e^-x)", uses_acts=True, uses_inputs=False),
                         NeuronParser("abs", "f(x) = |x|", uses_acts=False, uses_inputs=True),
                         NeuronParser("relu", "f(x) = max(0, x)", uses_acts=True, uses_inputs=False),
                         NeuronParser("softrelu", "f(x) = log(1 + e^x)", uses_acts=True, uses_inputs=False),
                         NeuronParser("square", "f(x) = x^2", uses_acts=False, uses_inputs=True),
                         NeuronParser("sqrt", "f(x) = sqrt(x)", uses_acts=True, uses_inputs=False),
                         ParamNeuronParser("tanh[a,b]", "f(x) = a * tanh(b * x)", uses_acts=True, uses_inputs=False),
                         ParamNeuronParser("brelu[a]", "f(x) = min(a, max(0, x))", uses_acts=True, uses_inputs=False),
                         ParamNeuronParser("linear[a,b]", "f(x) = a * x + b", uses_acts=True, uses_inputs=False)],
                        key=lambda x: x.type)

import plugin  # noqa: E402  (the plugin module imports names from this one)

layer_parsers.update(plugin.extra_layer_parsers)
neuron_parsers = sorted(neuron_parsers + plugin.extra_neuron_parsers, key=lambda x: x.type)
