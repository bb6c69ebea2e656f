"""Python half of the plugin mechanism: parsers for the layer / neuron types that live in
Plugins/*.so (Python/plugin.py:18-62 of the reference): the NIN `cccp` cross-channel layer, the
`dropout[prob]` neuron and the `cost.gsinlabel` cost.  `layer.py` merges these two registries
into its own at import time."""
from layer import CostParser, LayerParsingError, ParamNeuronParser, WeightLayerParser


class GroupSparsityInLabelCostParser(CostParser):
    def __init__(self):
        CostParser.__init__(self, num_inputs=2)

    def parse(self, name, mcp, prev_layers, model=None):
        dic = CostParser.parse(self, name, mcp, prev_layers, model)
        if dic["numInputs"][0] != 1:
            raise LayerParsingError("Layer '%s': dimensionality of first input must be 1" % name)
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["imgSize"] = mcp.safe_get_int(name, "imgSize")
        print("Initialized groupSparsityInLabel cost '%s' with %d channels and %dx%d inputs" % (name, dic["channels"], dic["imgSize"], dic["imgSize"]))
        return dic


class CCCPLayerParser(WeightLayerParser):
    """Cascaded cross-channel parametric pooling = a 1x1 convolution over `in_nodes` input maps
    producing `out_nodes` maps (per `channels` slice).  Weights (in_nodes, out_nodes*channels)
    Fortran-ordered, bias (1, out_nodes*channels)."""

    def parse(self, name, mcp, prev_layers, model=None):
        dic = WeightLayerParser.parse(self, name, mcp, prev_layers, model)
        dic["usesActs"] = False
        dic["channels"] = mcp.safe_get_int(name, "channels")
        dic["in_nodes"] = mcp.safe_get_int(name, "in_nodes")
        dic["out_nodes"] = mcp.safe_get_int(name, "out_nodes")
        pixels = dic["numInputs"][0] // (dic["in_nodes"] * dic["channels"])
        dic["outputs"] = dic["out_nodes"] * dic["channels"] * pixels
        self.verify_num_range(dic["outputs"], "outputs", 1, None)
        self.make_weights(dic["initW"], [dic["in_nodes"]], [dic["out_nodes"] * dic["channels"]], order="F")
        self.make_biases(1, dic["out_nodes"] * dic["channels"], order="F")
        print("Initialized filter-fully-connected layer '%s', producing %d outputs in %d channels" % (name, dic["out_nodes"], dic["channels"]))
        return dic


extra_layer_parsers = {"cost.gsinlabel": lambda: GroupSparsityInLabelCostParser(),
                       "cccp": lambda: CCCPLayerParser()}

extra_neuron_parsers = sorted([ParamNeuronParser("dropout[prob]", "f(x) = rand() < prob ? 0:x", uses_acts=False, uses_inputs=False)],
                              key=lambda x: x.type)
