"""Config-side half of the plugin mechanism: the schemas of the layer / neuron types whose device code lives in
Plugins/*.so (the reference's Python/plugin.py declares the same three): NIN's `cccp` cross-channel layer, the `dropout[prob]`
neuron and the `cost.gsinlabel` cost.  layer.py folds `extra_layer_parsers` / `extra_neuron_parsers` into its registries at
import; a third-party plugin adds its own types with `LayerParser.register_layer_parser`."""
from layer import CostParser, Key, LayerParsingError, NeuronParser, WeightLayerParser


class CCCPLayerParser(WeightLayerParser):
    """Cascaded cross-channel parametric pooling: at every pixel, `in_nodes` input maps -> `out_nodes` output maps through one
    shared matrix (a 1x1 convolution), independently in each of `channels` slices.  Weights (in_nodes, out_nodes*channels)
    Fortran-ordered, biases (1, out_nodes*channels)."""
    flags = dict(gradConsumer=True, usesActs=False)
    keys = WeightLayerParser.keys + (Key("channels", "int"), Key("in_nodes", "int"), Key("out_nodes", "int"))

    def build(self, mcp):
        dic = self.dic
        maps_out = dic["out_nodes"] * dic["channels"]
        dic["outputs"] = maps_out * (dic["numInputs"][0] // (dic["in_nodes"] * dic["channels"]))
        self.verify_num_range(dic["outputs"], "outputs", 1, None)
        self.make_weights(dic["initW"], [dic["in_nodes"]], [maps_out], order="F")
        self.make_biases(1, maps_out, order="F")
        print("Initialized filter-fully-connected layer '%s', producing %d outputs in %d channels" % (dic["name"], dic["out_nodes"], dic["channels"]))


class GroupSparsityInLabelCostParser(CostParser):
    """Inputs: labels, then (channels, imgSize, imgSize) maps whose per-class groups the cost keeps sparse."""
    num_inputs = 2
    keys = (Key("channels", "int"), Key("imgSize", "int"))

    def build(self, mcp):
        dic = self.dic
        if dic["numInputs"][0] != 1:
            raise LayerParsingError("Layer '%s': dimensionality of first input must be 1" % dic["name"])
        print("Initialized groupSparsityInLabel cost '%s' with %d channels and %dx%d inputs" % (dic["name"], dic["channels"], dic["imgSize"], dic["imgSize"]))


extra_layer_parsers = {"cccp": CCCPLayerParser, "cost.gsinlabel": GroupSparsityInLabelCostParser}
extra_neuron_parsers = [NeuronParser("dropout[prob]", "f(x) = rand() < prob ? 0:x", uses_acts=False, uses_inputs=False)]
