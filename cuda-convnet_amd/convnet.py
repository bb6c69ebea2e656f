"""`python convnet.py --data-path ... --layer-def ... --layer-params ...` : the ConvNet driver
(Python 3).  Same flags, same stdout format, same checkpoint layout as the reference's
Python/convnet.py:38-228; the native module it drives is this repo's `libkernel`
(csrc/runtime, HIP kernels through libmckernels.so).

Data parallelism (not in the reference): when launched as one process per GPU
(`python -m torch.distributed.run --nproc-per-node G convnet.py ...`, RANK / WORLD_SIZE /
LOCAL_RANK in the environment) every rank builds the same model from the same seed, takes its
slice of each global minibatch of `minibatch_size * G` cases and the native module all-reduces
the weight gradients over RCCL.  Only rank 0 prints and saves.
"""
import math as m
import os
import sys
from os import linesep as NL

import numpy as n
import numpy.random as nr

import layer as lay
from convdata import CIFARDataProvider, CroppedCIFARDataProvider, DummyConvNetDataProvider, SyntheticDataProvider
from data import DataProvider, DataProviderException
from gpumodel import IGPUModel, ModelStateException
from options import BooleanOptionParser, IntegerOptionParser, ListOptionParser, StringOptionParser
from util import is_windows_machine


import dp


def dp_env():
    return dp.env()


def shard_batch(mats, minibatch, rank, world):
    """Rank `rank`'s cases of a (dims, cases) batch list (see dp.shard_columns)."""
    if world == 1:
        return mats
    cols = dp.shard_columns(mats[0].shape[1], minibatch, rank, world)
    return [n.require(x[:, cols], dtype=n.single, requirements="C") for x in mats]


class ConvNet(IGPUModel):
    def __init__(self, op, load_dic, dp_params=None):
        dp_params = {} if dp_params is None else dp_params
        dp_params["multiview_test"] = op.get_value("multiview_test")
        dp_params["crop_border"] = op.get_value("crop_border")
        self.rank, self.world, self.local_rank = dp_env()
        if self.world > 1:
            import torch  # noqa: F401  (before libkernel: both must share one HIP runtime, torch's)
        IGPUModel.__init__(self, "ConvNet", op, load_dic, [], dp_params=dp_params)

    def import_model(self):
        lib_name = "pyconvnet" if is_windows_machine() else "libkernel"
        print("=========================")
        print("Importing %s C++ module" % lib_name)
        self.libmodel = __import__(lib_name)

    def init_model_lib(self):
        dev = self.device_ids[0]
        if self.world > 1:
            dev = self.local_rank if dev < 0 else dev
            self.libmodel.initModel(self.layers, self.minibatch_size, dev)
            dp.connect(self.libmodel, self.rank, self.world)
        else:
            self.libmodel.initModel(self.layers, self.minibatch_size, dev)

    def init_model_state(self):
        ms = self.model_state
        if not self.load_file and (self.world > 1 or "CONVNET_SEED" in os.environ):
            nr.seed(int(os.environ.get("CONVNET_SEED", 1)))      # identical initial weights on every rank / reproducible runs
        if self.load_file:
            ms["layers"] = lay.LayerParser.parse_layers(self.layer_def, self.layer_params, self, ms["layers"])
        else:
            ms["layers"] = lay.LayerParser.parse_layers(self.layer_def, self.layer_params, self)
        self.layers_dic = dict((l["name"], l) for l in ms["layers"])

        logreg_name = self.op.get_value("logreg_name")
        if logreg_name:
            self.logreg_idx = self.get_layer_idx(logreg_name, check_type="cost.logreg")

        for i, l in enumerate(ms["layers"]):
            if l["type"] == "conv" and l["name"] in self.op.get_value("conv_to_local"):
                lay.LocalLayerParser.conv_to_local(ms["layers"], i)
        for name_str in self.op.get_value("unshare_weights"):
            if not name_str:
                continue
            parsed = lay.WeightLayerParser.get_layer_name(name_str)
            if parsed is None:
                raise ModelStateException("Invalid layer name '%s'; unable to unshare." % name_str)
            if parsed[0] not in self.layers_dic:
                raise ModelStateException("Layer '%s' does not exist; unable to unshare" % parsed[0])
            lay.WeightLayerParser.unshare_weights(self.layers_dic[parsed[0]], ms["layers"], matrix_idx=parsed[1])
        self.op.set_value("conv_to_local", [], parse=False)
        self.op.set_value("unshare_weights", [], parse=False)

    def get_layer_idx(self, layer_name, check_type=None):
        names = [l["name"] for l in self.model_state["layers"]]
        if layer_name not in names:
            raise ModelStateException("Layer with name '%s' not defined." % layer_name)
        idx = names.index(layer_name)
        ltype = self.model_state["layers"][idx]["type"]
        if check_type and ltype != check_type:
            raise ModelStateException("Layer with name '%s' has type '%s'; should be '%s'." % (layer_name, ltype, check_type))
        return idx

    def fill_excused_options(self):
        if self.op.get_value("check_grads"):
            self.op.set_value("save_path", "")
            self.op.set_value("train_batch_range", "0")
            self.op.set_value("test_batch_range", "0")
            self.op.set_value("data_path", "")

    def parse_batch_data(self, batch_data, train=True):
        if any(d.dtype != n.single for d in batch_data[2]):
            raise DataProviderException("All matrices returned by data provider must consist of single-precision floats.")
        return batch_data

    def start_batch(self, batch_data, train=True):
        data = shard_batch(batch_data[2], self.minibatch_size, self.rank, self.world)
        self._held_batch = data          # the native module keeps views of these arrays until finishBatch
        if self.check_grads:
            self.libmodel.checkGradients(data)    # prints the report; the reference's worker exits the process here
            sys.stdout.flush()
            sys.exit(0)
        elif not train and self.multiview_test:
            self.libmodel.startMultiviewTest(data, self.train_data_provider.num_views, self.logreg_idx)
        else:
            self.libmodel.startBatch(data, not train)

    # -- printing (rank 0 only under data parallelism) ----------------------------------------
    def print_iteration(self):
        print("%d.%d..." % (self.epoch, self.batchnum), end=" ")

    def print_train_time(self, compute_time_py):
        print("(%.3f sec)" % compute_time_py)

    def print_costs(self, cost_outputs):
        costs, num_cases = cost_outputs[0], cost_outputs[1]
        for errname in costs.keys():
            costs[errname] = [(v / num_cases) for v in costs[errname]]
            print("%s: " % errname, end=" ")
            print(", ".join("%6f" % v for v in costs[errname]), end=" ")
            if any(m.isnan(v) or m.isinf(v) for v in costs[errname]):
                print("^ got nan or inf!")
                sys.exit(1)

    def print_mean_cost(self, cost_outputs):
        keys = list(cost_outputs[0][0].keys())
        for key in keys:
            mean = n.array([c[0][key] for c in cost_outputs]).mean(axis=0)
            print("%s: " % key, end=" ")
            print(", ".join("%6f" % v for v in mean), end=" ")

    def print_train_results(self):
        self.print_costs(self.train_outputs[-1])

    def print_test_status(self):
        pass

    def print_test_results(self):
        print("")
        print("======================Test output======================")
        self.print_costs(self.test_outputs[-1])
        print("")
        print("-------------------------------------------------------", end=" ")
        for l in self.layers:
            if "weights" in l:
                print("")
                print(NL.join("Layer '%s' weights[%d]: %e [mean %e] [max %e] [min %e]" % (l["name"], i, n.mean(n.abs(w)), n.mean(n.abs(wi)), n.max(n.abs(wi)), n.min(n.abs(wi)))
                              for i, (w, wi) in enumerate(zip(l["weights"], l["weightsInc"]))), end=" ")
                print("%sLayer '%s' biases: %e [mean %e] [max %e] [min %e]" % (NL, l["name"], n.mean(n.abs(l["biases"])), n.mean(n.abs(l["biasesInc"])), n.max(n.abs(l["biasesInc"])), n.min(n.abs(l["biasesInc"]))), end=" ")
        print("")

    def conditional_save(self):
        if self.rank == 0:
            self.save_state()
        print("-------------------------------------------------------")
        print("Saved checkpoint to %s" % os.path.join(self.save_path, self.save_file))
        print("=======================================================", end=" ")

    def aggregate_test_outputs(self, test_outputs):
        num_cases = sum(t[1] for t in test_outputs)
        for i in range(1, len(test_outputs)):
            for k, v in test_outputs[i][0].items():
                for j in range(len(v)):
                    test_outputs[0][0][k][j] += test_outputs[i][0][k][j]
        return (test_outputs[0][0], num_cases)

    @classmethod
    def get_options_parser(cls):
        op = IGPUModel.get_options_parser()
        op.add_option("mini", "minibatch_size", IntegerOptionParser, "Minibatch size", default=128)
        op.add_option("layer-def", "layer_def", StringOptionParser, "Layer definition file", set_once=True)
        op.add_option("layer-params", "layer_params", StringOptionParser, "Layer parameter file")
        op.add_option("check-grads", "check_grads", BooleanOptionParser, "Check gradients and quit?", default=0, excuses=["data_path", "save_path", "train_batch_range", "test_batch_range"])
        op.add_option("multiview-test", "multiview_test", BooleanOptionParser, "Cropped DP: test on multiple patches?", default=0, requires=["logreg_name"])
        op.add_option("crop-border", "crop_border", IntegerOptionParser, "Cropped DP: crop border size", default=4, set_once=True)
        op.add_option("logreg-name", "logreg_name", StringOptionParser, "Cropped DP: logreg layer name (for --multiview-test)", default="")
        op.add_option("conv-to-local", "conv_to_local", ListOptionParser(StringOptionParser), "Convert given conv layers to unshared local", default=[])
        op.add_option("unshare-weights", "unshare_weights", ListOptionParser(StringOptionParser), "Unshare weight matrices in given layers", default=[])
        op.add_option("conserve-mem", "conserve_mem", BooleanOptionParser, "Conserve GPU memory (slower)?", default=0)

        op.delete_option("max_test_err")
        op.options["max_filesize_mb"].default = 0
        op.options["testing_freq"].default = 50
        op.options["num_epochs"].default = 50000
        op.options["dp_type"].default = None

        DataProvider.register_data_provider("cifar", "CIFAR", CIFARDataProvider)
        DataProvider.register_data_provider("dummy-cn-n", "Dummy ConvNet", DummyConvNetDataProvider)
        DataProvider.register_data_provider("cifar-cropped", "Cropped CIFAR", CroppedCIFARDataProvider)
        DataProvider.register_data_provider("dummy-synth-n", "Seeded synthetic batches (benchmarks)", SyntheticDataProvider)
        return op


def main(argv=None):
    op = ConvNet.get_options_parser()
    op, load_dic = IGPUModel.parse_options(op, argv=argv)
    if dp_env()[0] != 0:
        sys.stdout = open(os.devnull, "w")     # one voice under data parallelism
    model = ConvNet(op, load_dic)
    model.start()


if __name__ == "__main__":
    main()
