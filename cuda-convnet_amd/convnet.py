"""`python convnet.py --data-path ... --layer-def ... --layer-params ...`: the ConvNet driver (Python 3).  Flags, stdout
format and checkpoint layout are the reference's (Python/convnet.py); the native module underneath is this repo's `libkernel`
(csrc/runtime over the HIP kernels of libmckernels.so).

Data parallelism (not in the reference): started as one process per GPU (`python -m torch.distributed.run --nproc-per-node G
convnet.py ...`, RANK / WORLD_SIZE / LOCAL_RANK in the environment) every rank builds the same model from the same seed, takes
its share of each global minibatch of `minibatch_size * G` cases, and the native module all-reduces the weight gradients over
RCCL.  Only rank 0 prints and saves."""
import math
import os
import sys
import time

import numpy as n
import numpy.random as nr

import convdata
import dp
import layer as lay
from data import DataProvider, DataProviderException
from gpumodel import IGPUModel, ModelStateException
from options import BooleanOptionParser, IntegerOptionParser, ListOptionParser, StringOptionParser
from util import is_windows_machine

BAR = "-------------------------------------------------------"

CONVNET_OPTIONS = [
    ("mini", "minibatch_size", IntegerOptionParser, "Minibatch size", dict(default=128)),
    ("layer-def", "layer_def", StringOptionParser, "Layer definition file", dict(set_once=True)),
    ("layer-params", "layer_params", StringOptionParser, "Layer parameter file", {}),
    ("check-grads", "check_grads", BooleanOptionParser, "Check gradients and quit?",
     dict(default=0, excuses=["data_path", "save_path", "train_batch_range", "test_batch_range"])),
    ("multiview-test", "multiview_test", BooleanOptionParser, "Cropped DP: test on multiple patches?", dict(default=0, requires=["logreg_name"])),
    ("crop-border", "crop_border", IntegerOptionParser, "Cropped DP: crop border size", dict(default=4, set_once=True)),
    ("logreg-name", "logreg_name", StringOptionParser, "Cropped DP: logreg layer name (for --multiview-test)", dict(default="")),
    ("conv-to-local", "conv_to_local", ListOptionParser(StringOptionParser), "Convert given conv layers to unshared local", dict(default=[])),
    ("unshare-weights", "unshare_weights", ListOptionParser(StringOptionParser), "Unshare weight matrices in given layers", dict(default=[])),
    ("conserve-mem", "conserve_mem", BooleanOptionParser, "Conserve GPU memory (slower)?", dict(default=0)),
]
CHANGED_DEFAULTS = {"max_filesize_mb": 0, "testing_freq": 50, "num_epochs": 50000, "dp_type": None}
CONVNET_PROVIDERS = [
    ("cifar", "CIFAR", convdata.CIFARDataProvider),
    ("mnist", "MNIST", convdata.MnistDataProvider),
    ("cifar-cropped", "Cropped CIFAR", convdata.CroppedCIFARDataProvider),
    ("dummy-cn-n", "Dummy ConvNet", convdata.DummyConvNetDataProvider),
    ("dummy-synth-n", "Seeded synthetic batches (benchmarks)", convdata.SyntheticDataProvider),
]


def dp_env():
    return dp.env()


def shard_batch(mats, minibatch, rank, world):
    """This rank's columns of every (dims, cases) matrix of a batch (dp.shard_columns says which)."""
    if world == 1:
        return mats
    mine = dp.shard_columns(mats[0].shape[1], minibatch, rank, world)
    return [n.require(x[:, mine], dtype=n.single, requirements="C") for x in mats]


def base_seed():
    """CONVNET_SEED when given (reproducible runs), else something two launches will not share."""
    if "CONVNET_SEED" in os.environ:
        return int(os.environ["CONVNET_SEED"])
    return (int(time.time()) ^ (os.getpid() << 16)) & 0x7FFFFFFF


class ConvNet(IGPUModel):
    def __init__(self, op, load_dic, dp_params=None):
        dp_params = dict(dp_params or {}, multiview_test=op.get_value("multiview_test"), crop_border=op.get_value("crop_border"))
        self.rank, self.world, self.local_rank = dp_env()
        if self.world > 1:
            import torch  # noqa: F401  (ahead of libkernel: the two must share torch's HIP runtime)
        IGPUModel.__init__(self, "ConvNet", op, load_dic, [], dp_params=dp_params)

    # ---------------------------------------------------------------- set-up
    def import_model(self):
        module = "pyconvnet" if is_windows_machine() else "libkernel"
        print("=========================")
        print("Importing %s C++ module" % module)
        self.libmodel = __import__(module)

    def init_model_lib(self):
        device = self.device_ids[0]
        if self.world > 1 and device < 0:
            device = self.local_rank
        self.libmodel.initModel(self.layers, self.minibatch_size, device)
        # dropout masks: every rank its own Philox stream (same-second launches would otherwise share time(0))
        self.libmodel.setSeed(base_seed() + self.rank)
        if self.world > 1:
            dp.connect(self.libmodel, self.rank, self.world)

    def init_model_state(self):
        state = self.model_state
        if not self.load_file and (self.world > 1 or "CONVNET_SEED" in os.environ):
            nr.seed(int(os.environ.get("CONVNET_SEED", 1)))     # every rank draws the same initial weights
        state["layers"] = lay.LayerParser.parse_layers(self.layer_def, self.layer_params, self, state["layers"] if self.load_file else None)
        layers = state["layers"]
        self.layers_dic = dict((l["name"], l) for l in layers)
        if self.op.get_value("logreg_name"):
            self.logreg_idx = self.get_layer_idx(self.op.get_value("logreg_name"), check_type="cost.logreg")

        # one-shot surgery options: applied now, then cleared so that a checkpoint does not apply them again
        wanted = self.op.get_value("conv_to_local")
        for idx in [i for i, l in enumerate(layers) if l["type"] == "conv" and l["name"] in wanted]:
            lay.LocalLayerParser.conv_to_local(layers, idx)
        for ref in filter(None, self.op.get_value("unshare_weights")):
            parsed = lay.WeightLayerParser.get_layer_name(ref)
            if parsed is None:
                raise ModelStateException("Invalid layer name '%s'; unable to unshare." % ref)
            if parsed[0] not in self.layers_dic:
                raise ModelStateException("Layer '%s' does not exist; unable to unshare" % parsed[0])
            lay.WeightLayerParser.unshare_weights(self.layers_dic[parsed[0]], layers, matrix_idx=parsed[1])
        for one_shot in ("conv_to_local", "unshare_weights"):
            self.op.set_value(one_shot, [], parse=False)

    def get_layer_idx(self, layer_name, check_type=None):
        for idx, l in enumerate(self.model_state["layers"]):
            if l["name"] == layer_name:
                if check_type and l["type"] != check_type:
                    raise ModelStateException("Layer with name '%s' has type '%s'; should be '%s'." % (layer_name, l["type"], check_type))
                return idx
        raise ModelStateException("Layer with name '%s' not defined." % layer_name)

    def fill_excused_options(self):
        if self.op.get_value("check_grads"):        # gradient checking runs on dummy data and saves nothing
            for name, value in (("save_path", ""), ("train_batch_range", "0"), ("test_batch_range", "0"), ("data_path", "")):
                self.op.set_value(name, value)

    # ---------------------------------------------------------------- batches
    def parse_batch_data(self, batch_data, train=True):
        if not all(x.dtype == n.single for x in batch_data[2]):
            raise DataProviderException("All matrices returned by data provider must consist of single-precision floats.")
        return batch_data

    def start_batch(self, batch_data, train=True):
        mats = shard_batch(batch_data[2], self.minibatch_size, self.rank, self.world)
        self._held_batch = mats          # the native module reads these arrays in place until finishBatch
        if self.check_grads:
            self.libmodel.checkGradients(mats)       # prints its report (the reference's worker ends the process here, too)
            sys.stdout.flush()
            sys.exit(0)
        if self.multiview_test and not train:
            self.libmodel.startMultiviewTest(mats, self.train_data_provider.num_views, self.logreg_idx)
            return
        # under data parallelism gradients are scaled by the GLOBAL size of each minibatch (ragged tails included)
        whole = dp.global_cases(batch_data[2][0].shape[1], self.minibatch_size, self.world) if self.world > 1 else 0
        self.libmodel.startBatch(mats, not train, whole)

    def finish_batch(self):
        # data parallel: every rank trained / tested on its shard; report (and checkpoint on) the sums over all of them
        return dp.sum_costs(self.libmodel.finishBatch(), self.world)

    def aggregate_test_outputs(self, test_outputs):
        """[(costs {name: [values]}, cases)] per batch -> one (summed costs, total cases)."""
        total = dict((name, list(values)) for name, values in test_outputs[0][0].items())
        for costs, _ in test_outputs[1:]:
            for name, values in costs.items():
                total[name] = [a + b for a, b in zip(total[name], values)]
        return total, sum(cases for _, cases in test_outputs)

    # ---------------------------------------------------------------- reporting (rank 0's stdout only, see main)
    def print_iteration(self):
        print("%d.%d..." % (self.epoch, self.batchnum), end=" ")

    def print_train_time(self, compute_time_py):
        print("(%.3f sec)" % compute_time_py)

    def print_costs(self, cost_outputs):
        costs, cases = cost_outputs[0], cost_outputs[1]
        for name in costs:
            costs[name] = [v / cases for v in costs[name]]       # in place: the stored outputs are per-case from here on
            print("%s: " % name, end=" ")
            print(", ".join("%6f" % v for v in costs[name]), end=" ")
            if not all(math.isfinite(v) for v in costs[name]):
                print("^ got nan or inf!")
                sys.exit(1)

    def print_mean_cost(self, cost_outputs):
        for name in list(cost_outputs[0][0]):
            mean = n.mean([out[0][name] for out in cost_outputs], axis=0)
            print("%s: " % name, end=" ")
            print(", ".join("%6f" % v for v in mean), end=" ")

    def print_train_results(self):
        self.print_costs(self.train_outputs[-1])

    def print_test_status(self):
        pass

    def print_test_results(self):
        def mags(x):
            a = n.abs(x)
            return n.mean(a), n.max(a), n.min(a)
        print("")
        print("======================Test output======================")
        self.print_costs(self.test_outputs[-1])
        print("")
        print(BAR, end=" ")
        for l in (l for l in self.layers if "weights" in l):
            print("")
            rows = ["Layer '%s' weights[%d]: %e [mean %e] [max %e] [min %e]" % ((l["name"], i, n.mean(n.abs(w))) + mags(inc))
                    for i, (w, inc) in enumerate(zip(l["weights"], l["weightsInc"]))]
            print(os.linesep.join(rows), end=" ")
            print("%sLayer '%s' biases: %e [mean %e] [max %e] [min %e]" % ((os.linesep, l["name"], n.mean(n.abs(l["biases"]))) + mags(l["biasesInc"])), end=" ")
        print("")

    def conditional_save(self):
        if self.rank == 0:
            self.save_state()
        print(BAR)
        print("Saved checkpoint to %s" % os.path.join(self.save_path, self.save_file))
        print("=======================================================", end=" ")

    # ---------------------------------------------------------------- options
    @classmethod
    def get_options_parser(cls):
        op = IGPUModel.get_options_parser()
        for flag, attr, parser, desc, kw in CONVNET_OPTIONS:
            op.add_option(flag, attr, parser, desc, **kw)
        op.delete_option("max_test_err")
        for name, default in CHANGED_DEFAULTS.items():
            op.options[name].default = default
        for name, desc, klass in CONVNET_PROVIDERS:
            DataProvider.register_data_provider(name, desc, klass)
        return op


def main(argv=None):
    op, load_dic = IGPUModel.parse_options(ConvNet.get_options_parser(), argv=argv)
    if dp_env()[0] != 0:
        sys.stdout = open(os.devnull, "w")      # one voice under data parallelism
    ConvNet(op, load_dic).start()


if __name__ == "__main__":
    main()
