"""Model-independent half of the training driver (Python 3): options, data providers, the train / test / checkpoint cycle
around a native module that exposes `startBatch` / `finishBatch` / `syncWithHost`.

Contract kept from the reference's Python/gpumodel.py (its flags, stdout lines and checkpoint layout are the boundary users
script against): option names and defaults; a checkpoint is `<save_path>/<save_file>/<epoch>.<batch>`, a pickle of
{"model_state": ..., "op": ...}; `-f` resumes from the newest file of such a directory with the saved options as defaults;
a test pass and a checkpoint every `testing_freq` batches; while the device works on batch i the host loads batch i+1.

Subclasses (convnet.ConvNet) supply the hooks: import_model, init_model_state, init_model_lib, fill_excused_options,
parse_batch_data, start_batch, the print_* family, conditional_save, aggregate_test_outputs."""
import os
import sys
import time

from data import DataProvider, DataProviderException, dp_types
from options import (BooleanOptionParser, FloatOptionParser, IntegerOptionParser, ListOptionParser, OptionException,
                     OptionExpression, OptionMissingException, OptionsParser, RangeOptionParser, StringOptionParser)
from util import GPU_LOCK_NO_LOCK, UnpickleError, alphanum_key, get_gpu_lock, pickle, unpickle

RULE = "========================="

# (flag, attribute, parser, description, keyword arguments of add_option)
BASE_OPTIONS = [
    ("f", "load_file", StringOptionParser, "Load file", dict(default="", excuses=OptionsParser.EXCLUDE_ALL)),
    ("train-range", "train_batch_range", RangeOptionParser, "Data batch range: training", {}),
    ("test-range", "test_batch_range", RangeOptionParser, "Data batch range: testing", {}),
    ("data-provider", "dp_type", StringOptionParser, "Data provider", dict(default="default")),
    ("test-freq", "testing_freq", IntegerOptionParser, "Testing frequency", dict(default=25)),
    ("epochs", "num_epochs", IntegerOptionParser, "Number of epochs", dict(default=500)),
    ("data-path", "data_path", StringOptionParser, "Data path", {}),
    ("save-path", "save_path", StringOptionParser, "Save path", {}),
    ("max-filesize", "max_filesize_mb", IntegerOptionParser, "Maximum save file size (MB)", dict(default=5000)),
    ("max-test-err", "max_test_err", FloatOptionParser, "Maximum test error for saving", {}),
    ("num-gpus", "num_gpus", IntegerOptionParser, "Number of GPUs", dict(default=1)),
    ("test-only", "test_only", BooleanOptionParser, "Test and quit?", dict(default=0)),
    ("zip-save", "zip_save", BooleanOptionParser, "Compress checkpoints?", dict(default=0)),
    ("test-one", "test_one", BooleanOptionParser, "Test on one batch at a time?", dict(default=1)),
    ("gpu", "gpu", ListOptionParser(IntegerOptionParser), "GPU override", dict(default=OptionExpression("[-1] * num_gpus"))),
]


class ModelStateException(Exception):
    """Raised by a subclass's init_model_state when a checkpoint and the command line cannot be reconciled."""


def _quit(*lines, code=None):
    """Prints what went wrong and ends the process the way the reference driver does (its callers script against the exit)."""
    for line in lines:
        print(line)
    sys.exit(code)


class IGPUModel(object):
    def __init__(self, model_name, op, load_dic, filename_options=None, dp_params=None):
        self.model_name, self.op, self.options = model_name, op, op.options
        self.load_dic, self.filename_options = load_dic, filename_options
        self.dp_params = {} if dp_params is None else dp_params
        resumed = bool(load_dic)
        self.range_train_outputs = list()
        self.get_gpus(); self.fill_excused_options()
        vars(self).update((opt.name, opt.value) for opt in op.get_options_list())   # every option is an attribute of the model
        if resumed:
            self._resume_from(load_dic)
        else:
            self._start_fresh(filename_options)
        self.init_data_providers()
        if resumed:                                  # the checkpoint names the last batch that was trained on
            self.train_data_provider.advance_batch()
        try:
            self.init_model_state()
        except ModelStateException as problem:
            _quit(problem, code=1)
        vars(self).update(self.model_state)          # epoch, batchnum, train_outputs, ... live on the model while it runs
        self.import_model(); self.init_model_lib()

    def _resume_from(self, checkpoint):
        self.model_state = checkpoint["model_state"]
        given = self.options["load_file"].value
        self.save_file = given if os.path.isdir(given) else os.path.dirname(given)

    def _start_fresh(self, filename_options):
        self.model_state = dict(train_outputs=[], test_outputs=[], epoch=1, batchnum=self.train_batch_range[0])
        if filename_options is None:
            return
        tags = ["%s_%s" % (char, self.options[name].get_str_value()) for name, char in filename_options]
        # "<model>_<c_value>_..._<time>"; with no tags the two separators stay ("ConvNet__<time>"), as tools expect
        self.save_file = "%s_%s_%s" % (self.model_name, "_".join(tags), time.strftime("%Y-%m-%d_%H.%M.%S"))

    # ---------------------------------------------------------------- hooks
    def import_model(self):
        print(RULE)
        print("Importing _%s C++ module" % self.model_name)
        self.libmodel = __import__("_%s" % self.model_name)

    def fill_excused_options(self):
        """Give values to options that an EXCLUDE/REQUIRE relation excused from the command line."""

    def init_model_state(self):
        """Build (fresh run) or reconcile (resumed run) `model_state`; may raise ModelStateException."""

    def init_model_lib(self):
        """Hand the model to the native module."""

    def print_model_state(self):
        """Describe the model at the top of a training run."""

    def print_mean_cost(self, costs):
        """Print the mean of a list of finishBatch results."""

    def init_data_providers(self):
        self.dp_params.update(convnet=self)
        common = dict(type=self.dp_type, dp_params=self.dp_params)
        at = self.model_state
        try:
            self.test_data_provider = DataProvider.get_instance(self.data_path, self.test_batch_range, test=True, **common)
            self.train_data_provider = DataProvider.get_instance(self.data_path, self.train_batch_range, at["epoch"], at["batchnum"],
                                                                 test=False, **common)
        except DataProviderException as problem:
            print("Unable to create data provider: %s" % problem)
            self.print_data_providers()
            _quit()

    # ---------------------------------------------------------------- the cycle
    def start(self):
        if not self.test_only:
            return self.train()
        self.test_outputs.append(self.get_test_error())
        self.print_test_results(); _quit(code=0)

    def train(self):
        self._print_banner()
        del self.range_train_outputs[:]
        upcoming = self.get_next_batch()
        while self.num_epochs >= self.epoch:
            batch, began = upcoming, time.time()
            self.epoch, self.batchnum = batch[:2]
            self.print_iteration(); sys.stdout.flush()
            self.start_batch(batch)
            upcoming = self.get_next_batch()            # host I/O while the device trains
            result = self.finish_batch()
            for history in (self.train_outputs, self.range_train_outputs):
                history.append(result)
            self.print_train_results()
            if not self.get_num_batches_done() % self.testing_freq:
                self.test_and_checkpoint()
            self.print_train_time(time.time() - began)
        self.cleanup()

    def _print_banner(self):
        print("%s\nTraining %s" % (RULE, self.model_name))
        self.op.print_values()
        print(RULE)
        self.print_model_state()
        print("Running on device(s) %s" % ", ".join(map(str, self.device_ids)))
        print("Current time: %s" % time.asctime(time.localtime()))
        print("Saving checkpoints to %s" % os.path.join(self.save_path, self.save_file))
        print(RULE)

    def test_and_checkpoint(self):
        """What happens every `testing_freq` batches: a test pass, the report, a checkpoint if the test error allows one."""
        self.sync_with_host()
        self.test_outputs.append(self.get_test_error())
        self.print_mean_train_results()
        self.range_train_outputs = list()
        for report in (self.print_test_results, self.print_test_status, self.conditional_save):
            report()

    def cleanup(self):
        _quit(code=0)

    def sync_with_host(self):
        self.libmodel.syncWithHost()

    def get_num_batches_done(self):
        per_epoch, first = len(self.train_batch_range), self.train_batch_range[0]
        return per_epoch * (self.epoch - 1) + (self.batchnum - first) + 1

    def get_next_batch(self, train=True):
        provider = self.train_data_provider if train else self.test_data_provider
        return self.parse_batch_data(provider.get_next_batch(), train=train)

    def parse_batch_data(self, batch_data, train=True):
        epoch, batchnum, payload = batch_data
        return epoch, batchnum, payload["data"]

    def start_batch(self, batch_data, train=True):
        self.libmodel.startBatch(batch_data[2], not train)

    def finish_batch(self):
        return self.libmodel.finishBatch()

    def get_test_error(self):
        """One test batch (`test_one`) or the whole test range, loading the next batch while the device evaluates this one."""
        results, last = [], self.test_batch_range[-1]
        batch = self.get_next_batch(train=False)
        while batch is not None:
            self.start_batch(batch, train=False)
            more = not self.test_one and batch[1] < last
            upcoming = self.get_next_batch(train=False) if more else None
            results.append(self.finish_batch())
            if self.test_only:
                print("batch %d: %s" % (batch[1], results[-1]), flush=True)
            batch = upcoming
        return self.aggregate_test_outputs(results)

    def aggregate_test_outputs(self, test_outputs):
        batches = 1 if self.test_one else len(self.test_batch_range)
        return tuple(sum(column) / batches for column in zip(*test_outputs))

    # ---------------------------------------------------------------- reporting
    def print_iteration(self):
        print("\t%d.%d..." % (self.epoch, self.batchnum), end=" ")

    def print_train_time(self, compute_time_py):
        print("(%.3f sec)" % compute_time_py)

    def print_train_results(self):
        cost = self.train_outputs[-1][0]
        if not 0 < cost < 2e20:
            print("Crazy train error: %.6f" % cost)
            self.cleanup()
        print("Train error: %.6f " % cost, end=" ")

    def print_mean_train_results(self):
        print("\n======================Train output======================\nmean train cost: ", end=" ")
        self.print_mean_cost(self.range_train_outputs)

    def print_test_results(self):
        print("%s\t\tTest error: %.6f" % (os.linesep, self.test_outputs[-1][0]), end=" ")

    def print_test_status(self):
        errs = [t[0] for t in self.test_outputs[-2:]]
        print("ok" if len(errs) == 1 or errs[1] < errs[0] else "WORSE", end=" ")

    def conditional_save(self):
        if not 0 < self.test_outputs[-1][0] < self.max_test_err:
            return print("\tTest error > %g, not saving." % self.max_test_err, end=" ")
        self.save_state()

    # ---------------------------------------------------------------- state
    def set_var(self, var_name, var_val):
        """Model variables live twice while the model runs: in `model_state` (what a checkpoint holds) and as attributes."""
        setattr(self, var_name, var_val)
        self.model_state.update({var_name: var_val})
        return var_val

    def get_var(self, var_name):
        return self.model_state.__getitem__(var_name)

    def has_var(self, var_name):
        return self.model_state.__contains__(var_name)

    def save_state(self):
        for key in self.model_state:                # pull the running values back into the state dict
            if hasattr(self, key):
                self.model_state[key] = getattr(self, key)
        where = os.path.join(self.save_path, self.save_file)
        os.makedirs(where, exist_ok=True)
        pickle(os.path.join(where, "%d.%d" % (self.epoch, self.batchnum)), {"model_state": self.model_state, "op": self.op}, compress=self.zip_save)

    @staticmethod
    def load_checkpoint(load_dir):
        """A checkpoint file, or the newest file ("<epoch>.<batch>", natural order) of a checkpoint directory."""
        if os.path.isdir(load_dir):
            load_dir = os.path.join(load_dir, max(os.listdir(load_dir), key=alphanum_key))
        return unpickle(load_dir)

    # ---------------------------------------------------------------- options
    @staticmethod
    def get_options_parser():
        parser = OptionsParser()
        for flag, attr, kind, desc, kw in BASE_OPTIONS:
            parser.add_option(flag, attr, kind, desc, **kw)
        return parser

    @staticmethod
    def print_data_providers():
        print("\n".join(["Available data providers:"] + ["    %s: %s" % item for item in dp_types.items()]))

    def get_gpus(self):
        self.device_ids = list(map(get_gpu_lock, self.op.get_value("gpu")))
        if any(d == GPU_LOCK_NO_LOCK for d in self.device_ids):
            _quit("Not enough free GPUs!")

    @staticmethod
    def parse_options(op, argv=None):
        """(options, checkpoint or None); with -f the checkpoint's options are the defaults and the command line overrides."""
        checkpoint = None
        try:
            given = op.parse(argv=argv)
            if given["load_file"].value_given:
                checkpoint = IGPUModel.load_checkpoint(given["load_file"].value)
                checkpoint["op"].merge_from(op)
                op = checkpoint["op"]
            op.eval_expr_defaults()
        except OptionMissingException as problem:
            print(problem)
            op.print_usage()
            _quit()
        except OptionException as problem:
            _quit(problem)
        except UnpickleError as problem:
            _quit("Error loading checkpoint:", problem)
        return op, checkpoint
