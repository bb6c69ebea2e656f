"""Training driver base class (Python 3): option handling, data-provider set-up, the
train / test / checkpoint loop.  Behaviour follows the reference's Python/gpumodel.py:44-348:
one batch is handed to the native module with `startBatch` while the next one is being loaded,
`finishBatch` returns the cost, every `testing_freq` batches the weights are synced to the host,
a test batch is evaluated and a checkpoint `<save_path>/<save_file>/<epoch>.<batch>` is written
as a pickle of {"model_state", "op"}."""
import os
import sys
from os import linesep as NL
from time import asctime, localtime, strftime, time

from data import DataProvider, DataProviderException, dp_types
from options import (BooleanOptionParser, FloatOptionParser, IntegerOptionParser, ListOptionParser, OptionException,
                     OptionExpression, OptionMissingException, OptionsParser, RangeOptionParser, StringOptionParser)
from util import GPU_LOCK_NO_LOCK, UnpickleError, alphanum_key, get_gpu_lock, pickle, unpickle


class ModelStateException(Exception):
    pass


class IGPUModel(object):
    def __init__(self, model_name, op, load_dic, filename_options=None, dp_params=None):
        self.model_name = model_name
        self.op = op
        self.options = op.options
        self.load_dic = load_dic
        self.filename_options = filename_options
        self.dp_params = {} if dp_params is None else dp_params
        self.get_gpus()
        self.fill_excused_options()
        self.range_train_outputs = []
        for o in op.get_options_list():
            setattr(self, o.name, o.value)

        if load_dic:
            self.model_state = load_dic["model_state"]
            self.save_file = self.options["load_file"].value
            if not os.path.isdir(self.save_file):
                self.save_file = os.path.dirname(self.save_file)
        else:
            self.model_state = {}
            if filename_options is not None:
                tags = ["%s_%s" % (char, self.options[opt].get_str_value()) for opt, char in filename_options]
                self.save_file = model_name + "_" + "_".join(tags) + "_" + strftime("%Y-%m-%d_%H.%M.%S")
            self.model_state["train_outputs"] = []
            self.model_state["test_outputs"] = []
            self.model_state["epoch"] = 1
            self.model_state["batchnum"] = self.train_batch_range[0]

        self.init_data_providers()
        if load_dic:
            self.train_data_provider.advance_batch()
        try:
            self.init_model_state()
        except ModelStateException as e:
            print(e)
            sys.exit(1)
        for var, val in self.model_state.items():
            setattr(self, var, val)
        self.import_model()
        self.init_model_lib()

    # -- hooks for subclasses ----------------------------------------------------------
    def import_model(self):
        print("=========================")
        print("Importing %s C++ module" % ("_" + self.model_name))
        self.libmodel = __import__("_" + self.model_name)

    def fill_excused_options(self):
        pass

    def init_model_state(self):
        pass

    def init_model_lib(self):
        pass

    def print_model_state(self):
        pass

    def print_mean_cost(self, costs):
        pass

    def init_data_providers(self):
        self.dp_params["convnet"] = self
        try:
            self.test_data_provider = DataProvider.get_instance(self.data_path, self.test_batch_range, type=self.dp_type,
                                                                dp_params=self.dp_params, test=True)
            self.train_data_provider = DataProvider.get_instance(self.data_path, self.train_batch_range,
                                                                 self.model_state["epoch"], self.model_state["batchnum"],
                                                                 type=self.dp_type, dp_params=self.dp_params, test=False)
        except DataProviderException as e:
            print("Unable to create data provider: %s" % e)
            self.print_data_providers()
            sys.exit()

    # -- main loop -----------------------------------------------------------------------
    def start(self):
        if self.test_only:
            self.test_outputs += [self.get_test_error()]
            self.print_test_results()
            sys.exit(0)
        self.train()

    def train(self):
        print("=========================")
        print("Training %s" % self.model_name)
        self.op.print_values()
        print("=========================")
        self.print_model_state()
        print("Running on device(s) %s" % ", ".join("%d" % d for d in self.device_ids))
        print("Current time: %s" % asctime(localtime()))
        print("Saving checkpoints to %s" % os.path.join(self.save_path, self.save_file))
        print("=========================")
        next_data = self.get_next_batch()
        self.range_train_outputs = []
        while self.epoch <= self.num_epochs:
            data = next_data
            self.epoch, self.batchnum = data[0], data[1]
            self.print_iteration()
            sys.stdout.flush()
            t0 = time()
            self.start_batch(data)
            next_data = self.get_next_batch()       # overlaps host loading with the device work
            batch_output = self.finish_batch()
            self.train_outputs += [batch_output]
            self.range_train_outputs += [batch_output]
            self.print_train_results()
            if self.get_num_batches_done() % self.testing_freq == 0:
                self.sync_with_host()
                self.test_outputs += [self.get_test_error()]
                self.print_mean_train_results()
                self.range_train_outputs = []
                self.print_test_results()
                self.print_test_status()
                self.conditional_save()
            self.print_train_time(time() - t0)
        self.cleanup()

    def cleanup(self):
        sys.exit(0)

    def sync_with_host(self):
        self.libmodel.syncWithHost()

    def get_num_batches_done(self):
        return len(self.train_batch_range) * (self.epoch - 1) + self.batchnum - self.train_batch_range[0] + 1

    def get_next_batch(self, train=True):
        dp = self.train_data_provider if train else self.test_data_provider
        return self.parse_batch_data(dp.get_next_batch(), train=train)

    def parse_batch_data(self, batch_data, train=True):
        return batch_data[0], batch_data[1], batch_data[2]["data"]

    def start_batch(self, batch_data, train=True):
        self.libmodel.startBatch(batch_data[2], not train)

    def finish_batch(self):
        return self.libmodel.finishBatch()

    # -- reporting -----------------------------------------------------------------------
    def print_iteration(self):
        print("\t%d.%d..." % (self.epoch, self.batchnum), end=" ")

    def print_train_time(self, compute_time_py):
        print("(%.3f sec)" % compute_time_py)

    def print_train_results(self):
        batch_error = self.train_outputs[-1][0]
        if not (batch_error > 0 and batch_error < 2e20):
            print("Crazy train error: %.6f" % batch_error)
            self.cleanup()
        print("Train error: %.6f " % batch_error, end=" ")

    def print_mean_train_results(self):
        print("\n======================Train output======================")
        print("mean train cost: ", end=" ")
        self.print_mean_cost(self.range_train_outputs)

    def print_test_results(self):
        print("%s\t\tTest error: %.6f" % (NL, self.test_outputs[-1][0]), end=" ")

    def print_test_status(self):
        better = len(self.test_outputs) == 1 or self.test_outputs[-1][0] < self.test_outputs[-2][0]
        print("ok" if better else "WORSE", end=" ")

    def conditional_save(self):
        batch_error = self.test_outputs[-1][0]
        if batch_error > 0 and batch_error < self.max_test_err:
            self.save_state()
        else:
            print("\tTest error > %g, not saving." % self.max_test_err, end=" ")

    def aggregate_test_outputs(self, test_outputs):
        div = 1 if self.test_one else len(self.test_batch_range)
        return tuple(sum(t[r] for t in test_outputs) / div for r in range(len(test_outputs[-1])))

    def get_test_error(self):
        next_data = self.get_next_batch(train=False)
        test_outputs = []
        while True:
            data = next_data
            self.start_batch(data, train=False)
            load_next = not self.test_one and data[1] < self.test_batch_range[-1]
            if load_next:
                next_data = self.get_next_batch(train=False)
            test_outputs += [self.finish_batch()]
            if self.test_only:
                print("batch %d: %s" % (data[1], str(test_outputs[-1])))
            if not load_next:
                break
            sys.stdout.flush()
        return self.aggregate_test_outputs(test_outputs)

    # -- state ---------------------------------------------------------------------------
    def set_var(self, var_name, var_val):
        setattr(self, var_name, var_val)
        self.model_state[var_name] = var_val
        return var_val

    def get_var(self, var_name):
        return self.model_state[var_name]

    def has_var(self, var_name):
        return var_name in self.model_state

    def save_state(self):
        for att in self.model_state:
            if hasattr(self, att):
                self.model_state[att] = getattr(self, att)
        dic = {"model_state": self.model_state, "op": self.op}
        checkpoint_dir = os.path.join(self.save_path, self.save_file)
        if not os.path.exists(checkpoint_dir):
            os.makedirs(checkpoint_dir)
        pickle(os.path.join(checkpoint_dir, "%d.%d" % (self.epoch, self.batchnum)), dic, compress=self.zip_save)

    @staticmethod
    def load_checkpoint(load_dir):
        if os.path.isdir(load_dir):
            return unpickle(os.path.join(load_dir, sorted(os.listdir(load_dir), key=alphanum_key)[-1]))
        return unpickle(load_dir)

    # -- options ---------------------------------------------------------------------------
    @staticmethod
    def get_options_parser():
        op = OptionsParser()
        op.add_option("f", "load_file", StringOptionParser, "Load file", default="", excuses=OptionsParser.EXCLUDE_ALL)
        op.add_option("train-range", "train_batch_range", RangeOptionParser, "Data batch range: training")
        op.add_option("test-range", "test_batch_range", RangeOptionParser, "Data batch range: testing")
        op.add_option("data-provider", "dp_type", StringOptionParser, "Data provider", default="default")
        op.add_option("test-freq", "testing_freq", IntegerOptionParser, "Testing frequency", default=25)
        op.add_option("epochs", "num_epochs", IntegerOptionParser, "Number of epochs", default=500)
        op.add_option("data-path", "data_path", StringOptionParser, "Data path")
        op.add_option("save-path", "save_path", StringOptionParser, "Save path")
        op.add_option("max-filesize", "max_filesize_mb", IntegerOptionParser, "Maximum save file size (MB)", default=5000)
        op.add_option("max-test-err", "max_test_err", FloatOptionParser, "Maximum test error for saving")
        op.add_option("num-gpus", "num_gpus", IntegerOptionParser, "Number of GPUs", default=1)
        op.add_option("test-only", "test_only", BooleanOptionParser, "Test and quit?", default=0)
        op.add_option("zip-save", "zip_save", BooleanOptionParser, "Compress checkpoints?", default=0)
        op.add_option("test-one", "test_one", BooleanOptionParser, "Test on one batch at a time?", default=1)
        op.add_option("gpu", "gpu", ListOptionParser(IntegerOptionParser), "GPU override", default=OptionExpression("[-1] * num_gpus"))
        return op

    @staticmethod
    def print_data_providers():
        print("Available data providers:")
        for dp, desc in dp_types.items():
            print("    %s: %s" % (dp, desc))

    def get_gpus(self):
        self.device_ids = [get_gpu_lock(g) for g in self.op.get_value("gpu")]
        if GPU_LOCK_NO_LOCK in self.device_ids:
            print("Not enough free GPUs!")
            sys.exit()

    @staticmethod
    def parse_options(op, argv=None):
        try:
            load_dic = None
            options = op.parse(argv=argv)
            if options["load_file"].value_given:
                load_dic = IGPUModel.load_checkpoint(options["load_file"].value)
                old_op = load_dic["op"]
                old_op.merge_from(op)
                op = old_op
            op.eval_expr_defaults()
            return op, load_dic
        except OptionMissingException as e:
            print(e)
            op.print_usage()
        except OptionException as e:
            print(e)
        except UnpickleError as e:
            print("Error loading checkpoint:")
            print(e)
        sys.exit()
