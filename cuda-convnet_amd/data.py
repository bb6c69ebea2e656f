"""Data providers (Python 3).  Mirrors the provider protocol of the reference's Python/data.py:
`get_next_batch()` -> (epoch, batchnum, data), batch ranges that wrap and bump the epoch,
`get_data_dims(idx)`, `get_num_classes()`, and the type registry with the `dummy-<name>-<dims>`
convention (Python/data.py:108-124, 275-286).  The JPEG / mat providers of the reference are not
part of the training hot path (they need PIL and real datasets) and are not provided."""
import os
import re

import numpy as n
import numpy.random as nr

from util import alphanum_key, unpickle

BATCH_META_FILE = "batches.meta"


class DataProviderException(Exception):
    pass


class DataProvider(object):
    BATCH_REGEX = re.compile(r"^data_batch_(\d+)(\.\d+)?$")

    def __init__(self, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        if batch_range is None:
            batch_range = DataProvider.get_batch_nums(data_dir)
        if init_batchnum is None or init_batchnum not in batch_range:
            init_batchnum = batch_range[0]
        self.data_dir = data_dir
        self.batch_range = list(batch_range)
        self.curr_epoch = init_epoch
        self.curr_batchnum = init_batchnum
        self.dp_params = dp_params or {}
        self.batch_meta = self.get_batch_meta(data_dir)
        self.data_dic = None
        self.test = test
        self.batch_idx = self.batch_range.index(init_batchnum)

    # -- iteration -----------------------------------------------------
    def get_next_batch(self):
        if self.data_dic is None or len(self.batch_range) > 1:
            self.data_dic = self.get_batch(self.curr_batchnum)
        epoch, batchnum = self.curr_epoch, self.curr_batchnum
        self.advance_batch()
        return epoch, batchnum, self.data_dic

    def advance_batch(self):
        self.batch_idx = self.get_next_batch_idx()
        self.curr_batchnum = self.batch_range[self.batch_idx]
        if self.batch_idx == 0:      # wrapped around: next epoch
            self.curr_epoch += 1

    def get_next_batch_idx(self):
        return (self.batch_idx + 1) % len(self.batch_range)

    def get_next_batch_num(self):
        return self.batch_range[self.get_next_batch_idx()]

    # -- files -----------------------------------------------------------
    def get_data_file_name(self, batchnum=None):
        if batchnum is None:
            batchnum = self.curr_batchnum
        return os.path.join(self.data_dir, "data_batch_%d" % batchnum)

    def get_batch(self, batch_num):
        base = self.get_data_file_name(batch_num)
        if not os.path.exists(base + ".1"):
            return unpickle(base)
        dic = unpickle(base + ".1")          # batch stored as sub-batches .1, .2, ...
        idx = 2
        while os.path.exists("%s.%d" % (base, idx)):
            dic["data"] = n.r_[dic["data"], unpickle("%s.%d" % (base, idx))["data"]]
            idx += 1
        return dic

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"]

    @staticmethod
    def get_batch_meta(data_dir):
        return unpickle(os.path.join(data_dir, BATCH_META_FILE))

    @staticmethod
    def get_batch_filenames(srcdir):
        return sorted([f for f in os.listdir(srcdir) if DataProvider.BATCH_REGEX.match(f)], key=alphanum_key)

    @staticmethod
    def get_batch_nums(srcdir):
        return sorted(set(int(DataProvider.BATCH_REGEX.match(f).group(1)) for f in DataProvider.get_batch_filenames(srcdir)))

    @staticmethod
    def get_num_batches(srcdir):
        return len(DataProvider.get_batch_nums(srcdir))

    # -- registry ----------------------------------------------------------
    @classmethod
    def get_instance(cls, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, type="default", dp_params=None, test=False):
        type = type or DataProvider.get_batch_meta(data_dir)["dp_type"]     # the data may name its provider
        if type.startswith("dummy-"):
            name = "-".join(type.split("-")[:-1]) + "-n"
            if name not in dp_types:
                raise DataProviderException("No such data provider: %s" % type)
            return dp_classes[name](int(type.split("-")[-1]))
        if type in dp_types:
            return dp_classes[type](data_dir, batch_range, init_epoch, init_batchnum, dp_params or {}, test)
        raise DataProviderException("No such data provider: %s" % type)

    @classmethod
    def register_data_provider(cls, name, desc, _class):
        if name in dp_types:
            if dp_classes[name] is _class:
                return
            raise DataProviderException("Data provider %s already registered" % name)
        dp_types[name] = desc
        dp_classes[name] = _class


class DummyDataProvider(DataProvider):
    def __init__(self, data_dim):
        self.batch_range = [1]
        self.batch_meta = {"num_vis": data_dim, "data_in_rows": True}
        self.curr_epoch = 1
        self.curr_batchnum = 1
        self.batch_idx = 0

    def get_next_batch(self):
        self.advance_batch()
        return self.curr_epoch, self.curr_batchnum, {"data": nr.rand(512, self.get_data_dims()).astype(n.single)}


class LabeledDummyDataProvider(DummyDataProvider):
    """512 uniform-random cases with uniform-random integer labels per batch (Python/data.py:217-239)."""

    def __init__(self, data_dim, num_classes=10, num_cases=512):
        DummyDataProvider.__init__(self, data_dim)
        self.batch_meta["label_names"] = [str(x) for x in range(num_classes)]
        self.num_cases = num_cases
        self.num_classes = num_classes

    def get_num_classes(self):
        return self.num_classes

    def get_next_batch(self):
        self.advance_batch()
        data = nr.rand(self.num_cases, self.get_data_dims()).astype(n.single)
        labels = n.require(n.c_[nr.randint(0, self.num_classes, self.num_cases)], requirements="C", dtype=n.single)
        return self.curr_epoch, self.curr_batchnum, {"data": data, "labels": labels}


class MemoryDataProvider(DataProvider):
    def __init__(self, data_dir, batch_range, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        DataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.data_dic = [self.get_batch(i) for i in self.batch_range]

    def get_next_batch(self):
        epoch, batchnum = self.curr_epoch, self.curr_batchnum
        self.advance_batch()
        return epoch, batchnum, self.data_dic[batchnum - self.batch_range[0]]


class LabeledDataProvider(DataProvider):
    def get_num_classes(self):
        return len(self.batch_meta["label_names"])


class LabeledMemoryDataProvider(LabeledDataProvider):
    def __init__(self, data_dir, batch_range, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        LabeledDataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.data_dic = []
        for i in self.batch_range:
            d = unpickle(self.get_data_file_name(i))
            d["labels"] = n.c_[n.require(d["labels"], dtype=n.single)]
            self.data_dic.append(d)

    def get_next_batch(self):
        epoch, batchnum = self.curr_epoch, self.curr_batchnum
        self.advance_batch()
        return epoch, batchnum, self.data_dic[batchnum - self.batch_range[0]]


dp_types = {"default": "The default data provider; loads one batch into memory at a time",
            "memory": "Loads the entire dataset into memory",
            "labeled": "Returns data and labels (used by classifiers)",
            "labeled-memory": "Combination labeled + memory",
            "dummy-n": "Dummy data provider for n-dimensional data",
            "dummy-labeled-n": "Labeled dummy data provider for n-dimensional data"}
dp_classes = {"default": DataProvider,
              "memory": MemoryDataProvider,
              "labeled": LabeledDataProvider,
              "labeled-memory": LabeledMemoryDataProvider,
              "dummy-n": DummyDataProvider,
              "dummy-labeled-n": LabeledDummyDataProvider}
