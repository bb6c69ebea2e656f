"""Data-provider protocol and the file-backed / dummy providers (Python 3).

A provider answers three questions for the driver -- `get_next_batch()` -> (epoch, batchnum, data), `get_data_dims(idx)`,
`get_num_classes()` -- and is looked up by the `--data-provider` string in one registry (`PROVIDERS`; `dp_types` /
`dp_classes` are the reference's two views of it, kept because gpumodel / convnet print and extend them).  The on-disk
format is the reference's (Python/data.py): a directory with `batches.meta` and `data_batch_<n>` pickles (optionally split
into `data_batch_<n>.1`, `.2`, ... sub-batches), batch numbers that cycle through `batch_range` with the epoch counting the
wrap-arounds, and `dummy-<kind>-<dims>` type strings that carry the dimensionality in the name.

Out of scope: the reference's JPEG (`jpeg`, `cropped-jpeg`, ...) and `mat` providers -- they need PIL /
scipy.io datasets that are not part of the training hot path."""
import os
import re

import numpy as n
import numpy.random as nr

from util import alphanum_key, unpickle

BATCH_META_FILE = "batches.meta"


class DataProviderException(Exception):
    pass


class BatchCursor(object):
    """Where a provider is in its cycle over `batch_range`: (epoch, batch number), and how to step."""

    def __init__(self, batch_range, epoch=1, batchnum=None):
        self.order = list(batch_range)
        self.epoch = epoch
        self.pos = self.order.index(batchnum) if batchnum in self.order else 0

    @property
    def batchnum(self):
        return self.order[self.pos]

    def peek(self):
        return (self.pos + 1) % len(self.order)

    def step(self):
        self.pos = self.peek()
        self.epoch += self.pos == 0       # back at the first batch: a new epoch


class DataProvider(object):
    BATCH_REGEX = re.compile(r"^data_batch_(\d+)(\.\d+)?$")

    def __init__(self, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        self.data_dir, self.test = data_dir, test
        self.dp_params = {} if dp_params is None else dp_params
        self._start(DataProvider.get_batch_nums(data_dir) if batch_range is None else batch_range, init_epoch, init_batchnum)
        self.batch_meta = self.get_batch_meta(data_dir)
        self.data_dic = None

    def _start(self, batch_range, epoch=1, batchnum=None):
        self._cursor = BatchCursor(batch_range, epoch, batchnum)
        self.batch_range = self._cursor.order

    # the cursor under the reference's attribute names (drivers and checkpoints read these)
    curr_epoch = property(lambda self: self._cursor.epoch, lambda self, e: setattr(self._cursor, "epoch", e))
    curr_batchnum = property(lambda self: self._cursor.batchnum, lambda self, b: setattr(self._cursor, "pos", self.batch_range.index(b)))
    batch_idx = property(lambda self: self._cursor.pos, lambda self, i: setattr(self._cursor, "pos", i))

    # ------------------------------------------------------------------ iteration
    def _take(self):
        """(epoch, batchnum) of the batch being handed out; moves on to the next one."""
        here = self._cursor.epoch, self._cursor.batchnum
        self._cursor.step()
        return here

    def get_next_batch(self):
        if self.data_dic is None or len(self.batch_range) > 1:       # a one-batch range is read once
            self.data_dic = self.get_batch(self.curr_batchnum)
        return self._take() + (self.data_dic,)

    def advance_batch(self):
        self._cursor.step()

    def get_next_batch_idx(self):
        return self._cursor.peek()

    def get_next_batch_num(self):
        return self.batch_range[self._cursor.peek()]

    # ------------------------------------------------------------------ files
    def get_data_file_name(self, batchnum=None):
        return os.path.join(self.data_dir, "data_batch_%d" % (self.curr_batchnum if batchnum is None else batchnum))

    def get_batch(self, batch_num):
        whole = self.get_data_file_name(batch_num)
        pieces = []
        while os.path.exists("%s.%d" % (whole, len(pieces) + 1)):
            pieces.append(unpickle("%s.%d" % (whole, len(pieces) + 1)))
        if not pieces:
            return unpickle(whole)
        batch = pieces[0]
        if len(pieces) > 1:
            batch["data"] = n.concatenate([p["data"] for p in pieces], axis=0)
        return batch

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"]

    @staticmethod
    def get_batch_meta(data_dir):
        return unpickle(os.path.join(data_dir, BATCH_META_FILE))

    @staticmethod
    def get_batch_filenames(srcdir):
        return sorted((f for f in os.listdir(srcdir) if DataProvider.BATCH_REGEX.match(f)), key=alphanum_key)

    @staticmethod
    def get_batch_nums(srcdir):
        return sorted(set(int(DataProvider.BATCH_REGEX.match(f).group(1)) for f in DataProvider.get_batch_filenames(srcdir)))

    @staticmethod
    def get_num_batches(srcdir):
        return len(DataProvider.get_batch_nums(srcdir))

    # ------------------------------------------------------------------ registry
    @classmethod
    def get_instance(cls, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, type="default", dp_params=None, test=False):
        kind = type or DataProvider.get_batch_meta(data_dir)["dp_type"]      # no --data-provider: the data names its own
        head, _, tail = kind.rpartition("-")
        if kind.startswith("dummy-") and head + "-n" in dp_classes and tail.isdigit():
            return dp_classes[head + "-n"](int(tail))                        # "dummy-cn-3072": the dimensionality rides in the name
        if kind not in dp_classes or kind.endswith("-n"):
            raise DataProviderException("No such data provider: %s" % kind)
        return dp_classes[kind](data_dir, batch_range, init_epoch, init_batchnum, {} if dp_params is None else dp_params, test)

    @classmethod
    def register_data_provider(cls, name, desc, _class):
        if dp_classes.get(name, _class) is not _class:
            raise DataProviderException("Data provider %s already registered" % name)
        dp_types[name], dp_classes[name] = desc, _class


class LabeledDataProvider(DataProvider):
    def get_num_classes(self):
        return len(self.batch_meta["label_names"])


class MemoryDataProvider(DataProvider):
    """Every batch of the range is read at construction; get_next_batch only indexes."""

    def __init__(self, data_dir, batch_range, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        DataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.data_dic = [self._resident(b) for b in self.batch_range]

    def _resident(self, batchnum):
        return self.get_batch(batchnum)

    def get_next_batch(self):
        epoch, batchnum = self._take()
        return epoch, batchnum, self.data_dic[self.batch_range.index(batchnum)]


class LabeledMemoryDataProvider(MemoryDataProvider, LabeledDataProvider):
    def _resident(self, batchnum):
        batch = unpickle(self.get_data_file_name(batchnum))
        batch["labels"] = n.asarray(batch["labels"], dtype=n.single).reshape(-1, 1)
        return batch


class DummyDataProvider(DataProvider):
    """No files: uniform-random (512, dims) batches."""
    num_cases = 512

    def __init__(self, data_dim):
        self._start([1])
        self.batch_meta = {"num_vis": data_dim, "data_in_rows": True}

    def _draw(self):
        return {"data": nr.rand(self.num_cases, self.get_data_dims()).astype(n.single)}

    def get_next_batch(self):
        self.advance_batch()          # (the reference's dummies step first, so the first batch reports epoch 2)
        return self.curr_epoch, self.curr_batchnum, self._draw()


class LabeledDummyDataProvider(DummyDataProvider):
    """... plus a (cases, 1) column of uniform-random integer labels."""

    def __init__(self, data_dim, num_classes=10, num_cases=512):
        DummyDataProvider.__init__(self, data_dim)
        self.num_classes, self.num_cases = num_classes, num_cases
        self.batch_meta["label_names"] = [str(c) for c in range(num_classes)]

    def get_num_classes(self):
        return self.num_classes

    def _draw(self):
        batch = DummyDataProvider._draw(self)
        batch["labels"] = nr.randint(0, self.num_classes, self.num_cases).astype(n.single).reshape(-1, 1)
        return batch


PROVIDERS = [   # (name, class, description)
    ("default", DataProvider, "The default data provider; loads one batch into memory at a time"),
    ("memory", MemoryDataProvider, "Loads the entire dataset into memory"),
    ("labeled", LabeledDataProvider, "Returns data and labels (used by classifiers)"),
    ("labeled-memory", LabeledMemoryDataProvider, "Combination labeled + memory"),
    ("dummy-n", DummyDataProvider, "Dummy data provider for n-dimensional data"),
    ("dummy-labeled-n", LabeledDummyDataProvider, "Labeled dummy data provider for n-dimensional data"),
]
dp_types = dict((name, desc) for name, _, desc in PROVIDERS)
dp_classes = dict((name, klass) for name, klass, _ in PROVIDERS)
