"""ConvNet data providers (Python 3): CIFAR, cropped CIFAR, the dummy provider used by
--check-grads, and a seeded synthetic provider for benchmarks.  Every provider hands the model a
list [data (dims, cases) float32 C-order, labels (1, cases) float32] (Python/convdata.py:31-46,
155-169)."""
import numpy as n
import numpy.random as nr

from data import DataProvider, LabeledDummyDataProvider, LabeledMemoryDataProvider


class CIFARDataProvider(LabeledMemoryDataProvider):
    def __init__(self, data_dir, batch_range, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        LabeledMemoryDataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.data_mean = self.batch_meta["data_mean"]
        self.num_colors = 3
        self.img_size = 32
        for d in self.data_dic:    # mean-subtract once; float32, C order, labels as a row
            d["data"] = n.require(d["data"] - self.data_mean, dtype=n.single, requirements="C")
            d["labels"] = n.require(d["labels"].reshape((1, d["data"].shape[1])), dtype=n.single, requirements="C")

    def get_next_batch(self):
        epoch, batchnum, dic = LabeledMemoryDataProvider.get_next_batch(self)
        return epoch, batchnum, [dic["data"], dic["labels"]]

    def get_data_dims(self, idx=0):
        return self.img_size ** 2 * self.num_colors if idx == 0 else 1

    def get_plottable_data(self, data):
        return n.require((data + self.data_mean).T.reshape(data.shape[1], 3, self.img_size, self.img_size)
                         .swapaxes(1, 3).swapaxes(1, 2) / 255.0, dtype=n.single)


class CroppedCIFARDataProvider(LabeledMemoryDataProvider):
    """Random (train) / centre or 5x2-view (test) crops of border `crop_border`, random mirror
    (Python/convdata.py:74-153).  Two rotating output buffers because the model keeps a view of the
    previous batch while the next one is being cut (Python/convdata.py:105,113)."""

    def __init__(self, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        LabeledMemoryDataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.border_size = self.dp_params["crop_border"]
        self.inner_size = 32 - self.border_size * 2
        self.multiview = self.dp_params["multiview_test"] and test
        self.num_views = 5 * 2
        self.data_mult = self.num_views if self.multiview else 1
        self.num_colors = 3
        for d in self.data_dic:
            d["data"] = n.require(d["data"], requirements="C")
            d["labels"] = n.require(n.tile(d["labels"].reshape((1, d["data"].shape[1])), (1, self.data_mult)), requirements="C")
        self.cropped_data = [n.zeros((self.get_data_dims(), self.data_dic[0]["data"].shape[1] * self.data_mult), dtype=n.single) for _ in range(2)]
        self.batches_generated = 0
        self.data_mean = self.batch_meta["data_mean"].reshape((3, 32, 32))[:, self.border_size:self.border_size + self.inner_size, self.border_size:self.border_size + self.inner_size].reshape((self.get_data_dims(), 1))

    def get_next_batch(self):
        epoch, batchnum, dic = LabeledMemoryDataProvider.get_next_batch(self)
        cropped = self.cropped_data[self.batches_generated % 2]
        self._trim_borders(dic["data"], cropped)
        cropped -= self.data_mean
        self.batches_generated += 1
        return epoch, batchnum, [cropped, dic["labels"]]

    def get_data_dims(self, idx=0):
        return self.inner_size ** 2 * 3 if idx == 0 else 1

    def _trim_borders(self, x, target):
        y = x.reshape(3, 32, 32, x.shape[1])
        b, s = self.border_size, self.inner_size
        if self.test:
            if self.multiview:
                starts = [(0, 0), (0, 2 * b), (b, b), (2 * b, 0), (2 * b, 2 * b)]
                cases = x.shape[1]
                for i, (sy, sx) in enumerate(starts):
                    pic = y[:, sy:sy + s, sx:sx + s, :]
                    target[:, i * cases:(i + 1) * cases] = pic.reshape((self.get_data_dims(), cases))
                    target[:, (self.num_views // 2 + i) * cases:(self.num_views // 2 + i + 1) * cases] = pic[:, :, ::-1, :].reshape((self.get_data_dims(), cases))
            else:
                target[:, :] = y[:, b:b + s, b:b + s, :].reshape((self.get_data_dims(), x.shape[1]))
        else:
            for c in range(x.shape[1]):
                sy, sx = nr.randint(0, 2 * b + 1), nr.randint(0, 2 * b + 1)
                pic = y[:, sy:sy + s, sx:sx + s, c]
                if nr.randint(2) == 0:
                    pic = pic[:, :, ::-1]
                target[:, c] = pic.reshape((self.get_data_dims(),))


class DummyConvNetDataProvider(LabeledDummyDataProvider):
    """`dummy-cn-<dims>`: 512 random cases / 10 classes per batch, transposed to (dims, cases)."""

    def __init__(self, data_dim):
        LabeledDummyDataProvider.__init__(self, data_dim)

    def get_next_batch(self):
        epoch, batchnum, dic = LabeledDummyDataProvider.get_next_batch(self)
        return epoch, batchnum, [n.require(dic["data"].T, requirements="C"), n.require(dic["labels"].T, requirements="C")]

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"] if idx == 0 else 1


class SyntheticDataProvider(DataProvider):
    """Seeded synthetic batches of a fixed shape (no files): data U[0,1) (seed 1234), labels
    uniform ints (seed 5678), the benchmark inputs BASELINE.md section 2 names.  Not in the
    reference; same protocol as the dummy provider."""

    def __init__(self, data_dim, num_classes=10, num_cases=10240, num_batches=1):
        self.batch_range = list(range(1, num_batches + 1))
        self.batch_meta = {"num_vis": data_dim, "label_names": [str(x) for x in range(num_classes)]}
        self.curr_epoch, self.curr_batchnum, self.batch_idx = 1, 1, 0
        self.num_classes = num_classes
        self.data = n.random.default_rng(1234).random((data_dim, num_cases), dtype=n.float32)
        self.labels = n.random.default_rng(5678).integers(0, num_classes, size=(1, num_cases)).astype(n.single)

    def get_num_classes(self):
        return self.num_classes

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"] if idx == 0 else 1

    def get_next_batch(self):
        epoch, batchnum = self.curr_epoch, self.curr_batchnum
        self.advance_batch()
        return epoch, batchnum, [self.data, self.labels]
