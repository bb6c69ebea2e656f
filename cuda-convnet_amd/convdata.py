"""The providers the ConvNet driver registers (Python 3).  Each hands the model a list
[data (dims, cases) float32 C-ordered, labels (1, cases) float32] -- the two matrices `libkernel.startBatch` takes.

  cifar / mnist     whole mean-subtracted images from `data_batch_<n>` pickles (reference Python/convdata.py:31-87)
  cifar-cropped     training: zero-pad by `crop_border`, cut a random 32x32 window, mirror with probability 1/2; testing: the
                    image itself; nothing subtracted.  This is what the fork's Python/convdata.py:90-153 sets out to do (its
                    training branch cannot run as written -- it names an undefined `border_size` and `n.pic` -- so there is no
                    reference output to pin against: parity unpinned; the test checks the geometry).  The upstream behaviour
                    the fork commented out -- 5 windows x 2 mirrors at test time for --multiview-test -- is kept, on the
                    padded image, so that `startMultiviewTest` has a provider that feeds it.
  dummy-cn-<dims>   random cases for --check-grads
  dummy-synth-<dims> seeded synthetic batches for benchmarks (not in the reference)
"""
import numpy as n
import numpy.random as nr

from data import DataProvider, LabeledDummyDataProvider, LabeledMemoryDataProvider

VIEW_WINDOWS = 5        # four corners + centre ...
NUM_VIEWS = 2 * VIEW_WINDOWS   # ... each also mirrored


def _row(labels, cases):
    return n.require(labels.reshape(1, cases), dtype=n.single, requirements="C")


class MeanSubtractedImages(LabeledMemoryDataProvider):
    """Square `img_size` x `img_size` x `num_colors` images, stored (dims, cases); the dataset mean comes off once, at load."""
    img_size, num_colors = None, None

    def __init__(self, data_dir, batch_range, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        LabeledMemoryDataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.data_mean = self.batch_meta["data_mean"]
        for batch in self.data_dic:
            batch["data"] = n.require(batch["data"] - self.data_mean, dtype=n.single, requirements="C")
            batch["labels"] = _row(batch["labels"], batch["data"].shape[1])

    def get_next_batch(self):
        epoch, batchnum, batch = LabeledMemoryDataProvider.get_next_batch(self)
        return epoch, batchnum, [batch["data"], batch["labels"]]

    def get_data_dims(self, idx=0):
        return self.num_colors * self.img_size ** 2 if idx == 0 else 1

    def get_plottable_data(self, data):
        """(dims, cases) as handed to the model -> (cases, size, size, colors) in [0, 1] for plotting."""
        pics = (data + self.data_mean).T.reshape(data.shape[1], self.num_colors, self.img_size, self.img_size)
        return n.require(pics.transpose(0, 2, 3, 1) / 255.0, dtype=n.single)


class CIFARDataProvider(MeanSubtractedImages):
    img_size, num_colors = 32, 3


class MnistDataProvider(MeanSubtractedImages):
    img_size, num_colors = 28, 1


class CroppedCIFARDataProvider(LabeledMemoryDataProvider):
    img_size, num_colors = 32, 3

    def __init__(self, data_dir, batch_range=None, init_epoch=1, init_batchnum=None, dp_params=None, test=False):
        LabeledMemoryDataProvider.__init__(self, data_dir, batch_range, init_epoch, init_batchnum, dp_params, test)
        self.border_size = self.dp_params["crop_border"]
        self.inner_size = self.img_size                      # the window is as large as the image; the padding gives it room
        self.multiview = bool(self.dp_params.get("multiview_test")) and test
        self.num_views = NUM_VIEWS
        self.data_mult = NUM_VIEWS if self.multiview else 1
        self.data_mean = n.zeros((self.get_data_dims(), 1), dtype=n.single)
        for batch in self.data_dic:
            cases = batch["data"].shape[1]
            batch["data"] = n.require(batch["data"], requirements="C")
            batch["labels"] = _row(n.tile(batch["labels"].reshape(1, cases), (1, self.data_mult)), cases * self.data_mult)
        # two output buffers in turn: the model still reads the previous batch while the next one is cut
        widest = max(b["data"].shape[1] for b in self.data_dic) * self.data_mult
        self.cropped_data = [n.zeros((self.get_data_dims(), widest), dtype=n.single) for _ in range(2)]
        self.batches_generated = 0

    def get_data_dims(self, idx=0):
        return self.num_colors * self.inner_size ** 2 if idx == 0 else 1

    def get_next_batch(self):
        epoch, batchnum, batch = LabeledMemoryDataProvider.get_next_batch(self)
        cases = batch["data"].shape[1] * self.data_mult
        out = self.cropped_data[self.batches_generated % 2][:, :cases]
        if out.shape[1] != self.cropped_data[0].shape[1]:
            out = n.require(out, requirements="C")           # a short last batch: the model wants contiguous rows
        self.batches_generated += 1
        self.cut(batch["data"], out)
        return epoch, batchnum, [out, batch["labels"]]

    def get_plottable_data(self, data):
        pics = data.T.reshape(data.shape[1], self.num_colors, self.inner_size, self.inner_size)
        return n.require(pics.transpose(0, 2, 3, 1) / 255.0, dtype=n.single)

    def cut(self, flat, out):
        """flat (3*32*32, cases) -> out (3*32*32, cases * data_mult)."""
        cases, b, s = flat.shape[1], self.border_size, self.inner_size
        pics = flat.reshape(self.num_colors, self.img_size, self.img_size, cases)
        if self.test and not self.multiview:
            out[:] = flat
            return
        padded = n.pad(pics, ((0, 0), (b, b), (b, b), (0, 0)), "constant", constant_values=0)
        if self.test:
            corners = [(0, 0), (0, 2 * b), (b, b), (2 * b, 0), (2 * b, 2 * b)]
            for v, (y, x) in enumerate(corners):
                window = padded[:, y:y + s, x:x + s, :]
                out[:, v * cases:(v + 1) * cases] = window.reshape(-1, cases)
                out[:, (VIEW_WINDOWS + v) * cases:(VIEW_WINDOWS + v + 1) * cases] = window[:, :, ::-1, :].reshape(-1, cases)
            return
        for c in range(cases):
            y, x = nr.randint(0, 2 * b + 1), nr.randint(0, 2 * b + 1)
            window = padded[:, y:y + s, x:x + s, c]
            out[:, c] = (window[:, :, ::-1] if nr.randint(2) == 0 else window).reshape(-1)


class DummyConvNetDataProvider(LabeledDummyDataProvider):
    """`dummy-cn-<dims>`: the labeled dummy's 512 random cases in 10 classes, turned to (dims, cases)."""

    def get_next_batch(self):
        epoch, batchnum, batch = LabeledDummyDataProvider.get_next_batch(self)
        return epoch, batchnum, [n.require(batch[k].T, requirements="C") for k in ("data", "labels")]

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"] if idx == 0 else 1


class SyntheticDataProvider(DataProvider):
    """`dummy-synth-<dims>`: one fixed batch, data U[0,1) from seed 1234 and integer labels from seed 5678 -- the benchmark
    inputs BASELINE.md section 2 names.  Same protocol as the dummy providers; the same arrays every time."""

    def __init__(self, data_dim, num_classes=10, num_cases=10240, num_batches=1):
        self._start(range(1, num_batches + 1))
        self.num_classes = num_classes
        self.batch_meta = {"num_vis": data_dim, "label_names": [str(c) for c in range(num_classes)]}
        self.data = n.random.default_rng(1234).random((data_dim, num_cases), dtype=n.float32)
        self.labels = n.random.default_rng(5678).integers(0, num_classes, size=(1, num_cases)).astype(n.single)

    def get_num_classes(self):
        return self.num_classes

    def get_data_dims(self, idx=0):
        return self.batch_meta["num_vis"] if idx == 0 else 1

    def get_next_batch(self):
        return self._take() + ([self.data, self.labels],)
