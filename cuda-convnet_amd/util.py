"""Small helpers of the driver (Python 3): checkpoint pickling, natural sort, GPU id passthrough.
Behaviour follows the reference's Python/util.py:26-110."""
import gzip
import os
import pickle as _pickle
import platform
import re

GPU_LOCK_NO_SCRIPT = -2
GPU_LOCK_NO_LOCK = -1


class UnpickleError(Exception):
    pass


def get_gpu_lock(id=-1):
    """The reference consults an optional site script; without one it passes an explicit id through
    and maps "any" (-1) to GPU_LOCK_NO_SCRIPT, which the native module reads as "pick for me"
    (Python/util.py:48-58).  There is no such script here."""
    return GPU_LOCK_NO_SCRIPT if id < 0 else id


def pickle(filename, data, compress=False):
    opener = gzip.open if compress else open
    with opener(filename, "wb") as fo:
        _pickle.dump(data, fo, protocol=_pickle.HIGHEST_PROTOCOL)


def unpickle(filename):
    """Loads a checkpoint / data batch.  Python-2 cPickle files (the reference's own format) carry their
    NumPy payloads as 8-bit `str`: encoding='latin1' maps those bytes one to one (the default 'ASCII'
    raises on the first byte above 127, or silently mangles a payload that happens to be all-ASCII);
    the argument has no effect on files Python 3 wrote."""
    if not os.path.exists(filename):
        raise UnpickleError("Path '%s' does not exist." % filename)
    with open(filename, "rb") as fo:
        magic = fo.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(filename, "rb") as fo:
        return _pickle.load(fo, encoding="latin1")


def is_windows_machine():
    return platform.system() == "Windows"


def alphanum_key(s):
    """Natural sort key: 'data_batch_10' after 'data_batch_9'."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]
