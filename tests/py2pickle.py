"""Test helper: writes the byte stream Python 2's cPickle (protocol 2) + NumPy produce for the
object kinds found in the reference's checkpoints and data batches (gpumodel.py:272-281,
data.py:254-273) -- 8-bit `str` (SHORT_BINSTRING / BINSTRING), old-style class instances (OBJ + BUILD),
class references (GLOBAL), ndarrays through numpy.core.multiarray._reconstruct with a raw `str` payload.
No Python 2 exists in this image, so the stream is emitted opcode by opcode."""
import struct

import numpy as np


class Inst(object):
    """An old-style class instance: module.cls with the given attribute dict."""

    def __init__(self, module, cls, attrs):
        self.module, self.cls, self.attrs = module, cls, attrs


class Cls(object):
    """A reference to a class (pickled by name)."""

    def __init__(self, module, name):
        self.module, self.name = module, name


def _s(b):
    b = b if isinstance(b, bytes) else b.encode("latin1")
    return (b"U" + bytes([len(b)]) + b) if len(b) < 256 else (b"T" + struct.pack("<i", len(b)) + b)


def _i(v):
    return b"K" + bytes([v]) if 0 <= v < 256 else b"J" + struct.pack("<i", v)


def _g(mod, name):
    return b"c" + mod.encode() + b"\n" + name.encode() + b"\n"


def _nd(a):
    order_f = a.flags["F_CONTIGUOUS"] and not a.flags["C_CONTIGUOUS"]
    code = {"float32": ("f4", "<"), "uint8": ("u1", "|"), "float64": ("f8", "<")}[a.dtype.name]
    out = _g("numpy.core.multiarray", "_reconstruct") + _g("numpy", "ndarray") + _i(0) + b"\x85" + _s("b") + b"\x87R"
    out += b"(" + _i(1) + b"(" + b"".join(_i(d) for d in a.shape) + b"t"
    out += _g("numpy", "dtype") + b"(" + _s(code[0]) + _i(0) + _i(1) + b"tR(" + _i(3) + _s(code[1]) + b"NNN" + _i(-1) + _i(-1) + _i(0) + b"tb"
    return out + (b"\x88" if order_f else b"\x89") + _s(a.tobytes(order="F" if order_f else "C")) + b"tb"


def _emit(o):
    if o is None:
        return b"N"
    if isinstance(o, bool):
        return b"\x88" if o else b"\x89"
    if isinstance(o, (int, np.integer)):
        return _i(int(o))
    if isinstance(o, float):
        return b"G" + struct.pack(">d", o)
    if isinstance(o, (str, bytes)):
        return _s(o)
    if isinstance(o, np.ndarray):
        return _nd(o)
    if isinstance(o, list):
        return b"](" + b"".join(_emit(x) for x in o) + b"e"
    if isinstance(o, tuple):
        return b"(" + b"".join(_emit(x) for x in o) + b"t"
    if isinstance(o, dict):
        return b"}(" + b"".join(_emit(k) + _emit(v) for k, v in o.items()) + b"u"
    if isinstance(o, Cls):
        return _g(o.module, o.name)
    if isinstance(o, Inst):
        return b"(" + _g(o.module, o.cls) + b"o" + _emit(o.attrs) + b"b"
    raise TypeError(type(o))


def dumps(obj):
    return b"\x80\x02" + _emit(obj) + b"."
