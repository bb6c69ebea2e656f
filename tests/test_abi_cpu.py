"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mc_kernels.h declares; the `libkernel` extension exposes the reference's seven entry points
(Kernel/src/convnet/pyconvnet.cu:49-57) with their argument checks; the plugin exports the two
plugin symbols (PluginsSrc/src/extralayers.cu:200-215).  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-convnet_amd")


def _declared():
    text = open(os.path.join(ROOT, "include", "mc_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(os.path.join(PKG, "libmckernels.so"))
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libmckernels.so does not export " + n


def test_ctypes_binding_covers_the_header():
    import mckernels
    assert sorted(mckernels.exported_symbols()) == _declared()


def test_rejected_call_reports_an_error_without_a_gpu():
    import mckernels
    L = mckernels.lib()
    # argument validation happens before any device work: a null pointer is rejected with MC_EINVAL
    rc = L.mc_neuron_activate(1, 0.0, 0.0, None, None, 16, None)
    assert rc == -1 and b"null" in L.mc_last_error().lower()
    rc = L.mc_conv_filter_acts(1, 1, 1, 128, 3, 32, 32, 32, 5, 2, 1, 32, 32, 1, 0.0, 1.0, None)   # positive padding
    assert rc == -1 and b"paddingStart" in L.mc_last_error()


def test_libkernel_entry_points():
    import libkernel
    for name in ("initModel", "startBatch", "finishBatch", "checkGradients", "startMultiviewTest", "startFeatureWriter", "syncWithHost"):
        assert callable(getattr(libkernel, name))
    with pytest.raises(TypeError):
        libkernel.initModel("not a list", 128, 0)
    with pytest.raises(RuntimeError):
        libkernel.finishBatch()          # no model yet: an error, not the reference's assert()
    with pytest.raises(RuntimeError):
        libkernel.startBatch([np.zeros((4, 4), np.float32)], 0)


def test_plugin_exports():
    so = os.path.join(PKG, "Plugins", "extralayers.so")
    ctypes.CDLL(os.path.join(PKG, "libkernel.so"), mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(so)
    assert hasattr(lib, "layerConstructor") and hasattr(lib, "neuronConstructor")
