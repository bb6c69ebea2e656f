"""mc_conv_weight_grad_noalloc (the entry point that allocates nothing: split-K slabs in a library-owned device array)
against mc_conv_weight_grad (caller's workspace) through the C-ABI: bit-identical, on two calls in a row, the second
with a smaller input (the array keeps the first call's contents), and against the CPU oracle at rel 1e-4.
Collected last (file name): a failure here must not hide the other tests from `pytest -x`."""
import numpy as np
import pytest

from oracle import golden as G
from gpu_util import dev, host, nans, ptr, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _modules(img, fs, pad, stride):
    return 1 + int(np.ceil((2 * pad + img - fs) / float(stride)))


def _both(m, rng, N, C, H, fs, F, pad, stride, groups, scale_t, scale_o):
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)
    wshape = ((C // groups) * fs * fs, F)
    prev = rng.standard_normal(wshape).astype(np.float32)
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, groups)
    assert wsb <= m.lib().mc_noalloc_scratch_bytes()
    ws = nans(max(1, wsb // 4))
    dimg, dhid = dev(img), dev(hid)
    ref = dev(prev) if scale_t else nans(*wshape)
    m.call("mc_conv_weight_grad", ptr(dimg), ptr(dhid), ptr(ref), *geo, scale_t, scale_o, ptr(ws), wsb, None)
    got = dev(prev) if scale_t else nans(*wshape)
    m.call("mc_conv_weight_grad_noalloc", ptr(dimg), ptr(dhid), ptr(got), *geo, scale_t, scale_o, None)
    exp = scale_t * prev + scale_o * G.weight_acts(img, hid, *geo, partial_sum=0)
    return host(ref), host(got), exp, wsb


def test_noalloc_weight_grad_equals_workspace_form_twice_second_smaller():
    import mckernels as m
    rng = np.random.default_rng(77)
    # first call: NIN conv3 shape at 64 cases (several K chunks: the slabs fill a good part of the array) ...
    ref, got, exp, wsb1 = _both(m, rng, 64, 192, 8, 3, 192, 1, 1, 1, 0.0, 1.0)
    assert wsb1 > 0, "the case must actually use the scratch array"
    assert np.array_equal(ref, got)
    assert relerr(got, exp) < TOL
    # ... second call right after it, smaller input, accumulate form: stale slab contents must not leak in
    ref, got, exp, wsb2 = _both(m, rng, 32, 32, 8, 5, 32, 2, 1, 1, 0.5, 0.01)
    assert 0 < wsb2 < wsb1
    assert np.array_equal(ref, got)
    assert relerr(got, exp) < TOL
    # a grouped, strided, odd-sized shape (scalar loads, tails)
    ref, got, exp, _ = _both(m, rng, 12, 8, 9, 3, 16, 1, 2, 2, 0.0, 2.0)
    assert np.array_equal(ref, got)
    assert relerr(got, exp) < TOL


def test_noalloc_entry_first_on_a_fresh_shape_under_stream_capture():
    """The headline contract, not only the bits: on a shape the library has never seen (no cached tile: the static rule's tile
    runs) the entry point is called FIRST, inside a stream capture -- where hipMalloc, hipFree, event queries and blocking calls
    are illegal and would invalidate the capture; the device's free memory is the same before and after a direct call.  The captured
    graph is launched twice; the result equals the workspace form (called afterwards, which also tunes the shape) bit for bit."""
    import ctypes
    import torch
    import mckernels as m
    hip = ctypes.CDLL("libamdhip64.so")   # the runtime torch already loaded
    rng = np.random.default_rng(78)
    N, C, H, fs, F, pad, stride, groups = 48, 24, 11, 3, 40, 1, 1, 1          # used by no other test: untuned
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)
    wshape = ((C // groups) * fs * fs, F)
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, groups)
    dimg, dhid, got, ref, ws = dev(img), dev(hid), nans(*wshape), nans(*wshape), nans(max(1, wsb // 4))
    # the library's once-per-device set-up (hipGetSymbolAddress of the array, kernel attributes) happens on the first call on a
    # device: make that call here, on another shape, so that the captured call below is an ordinary one when this test runs alone
    _both(m, rng, 8, 4, 6, 3, 8, 1, 1, 1, 0.0, 1.0)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    handle = ctypes.c_void_p(st.cuda_stream)
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(handle, 2) == 0                          # hipStreamCaptureModeRelaxed
    rc = m.lib().mc_conv_weight_grad_noalloc(ptr(dimg), ptr(dhid), ptr(got), *geo, 0.0, 1.0, handle)
    assert hip.hipStreamEndCapture(handle, ctypes.byref(graph)) == 0 and graph.value, "the capture was invalidated"
    assert rc == 0, m.lib().mc_last_error().decode()
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    for _ in range(2):
        assert hip.hipGraphLaunch(gexec, handle) == 0
    assert hip.hipStreamSynchronize(handle) == 0
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    # free device memory across a direct call (the HIP runtime itself takes a few MiB for a capture and for the first launch of a
    # kernel, so the capture above is not the place to measure): the shape's kernels have run, the shape is still untuned
    free0, free1, tot = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(tot)) == 0
    m.call("mc_conv_weight_grad_noalloc", ptr(dimg), ptr(dhid), ptr(got), *geo, 0.0, 1.0, handle)
    assert hip.hipStreamSynchronize(handle) == 0
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(tot)) == 0
    # (>=: the runtime may hand back what the destroyed graph held; what must not happen is that less is free)
    assert free1.value >= free0.value, "the entry point took device memory"
    m.call("mc_conv_weight_grad", ptr(dimg), ptr(dhid), ptr(ref), *geo, 0.0, 1.0, ptr(ws), wsb, None)
    assert np.array_equal(host(got), host(ref))
    assert relerr(host(got), G.weight_acts(img, hid, *geo, partial_sum=0)) < TOL


def test_noalloc_array_covers_the_baseline_layers():
    """The slabs of one launch are (workgroups in one round) x (tile): at most 256 x 192 x 128 floats = 25.2 MB on this
    device, so the 32 MiB array serves every shape the chunk plan produces here; the size check in the entry point guards
    other CU counts / plans (it cannot be made to fire on a 256-CU device, which is why no rejection case is run)."""
    import mckernels as m
    L = m.lib()
    cap = L.mc_noalloc_scratch_bytes()
    assert cap == 32 << 20
    # N, C, F, fs, modules, groups: NIN conv1-3, layers-80sec conv1-3, conv-local conv2, imagenet conv1-5
    for shp in [(128, 3, 192, 5, 1024, 1), (128, 96, 192, 5, 256, 1), (128, 192, 192, 3, 64, 1), (128, 3, 32, 5, 1024, 1),
                (128, 32, 32, 5, 256, 1), (128, 32, 64, 5, 64, 1), (128, 64, 64, 5, 256, 1), (128, 3, 96, 11, 3025, 1),
                (128, 96, 256, 5, 729, 1), (128, 256, 192, 3, 169, 1), (128, 192, 192, 3, 169, 1), (128, 192, 128, 3, 169, 1)]:
        assert L.mc_conv_weight_grad_workspace_bytes(*shp) <= cap, shp
