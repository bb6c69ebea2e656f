"""GPU parity of the image-layer and element-wise-max kernels (include/mc_kernels.h, image_ops.hip) against the
CPU oracle through the C-ABI, on the same seeded inputs.  Routing / selection ops (bed of nails, eltmax and its
gradient) are compared bit for bit; blur, resize and the colour transforms are short fp32 sums whose fused
multiply-adds the oracle emulates in float64, compared at 1e-6 of the tensor's scale (the L*a*b* cube root,
x ** 0.3333 through powf, at 1e-5).  Sizes cover the vector path (numImages % 4 == 0) and the scalar one."""
import numpy as np
import pytest

from oracle import golden as G
from gpu_util import dev, host, nans, ptr

pytestmark = pytest.mark.gpu


def mc():
    import mckernels
    return mckernels


def close(got, exp, tol):
    scale = max(float(np.abs(exp).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - exp).max()) / scale
    assert err < tol, err


@pytest.mark.parametrize("N,C,S,stride", [(128, 8, 16, 2), (16, 3, 11, 3), (5, 2, 7, 4), (6, 4, 12, 2)])
def test_bed_of_nails_and_undo(N, C, S, stride):
    m = mc()
    rng = np.random.default_rng(31)
    ox = (S + stride - 1) // stride
    start = (S - stride * (ox - 1)) // 2
    x = rng.standard_normal((C * S * S, N)).astype(np.float32)
    g = rng.standard_normal((C * ox * ox, N)).astype(np.float32)
    dx, dg = dev(x), dev(g)
    y = nans(C * ox * ox, N)
    m.call("mc_conv_bed_of_nails", ptr(dx), ptr(y), N, C, S, start, stride, 0.0, 1.0, None)
    assert np.array_equal(host(y), G.bed_of_nails(x, N, C, S, start, stride))
    prev = rng.standard_normal(g.shape).astype(np.float32)
    y2 = dev(prev)
    m.call("mc_conv_bed_of_nails", ptr(dx), ptr(y2), N, C, S, start, stride, 0.5, 2.0, None)
    np.testing.assert_allclose(host(y2), G.bed_of_nails(x, N, C, S, start, stride, target=prev, scale_targets=0.5, scale_output=2.0), rtol=1e-6, atol=1e-6)

    t = nans(C * S * S, N)                                    # scaleTargets == 0 must overwrite every pixel (zeros off the nails)
    m.call("mc_conv_bed_of_nails_undo", ptr(dg), ptr(t), N, C, S, start, stride, 0.0, 1.0, None)
    assert np.array_equal(host(t), G.bed_of_nails_undo(g, N, C, S, start, stride))
    prevx = rng.standard_normal(x.shape).astype(np.float32)
    t2 = dev(prevx)
    m.call("mc_conv_bed_of_nails_undo", ptr(dg), ptr(t2), N, C, S, start, stride, 1.0, 1.0, None)
    assert np.array_equal(host(t2), G.bed_of_nails_undo(g, N, C, S, start, stride, target=prevx, scale_targets=1.0, scale_output=1.0))


@pytest.mark.parametrize("N,C,S,fs", [(128, 3, 32, 5), (16, 8, 12, 3), (5, 2, 9, 9), (6, 3, 7, 7), (4, 1, 3, 3)])
def test_gaussian_blur(N, C, S, fs):
    m = mc()
    rng = np.random.default_rng(32)
    half = fs // 2
    f = np.exp(-(half - np.arange(fs)) ** 2 / (2.0 * 1.3 ** 2)).astype(np.float32)
    f /= f.sum()
    x = rng.standard_normal((C * S * S, N)).astype(np.float32)
    dx, df = dev(x), dev(f.reshape(1, fs))
    for horiz in (1, 0):
        y = nans(*x.shape)
        m.call("mc_conv_gaussian_blur", ptr(dx), ptr(df), ptr(y), N, C, S, half, horiz, 0.0, 1.0, None)
        close(host(y), G.gaussian_blur(x, f, N, C, S, bool(horiz)), 1e-6)
    # the layer's forward pass: rows into the output, then columns IN PLACE (layer.cu:796-799)
    y = nans(*x.shape)
    m.call("mc_conv_gaussian_blur", ptr(dx), ptr(df), ptr(y), N, C, S, half, 1, 0.0, 1.0, None)
    m.call("mc_conv_gaussian_blur", ptr(y), ptr(df), ptr(y), N, C, S, half, 0, 0.0, 1.0, None)
    exp = G.gaussian_blur(G.gaussian_blur(x, f, N, C, S, True), f, N, C, S, False)
    close(host(y), exp, 1e-6)
    # accumulate form (the backward pass into a gradient that already holds a contribution, layer.cu:806)
    prev = rng.standard_normal(x.shape).astype(np.float32)
    t = dev(prev)
    m.call("mc_conv_gaussian_blur", ptr(dx), ptr(df), ptr(t), N, C, S, half, 0, 1.0, 1.0, None)
    close(host(t), G.gaussian_blur(x, f, N, C, S, False, target=prev, scale_targets=1.0, scale_output=1.0), 1e-6)


@pytest.mark.parametrize("N,C,S,scale", [(128, 3, 32, 1.25), (16, 3, 16, 0.8), (5, 2, 15, 2.0), (6, 1, 9, 0.5), (8, 3, 12, 1.0)])
def test_resize_bilinear(N, C, S, scale):
    m = mc()
    rng = np.random.default_rng(33)
    T = int(np.floor(S / scale))
    x = rng.random((C * S * S, N)).astype(np.float32)
    dx, y = dev(x), nans(C * T * T, N)
    m.call("mc_conv_resize_bilinear", ptr(dx), ptr(y), N, C, S, T, float(scale), None)
    close(host(y), G.resize_bilinear(x, N, C, S, T, scale), 1e-6)


@pytest.mark.parametrize("N,P", [(128, 1024), (16, 100), (5, 49), (3, 1)])
def test_colour_transforms(N, P):
    m = mc()
    rng = np.random.default_rng(34)
    x = rng.random((3 * P, N)).astype(np.float32)
    x[:, 0] = 0.0                                             # black: the linear branch of labf
    dx = dev(x)
    y = nans(3 * P, N)
    m.call("mc_conv_rgb_to_yuv", ptr(dx), ptr(y), N, P, None)
    close(host(y), G.rgb_to_yuv(x, N, P), 1e-6)
    for center in (0, 1):
        lab = nans(3 * P, N)
        m.call("mc_conv_rgb_to_lab", ptr(dx), ptr(lab), N, P, center, None)
        close(host(lab), G.rgb_to_lab(x, N, P, bool(center)), 1e-5)


@pytest.mark.parametrize("rows,N", [(192, 128), (7, 5), (1, 3)])
def test_eltwise_max_and_grad(rows, N):
    m = mc()
    rng = np.random.default_rng(35)
    a, b, c = (rng.integers(-2, 3, size=(rows, N)).astype(np.float32) for _ in range(3))   # exact ties on purpose
    da, db, dc = dev(a), dev(b), dev(c)
    out = nans(rows, N)
    m.call("mc_eltwise_max", ptr(db), ptr(da), ptr(out), rows * N, None)                   # input 1 against input 0
    m.call("mc_eltwise_max", ptr(out), ptr(dc), ptr(out), rows * N, None)                  # further inputs in place
    eo = G.eltwise_max([a, b, c])
    assert np.array_equal(host(out), eo)
    g = rng.standard_normal((rows, N)).astype(np.float32)
    dg = dev(g)
    for x, dxx in ((a, da), (b, db), (c, dc)):
        t = nans(rows, N)
        m.call("mc_compute_eltwise_max_grad", ptr(dg), ptr(dxx), ptr(out), ptr(t), 0, rows * N, None)
        assert np.array_equal(host(t), G.eltwise_max_grad(g, x, eo))
        prev = rng.standard_normal((rows, N)).astype(np.float32)
        t2 = dev(prev)
        m.call("mc_compute_eltwise_max_grad", ptr(dg), ptr(dxx), ptr(out), ptr(t2), 1, rows * N, None)
        assert np.array_equal(host(t2), G.eltwise_max_grad(g, x, eo, target=prev, add=True))


def test_image_ops_reject_bad_arguments():
    m = mc()
    x = dev(np.zeros((3 * 16, 4), np.float32))
    y = nans(3 * 16, 4)
    f = dev(np.ones((1, 11), np.float32))
    for name, args in [
        ("mc_conv_bed_of_nails", (ptr(x), ptr(y), 4, 3, 4, 0, 1, 0.0, 1.0, None)),          # stride must exceed 1
        ("mc_conv_bed_of_nails_undo", (ptr(x), ptr(y), 4, 3, 4, 3, 2, 0.0, 1.0, None)),     # last nail outside the image
        ("mc_conv_gaussian_blur", (ptr(x), ptr(f), ptr(y), 4, 3, 4, 5, 1, 0.0, 1.0, None)), # radius 5
        ("mc_conv_gaussian_blur", (ptr(x), ptr(f), ptr(y), 4, 3, 4, 2, 1, 0.0, 1.0, None)), # 5 taps on a 4-pixel line
        ("mc_conv_resize_bilinear", (ptr(x), ptr(x), 4, 3, 4, 4, 1.0, None)),               # in place
        ("mc_conv_rgb_to_yuv", (None, ptr(y), 4, 16, None)),
    ]:
        with pytest.raises(m.McError):
            m.call(name, *args)


@pytest.mark.parametrize("N,ch,P,classes", [(128, 8, 144, 10), (22, 3, 16, 4), (5, 2, 1, 5)])
def test_group_sparsity_in_label_cost_and_grad(N, ch, P, classes):
    m = mc()
    rng = np.random.default_rng(36)
    labels = rng.integers(0, classes, size=(1, N)).astype(np.float32)
    acts = rng.standard_normal((ch * P, N)).astype(np.float32)
    dl, da = dev(labels), dev(acts)
    q, sq, terms, cnt = nans(ch, N), nans(ch, N), nans(ch, N), nans(1, N)
    m.call("mc_group_sparsity_in_label_cost", ptr(dl), ptr(da), ptr(q), ptr(sq), ptr(cnt), ptr(terms), N, ch, P, None)
    cost, S, counts = G.group_sparsity_in_label_cost(labels, acts, ch)
    assert np.array_equal(host(cnt).reshape(-1), counts)
    close(host(sq), S, 1e-5)
    assert abs(float(host(terms).astype(np.float64).sum()) - cost) < 1e-5 * cost
    for sT in (0.0, 1.0):
        prev = rng.standard_normal(acts.shape).astype(np.float32)
        t = nans(*acts.shape) if sT == 0 else dev(prev)
        m.call("mc_group_sparsity_in_label_grad", ptr(da), ptr(sq), ptr(cnt), ptr(t), N, ch, P, sT, -0.5, None)
        close(host(t), G.group_sparsity_in_label_grad(labels, acts, ch, 0.5, target=prev, scale_targets=sT), 1e-5)


def test_group_sparsity_datagen_vector():
    """The HIP kernels on the re-created vector of Kernel/unittest/groupSparsity/DataGen.m (tests/golden/datagen_group_sparsity.npz,
    expected cost and gradient computed in NumPy float64 independently of the oracle): what the reference's test.cu:74-80 prints."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datagen_group_sparsity.npz"))
    N, ch, P = (int(v) for v in z["geom"])
    m = mc()
    dl, da = dev(z["labels"]), dev(z["acts"])
    q, sq, terms, cnt = nans(ch, N), nans(ch, N), nans(ch, N), nans(1, N)
    m.call("mc_group_sparsity_in_label_cost", ptr(dl), ptr(da), ptr(q), ptr(sq), ptr(cnt), ptr(terms), N, ch, P, None)
    cost = float(host(terms).astype(np.float64).sum())
    assert abs(cost - float(z["expected_cost"])) < 1e-5 * cost
    t = nans(ch * P, N)
    m.call("mc_group_sparsity_in_label_grad", ptr(da), ptr(sq), ptr(cnt), ptr(t), N, ch, P, 0.0, 1.0, None)
    np.testing.assert_allclose(host(t), z["expected_grad"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("N,C,P,groups,fc", [(128, 8, 64, 4, 4), (16, 6, 49, 3, 4), (5, 16, 9, 2, 8), (6, 3, 4, 1, 3)])
def test_channel_gather_and_scatter(N, C, P, groups, fc):
    """randSparse plumbing: gather of the connected channels and the per-channel sum of their gradients, bit for bit."""
    import torch
    m = mc()
    rng = np.random.default_rng(37)
    K = groups * fc
    O = K // C
    conns = np.array([int(x) for _ in range(O) for x in rng.permutation(C)], np.int32)
    inv = np.full((C, O), -1, np.int32)
    seen = np.zeros(C, np.int64)
    for k, c in enumerate(conns):
        inv[c, seen[c]] = k
        seen[c] += 1
    x = rng.standard_normal((C * P, N)).astype(np.float32)
    dx = dev(x)
    dconns = torch.from_numpy(conns).cuda()
    dinv = torch.from_numpy(inv.reshape(-1)).cuda()
    y = nans(K * P, N)
    m.call("mc_conv_gather_channels", ptr(dx), dconns.data_ptr(), ptr(y), N, C, K, P, None)
    assert np.array_equal(host(y), G.gather_channels(x, conns, C, N))
    g = rng.standard_normal((K * P, N)).astype(np.float32)
    dg = dev(g)
    t = nans(C * P, N)
    m.call("mc_conv_scatter_channels", ptr(dg), dinv.data_ptr(), ptr(t), N, C, O, P, 0.0, 1.0, None)
    assert np.array_equal(host(t), G.scatter_channels(g, conns, C, N))
    prev = rng.standard_normal(x.shape).astype(np.float32)
    t2 = dev(prev)
    m.call("mc_conv_scatter_channels", ptr(dg), dinv.data_ptr(), ptr(t2), N, C, O, P, 1.0, 1.0, None)
    np.testing.assert_allclose(host(t2), G.scatter_channels(g, conns, C, N, target=prev, scale_targets=1.0), rtol=1e-6, atol=1e-6)
