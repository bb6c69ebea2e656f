"""Oracle checks for the image layers and the element-wise max (nailbed, blur, resize, rgb2yuv,
rgb2lab, eltmax).  The reference holds no vectors for these kernels (no unit test covers conv_util.cu's image
ops): parity is UNPINNED BY REFERENCE DATA.  What pins the NumPy restatement here is independent of it:
SciPy's separable correlation and linear interpolation, the defining matrices / known colours of the two
colour spaces, and the adjoint identity <op(x), g> == <x, undo(g)> for the two ops that have a backward pass."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import golden as G


def _img(rng, C, S, N, lo=0.0, hi=1.0):
    return (lo + (hi - lo) * rng.random((C * S * S, N))).astype(np.float32)


@pytest.mark.parametrize("fs,stdev", [(3, 0.8), (5, 1.2), (7, 2.0), (9, 3.0)])
def test_blur_is_zero_padded_separable_correlation(fs, stdev):
    rng = np.random.default_rng(1)
    N, C, S = 6, 2, 11
    half = fs // 2
    f = np.exp(-(half - np.arange(fs)) ** 2 / (2.0 * stdev ** 2)).astype(np.float32)     # layer.py:443-446 of the reference
    f /= f.sum()
    x = _img(rng, C, S, N, -1, 1)
    x4 = x.reshape(C, S, S, N).astype(np.float64)
    for horiz, axis in ((True, 2), (False, 1)):
        ref = ndimage.correlate1d(x4, f.astype(np.float64), axis=axis, mode="constant", cval=0.0)
        got = G.gaussian_blur(x, f, N, C, S, horiz)
        np.testing.assert_allclose(got, ref.reshape(x.shape), rtol=0, atol=2e-7)
    # accumulate form and the symmetric filter's self-adjointness (why the layer's backward pass is the same blur)
    prev = _img(rng, C, S, N)
    got = G.gaussian_blur(x, f, N, C, S, True, target=prev, scale_targets=0.5, scale_output=2.0)
    np.testing.assert_allclose(got, 0.5 * prev + 2.0 * G.gaussian_blur(x, f, N, C, S, True), rtol=1e-6, atol=1e-6)
    g = _img(rng, C, S, N, -1, 1)
    lhs = float((G.gaussian_blur(x, f, N, C, S, False).astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * G.gaussian_blur(g, f, N, C, S, False)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


@pytest.mark.parametrize("S,stride", [(16, 2), (12, 2), (11, 3), (7, 4)])
def test_bed_of_nails_picks_the_nails_and_undo_is_its_adjoint(S, stride):
    rng = np.random.default_rng(2)
    N, C = 5, 3
    ox = (S + stride - 1) // stride
    start = (S - stride * (ox - 1)) // 2                      # NailbedLayerParser, layer.py:413-414 of the reference
    x = _img(rng, C, S, N, -1, 1)
    y = G.bed_of_nails(x, N, C, S, start, stride)
    x4 = x.reshape(C, S, S, N)
    for c in range(C):
        for oy in range(ox):
            for oxx in range(ox):
                np.testing.assert_array_equal(y.reshape(C, ox, ox, N)[c, oy, oxx], x4[c, start + oy * stride, start + oxx * stride])
    g = _img(rng, C, ox, N, -1, 1)
    back = G.bed_of_nails_undo(g, N, C, S, start, stride)
    assert np.count_nonzero(back) <= g.size
    assert abs(float((y.astype(np.float64) * g).sum()) - float((x.astype(np.float64) * back).sum())) < 1e-5
    # accumulating undo: only the nail pixels change
    prev = _img(rng, C, S, N)
    acc = G.bed_of_nails_undo(g, N, C, S, start, stride, target=prev, scale_targets=1.0, scale_output=1.0)
    changed = (acc != prev).reshape(C, S, S, N)
    mask = np.zeros((S, S), bool)
    mask[start::stride, start::stride] = True
    assert not changed[:, ~mask, :].any()
    np.testing.assert_allclose(acc, prev + back, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("S,scale", [(16, 1.25), (16, 0.8), (15, 2.0), (9, 0.5), (12, 1.0)])
def test_resize_is_clamped_bilinear_interpolation(S, scale):
    rng = np.random.default_rng(3)
    N, C = 4, 2
    T = int(np.floor(S / scale))                              # ResizeLayerParser, layer.py:470 of the reference
    x = _img(rng, C, S, N)
    got = G.resize_bilinear(x, N, C, S, T, scale).reshape(C, T, T, N)
    centre = S * 0.5 - T * 0.5 * scale
    src = np.clip(np.arange(T) * scale + centre, 0.0, S - 1.01)
    yy, xx = np.meshgrid(src, src, indexing="ij")
    for c in range(C):
        for n in range(N):
            ref = ndimage.map_coordinates(x.reshape(C, S, S, N)[c, :, :, n].astype(np.float64), [yy, xx], order=1)
            np.testing.assert_allclose(got[c, :, :, n], ref, rtol=0, atol=3e-5)     # coordinates are float32 in the kernel
    assert got.min() >= x.min() - 1e-6 and got.max() <= x.max() + 1e-6          # a convex combination of 4 pixels


def test_colour_transforms():
    rng = np.random.default_rng(4)
    N, P = 7, 20
    rgb = rng.random((3 * P, N)).astype(np.float32)
    yuv = G.rgb_to_yuv(rgb, N, P).reshape(3, P, N)
    M = np.array([[0.2126, 0.7152, 0.0722], [-0.09991, -0.33609, 0.436], [0.615, -0.55861, -0.05639]])
    np.testing.assert_allclose(yuv, np.einsum("oc,cpn->opn", M, rgb.reshape(3, P, N).astype(np.float64)), rtol=0, atol=1e-6)
    grey = np.tile(rng.random((1, P, N)).astype(np.float32), (3, 1, 1)).reshape(3 * P, N)
    g = G.rgb_to_yuv(grey, N, P).reshape(3, P, N)
    np.testing.assert_allclose(g[0], grey.reshape(3, P, N)[0], atol=1e-6)          # Y of a grey pixel is the grey level
    assert np.abs(g[1:]).max() < 1e-4                                              # and it has no chroma

    # L*a*b*: black -> L = 0, white -> L = 100 (the kernel's XYZ rows each sum to 1), greys have a = b = 0
    ends = np.zeros((3, 2, 1), np.float32)
    ends[:, 1] = 1.0
    lab = G.rgb_to_lab(ends.reshape(6, 1), 1, 2).reshape(3, 2)
    np.testing.assert_allclose(lab[:, 0], [0.0, 0.0, 0.0], atol=2e-4)
    np.testing.assert_allclose(lab[:, 1], [100.0, 0.0, 0.0], atol=2e-2)            # 1 ** 0.3333 and row sums of 0.99997
    labc = G.rgb_to_lab(ends.reshape(6, 1), 1, 2, center=True).reshape(3, 2)
    np.testing.assert_allclose(labc[0], lab[0] - 50.0, atol=1e-5)
    lg = G.rgb_to_lab(grey, N, P).reshape(3, P, N)
    assert np.abs(lg[1:]).max() < 0.05
    assert (np.diff(G.rgb_to_lab(np.repeat(np.linspace(0, 1, 50, dtype=np.float32)[None], 3, 0).reshape(150, 1), 1, 50)
                    .reshape(3, 50)[0]) > 0).all()                                  # L* is monotone in the grey level


def test_eltwise_max_and_grad_routing():
    rng = np.random.default_rng(5)
    a, b, c = (rng.integers(-2, 3, size=(6, 9)).astype(np.float32) for _ in range(3))    # small ints: many exact ties
    out = G.eltwise_max([a, b, c])
    np.testing.assert_array_equal(out, np.maximum(np.maximum(a, b), c))
    g = rng.standard_normal(out.shape).astype(np.float32)
    grads = [G.eltwise_max_grad(g, x, out) for x in (a, b, c)]
    for x, gr in zip((a, b, c), grads):
        np.testing.assert_array_equal(gr, np.where(x == out, g, 0))                       # every input equal to the max gets the gradient
    prev = rng.standard_normal(out.shape).astype(np.float32)
    np.testing.assert_array_equal(G.eltwise_max_grad(g, a, out, target=prev, add=True), prev + grads[0])


def test_group_sparsity_in_label_cost_and_its_gradient():
    """cost.gsinlabel (PluginsSrc/src/extrautils.cu): the cost by its definition over explicit label groups, and
    the analytic gradient against central differences of that definition."""
    rng = np.random.default_rng(6)
    N, ch, P = 9, 3, 4
    labels = np.array([[2, 0, 2, 5, 0, 2, 7, 5, 2]], np.float32)
    acts = rng.standard_normal((ch * P, N)).astype(np.float32)

    def cost_by_definition(a):
        a = a.reshape(ch, P, N).astype(np.float64)
        total = 0.0
        for v in np.unique(labels):
            cols = np.flatnonzero(labels[0] == v)
            for c in range(ch):
                total += np.sqrt(len(cols) * (a[c][:, cols] ** 2).sum())
        return total

    cost, S, counts = G.group_sparsity_in_label_cost(labels, acts, ch)
    assert abs(cost - cost_by_definition(acts)) < 1e-9 * cost
    np.testing.assert_array_equal(counts, [4, 2, 4, 2, 2, 4, 1, 2, 4])
    g = G.group_sparsity_in_label_grad(labels, acts, ch, coeff=1.0)           # = -dCost/dActs
    h = 1e-3
    for idx in [(0, 0), (5, 3), (11, 6), (7, 8)]:
        up, dn = acts.copy(), acts.copy()
        up[idx] += h
        dn[idx] -= h
        num = (cost_by_definition(up) - cost_by_definition(dn)) / (up[idx].astype(np.float64) - dn[idx])
        assert abs(-g[idx] - num) < 1e-3 * max(1.0, abs(num)), (idx, g[idx], num)


def test_sparse_connectivity_equals_dense_conv_with_zero_embedded_filters():
    """randSparse (Kernel/src/conv/filter_acts.cu:259-413): group g's filters see channels conns[g*fc + c].  The
    oracle's route -- gather the channels, then the dense grouped contraction -- must equal an ordinary dense
    convolution whose filters are zero on every unconnected channel; its backward passes must be the adjoints."""
    rng = np.random.default_rng(7)
    N, C, S, fs, pad, stride, groups, fc, fpg = 4, 6, 7, 3, 1, 1, 3, 4, 2      # overSample = 3*4/6 = 2
    F = groups * fpg
    M = 1 + int(np.ceil((2 * pad + S - fs) / float(stride)))
    conns = [int(x) for _ in range(groups * fc // C) for x in rng.permutation(C)]
    x = rng.standard_normal((C * S * S, N)).astype(np.float32)
    W = rng.standard_normal((fc * fs * fs, F)).astype(np.float32)
    xg = G.gather_channels(x, conns, C, N)
    sparse = G.filter_acts(xg, W, N, groups * fc, S, S, F, fs, -pad, stride, M, M, groups)
    Wfull = np.zeros((C, fs * fs, F), np.float32)
    for g in range(groups):
        for c in range(fc):
            Wfull[conns[g * fc + c], :, g * fpg:(g + 1) * fpg] += W.reshape(fc, fs * fs, F)[c, :, g * fpg:(g + 1) * fpg]
    dense = G.filter_acts(x, Wfull.reshape(C * fs * fs, F), N, C, S, S, F, fs, -pad, stride, M, M, 1)
    np.testing.assert_allclose(sparse, dense, rtol=1e-5, atol=1e-5)
    # backward data: gradient of the gathered tensor scattered back == dense backward with the embedded filters
    gout = rng.standard_normal(sparse.shape).astype(np.float32)
    dxg = G.img_acts(gout, W, N, groups * fc, S, S, F, fs, -pad, stride, M, M, groups)
    dx = G.scatter_channels(dxg, conns, C, N)
    dx_dense = G.img_acts(gout, Wfull.reshape(C * fs * fs, F), N, C, S, S, F, fs, -pad, stride, M, M, 1)
    np.testing.assert_allclose(dx, dx_dense, rtol=1e-5, atol=1e-5)
    # accumulate form of the scatter
    prev = rng.standard_normal(x.shape).astype(np.float32)
    np.testing.assert_allclose(G.scatter_channels(dxg, conns, C, N, target=prev, scale_targets=1.0), prev + dx, rtol=1e-6, atol=1e-6)
