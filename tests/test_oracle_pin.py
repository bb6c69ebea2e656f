"""Pins the CPU oracle (oracle/golden.c) to the only reference-authored expectations for
the conv path: the three MATLAB generators Kernel/unittest/{filter_acts,img_acts,
weight_acts}/DataGen.m, whose expected values are convn() calls (no vectors are stored in
the reference).  They are re-created here with scipy.signal on the generators' sizes,
then widened (padding, stride, groups, local) against the same scipy construction."""
import numpy as np
import pytest
from scipy.signal import convolve2d, correlate2d

from oracle import golden as G


def _modules(img, fs, pad, stride):
    return 1 + int(np.ceil((2 * pad + img - fs) / float(stride)))   # Python/layer.py:910


def _ref_fwd(img, W, pad, stride, groups):
    # img (C,H,W,N), W (fc,fs,fs,F) -> (F,M,M,N) via padded 'valid' cross-correlation
    C, H, _, N = img.shape
    fc, fs, _, F = W.shape
    M = _modules(H, fs, pad, stride)
    need = (M - 1) * stride + fs
    extra = max(0, need - (H + 2 * pad))
    p = np.pad(img.astype(np.float64), ((0, 0), (pad, pad + extra), (pad, pad + extra), (0, 0)))
    out = np.zeros((F, M, M, N))
    fpg = F // groups
    for f in range(F):
        g = f // fpg
        for n in range(N):
            acc = 0
            for c in range(fc):
                acc = acc + correlate2d(p[g * fc + c, :, :, n], W[c, :, :, f].astype(np.float64), mode="valid")
            out[f, :, :, n] = acc[::stride, ::stride][:M, :M]
    return out


# -- Kernel/unittest/filter_acts/DataGen.m:3-8,50 : 10x10, 2 colours, 10 imgs, 5x5, 4 filters
def test_datagen_filter_acts():
    rng = np.random.default_rng(0)
    N, C, H, fs, F = 10, 2, 10, 5, 4
    img = rng.standard_normal((C, H, H, N)).astype(np.float32)
    W = rng.standard_normal((C, fs, fs, F)).astype(np.float32)
    got = G.filter_acts(img.reshape(-1, N), W.reshape(-1, F), N, C, H, H, F, fs, 0, 1, 6, 6)
    exp = _ref_fwd(img, W, 0, 1, 1)
    np.testing.assert_allclose(got.reshape(exp.shape), exp, rtol=1e-5, atol=1e-5)


# -- Kernel/unittest/img_acts/DataGen.m:7,52 : 30 filters, full convn(act, filter)
def test_datagen_img_acts():
    rng = np.random.default_rng(1)
    N, C, H, fs, F, M = 10, 2, 10, 5, 30, 6
    hid = rng.standard_normal((F, M, M, N)).astype(np.float32)
    W = rng.standard_normal((C, fs, fs, F)).astype(np.float32)
    got = G.img_acts(hid.reshape(-1, N), W.reshape(-1, F), N, C, H, H, F, fs, 0, 1, M, M)
    exp = np.zeros((C, H, H, N))
    for c in range(C):
        for n in range(N):
            for f in range(F):
                exp[c, :, :, n] += convolve2d(hid[f, :, :, n].astype(np.float64), W[c, :, :, f].astype(np.float64), mode="full")
    np.testing.assert_allclose(got.reshape(exp.shape), exp, rtol=1e-5, atol=1e-4)


# -- Kernel/unittest/weight_acts/DataGen.m:7,50 + test.cu:62-66 : 2 filters, partialSum=1 then sum over modules
def test_datagen_weight_acts():
    rng = np.random.default_rng(2)
    N, C, H, fs, F, M = 10, 2, 10, 5, 2, 6
    img = rng.standard_normal((C, H, H, N)).astype(np.float32)
    hid = rng.standard_normal((F, M, M, N)).astype(np.float32)
    part = G.weight_acts(img.reshape(-1, N), hid.reshape(-1, N), N, C, H, H, F, fs, 0, 1, M, M, partial_sum=1)
    assert part.shape == (M * M * C * fs * fs, F)
    got = part.reshape(M * M, C * fs * fs, F).sum(axis=0)
    full = G.weight_acts(img.reshape(-1, N), hid.reshape(-1, N), N, C, H, H, F, fs, 0, 1, M, M, partial_sum=0)
    exp = np.zeros((C, fs, fs, F))
    for c in range(C):
        for f in range(F):
            for n in range(N):
                exp[c, :, :, f] += correlate2d(img[c, :, :, n].astype(np.float64), hid[f, :, :, n].astype(np.float64), mode="valid")
    np.testing.assert_allclose(got.reshape(exp.shape), exp, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(full.reshape(exp.shape), exp, rtol=1e-4, atol=1e-4)


CASES = [  # N, C, H, fs, F, pad, stride, groups
    (4, 3, 9, 5, 6, 2, 1, 1),
    (3, 8, 8, 3, 8, 1, 1, 2),
    (5, 3, 11, 4, 4, 0, 3, 1),
    (2, 4, 7, 3, 4, 1, 2, 1),
]


@pytest.mark.parametrize("N,C,H,fs,F,pad,stride,groups", CASES)
def test_conv_triplet_consistency(N, C, H, fs, F, pad, stride, groups):
    """fwd against padded/strided scipy; bwd-data and bwd-weights as the exact adjoints of fwd."""
    rng = np.random.default_rng(3)
    fc = C // groups
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C, H, H, N)).astype(np.float32)
    W = rng.standard_normal((fc, fs, fs, F)).astype(np.float32)
    hid = rng.standard_normal((F, M, M, N)).astype(np.float32)
    out = G.filter_acts(img.reshape(-1, N), W.reshape(-1, F), N, C, H, H, F, fs, -pad, stride, M, M, groups)
    np.testing.assert_allclose(out.reshape(F, M, M, N), _ref_fwd(img, W, pad, stride, groups), rtol=1e-4, atol=1e-4)
    # <hid, fwd(img)> == <bwd_data(hid), img> == <bwd_weights(img,hid), W>
    lhs = float((hid.reshape(-1, N).astype(np.float64) * out).sum())
    din = G.img_acts(hid.reshape(-1, N), W.reshape(-1, F), N, C, H, H, F, fs, -pad, stride, M, M, groups)
    dw = G.weight_acts(img.reshape(-1, N), hid.reshape(-1, N), N, C, H, H, F, fs, -pad, stride, M, M, groups)
    assert abs(float((din.astype(np.float64) * img.reshape(-1, N)).sum()) - lhs) < 1e-3 * max(1, abs(lhs))
    assert abs(float((dw.astype(np.float64) * W.reshape(-1, F)).sum()) - lhs) < 1e-3 * max(1, abs(lhs))


def test_local_equals_conv_with_tiled_filters():
    """Unshared layer with the same filter bank at every module == conv (filter_acts.cu:100-102)."""
    rng = np.random.default_rng(4)
    N, C, H, fs, F, pad, stride = 3, 4, 6, 3, 4, 1, 1
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    W = rng.standard_normal((C * fs * fs, F)).astype(np.float32)
    Wl = np.tile(W, (M * M, 1))
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    a = G.filter_acts(img, W, N, C, H, H, F, fs, -pad, stride, M, M)
    b = G.filter_acts(img, Wl, N, C, H, H, F, fs, -pad, stride, M, M, conv=False)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    a = G.img_acts(hid, W, N, C, H, H, F, fs, -pad, stride, M, M)
    b = G.img_acts(hid, Wl, N, C, H, H, F, fs, -pad, stride, M, M, conv=False)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    a = G.weight_acts(img, hid, N, C, H, H, F, fs, -pad, stride, M, M, partial_sum=1)
    b = G.weight_acts(img, hid, N, C, H, H, F, fs, -pad, stride, M, M, conv=False)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)


def test_pool_and_undo_small():
    rng = np.random.default_rng(5)
    N, C, S = 3, 2, 7
    x = rng.standard_normal((C, S, S, N)).astype(np.float32)
    for kind, size, start, stride in (("max", 3, 0, 2), ("avg", 3, 0, 2), ("avg", 4, -2, 4), ("max", 2, 1, 2)):
        outX = int(np.ceil((S - start - size) / float(stride))) + 1      # Python/layer.py:1063
        y = G.pool(x.reshape(-1, N), N, C, S, size, start, stride, outX, kind).reshape(C, outX, outX, N)
        for oy in range(outX):
            for ox in range(outX):
                y0, x0 = max(0, start + oy * stride), max(0, start + ox * stride)
                y1, x1 = min(S, start + oy * stride + size), min(S, start + ox * stride + size)
                win = x[:, y0:y1, x0:x1, :]
                exp = win.max(axis=(1, 2)) if kind == "max" else win.sum(axis=(1, 2)) / ((y1 - y0) * (x1 - x0))
                np.testing.assert_allclose(y[:, oy, ox, :], exp, rtol=1e-6, atol=1e-6)
        g = rng.standard_normal(y.shape).astype(np.float32)
        if kind == "max":
            d = G.max_undo(x.reshape(-1, N), g.reshape(-1, N), y.reshape(-1, N), N, C, S, size, start, stride, outX)
        else:
            d = G.avg_undo(g.reshape(-1, N), N, C, S, size, start, stride, outX)
        d = d.reshape(C, S, S, N)
        exp = np.zeros_like(d, dtype=np.float64)
        for oy in range(outX):
            for ox in range(outX):
                y0, x0 = max(0, start + oy * stride), max(0, start + ox * stride)
                y1, x1 = min(S, start + oy * stride + size), min(S, start + ox * stride + size)
                if kind == "max":
                    exp[:, y0:y1, x0:x1, :] += (x[:, y0:y1, x0:x1, :] == y[:, oy:oy + 1, ox:ox + 1, :]) * g[:, oy:oy + 1, ox:ox + 1, :]
                else:
                    exp[:, y0:y1, x0:x1, :] += g[:, oy:oy + 1, ox:ox + 1, :] / ((y1 - y0) * (x1 - x0))
        np.testing.assert_allclose(d, exp, rtol=1e-5, atol=1e-6)


def test_norm_fwd_formulas():
    rng = np.random.default_rng(6)
    N, C, S, size = 2, 16, 5, 3
    x = rng.standard_normal((C, S, S, N)).astype(np.float32)
    scale, pw = 0.001 / size ** 2, 0.75
    y, d = G.rnorm(x.reshape(-1, N), N, C, S, size, scale, pw)
    p = np.pad(x.astype(np.float64) ** 2, ((0, 0), (1, 1), (1, 1), (0, 0)))
    s = sum(p[:, dy:dy + S, dx:dx + S, :] for dy in range(3) for dx in range(3))
    np.testing.assert_allclose(d.reshape(x.shape), 1 + scale * s, rtol=1e-6)
    np.testing.assert_allclose(y.reshape(x.shape), x * (1 + scale * s) ** -pw, rtol=1e-5)
    size = 5
    scale = 0.0001 / size
    y, d = G.cmrnorm(x.reshape(-1, N), N, C, S, size, scale, pw, False, 2.0)
    p = np.pad(x.astype(np.float64) ** 2, ((2, 2), (0, 0), (0, 0), (0, 0)))
    s = sum(p[i:i + C] for i in range(5))
    np.testing.assert_allclose(d.reshape(x.shape), 2.0 + scale * s, rtol=1e-6)
    np.testing.assert_allclose(y.reshape(x.shape), x * (2.0 + scale * s) ** -pw, rtol=1e-5)


# -- Kernel/unittest/groupSparsity/DataGen.m:1-46 + test.cu:74-80 : 16x16x3, 12 cases, labels 1..10; cost and gradient in closed form
def test_datagen_group_sparsity():
    """The fourth (and last) generator the reference ships, re-created in NumPy (tests/golden/make_golden.py, independent of
    oracle/): the plugin's cost.gsinlabel arithmetic (PluginsSrc/src/extrautils.cu).  The reference's kernel adds 2.5e-7 to the
    root before dividing (extrautils.cu:190), the generator does not: with roots of ~10 that is 3e-8 relative."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datagen_group_sparsity.npz"))
    N, ch, P = (int(v) for v in z["geom"])
    assert z["acts"].shape == (ch * P, N)
    cost, S, counts = G.group_sparsity_in_label_cost(z["labels"], z["acts"], ch)
    assert abs(cost - float(z["expected_cost"])) < 1e-6 * cost
    grad = G.group_sparsity_in_label_grad(z["labels"], z["acts"], ch, coeff=-1.0)     # the layer's bprop is -coeff * this
    np.testing.assert_allclose(grad, z["expected_grad"], rtol=1e-5, atol=1e-7)
