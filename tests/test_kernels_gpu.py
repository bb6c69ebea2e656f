"""GPU parity: every C-ABI entry point of libmckernels.so against the CPU oracle on the same
seeded inputs.  Tolerance: fp32 contractions rel 1e-4 (north_star); pooling, max-undo routing,
relu, dropout and the SGD update are compared exactly."""
import numpy as np
import pytest

from oracle import golden as G
from gpu_util import dev, host, keep, nans, ptr, relerr, zeros

pytestmark = pytest.mark.gpu
TOL = 1e-4


def mc():
    import mckernels
    return mckernels


def _modules(img, fs, pad, stride):
    return 1 + int(np.ceil((2 * pad + img - fs) / float(stride)))


# N, C, H, fs, F, pad, stride, groups
CONV_CASES = [
    (128, 3, 32, 5, 32, 2, 1, 1),     # layers-80sec conv1
    (128, 32, 16, 5, 32, 2, 1, 1),    # layers-80sec conv2
    (128, 32, 8, 5, 64, 2, 1, 1),     # layers-80sec conv3
    (16, 3, 32, 5, 32, 2, 1, 1),      # 16-case tail minibatch (10000 % 128)
    (32, 96, 8, 5, 192, 2, 1, 1),     # NIN conv2-like channel/filter counts
    (64, 192, 8, 3, 192, 1, 1, 1),    # NIN conv3
    (8, 3, 23, 11, 96, 0, 4, 1),      # imagenet conv1-like: 11x11 stride 4
    (12, 8, 9, 3, 16, 1, 2, 2),       # groups + stride 2
    (5, 3, 10, 5, 4, 0, 1, 1),        # odd batch: scalar path; DataGen.m sizes
    (6, 4, 7, 3, 10, 1, 1, 1),        # filters not a multiple of 4 (cccp6-like F=10)
    (128, 16, 6, 3, 32, 1, 3, 1),     # stride 3 with padding
]


@pytest.mark.parametrize("N,C,H,fs,F,pad,stride,groups", CONV_CASES)
def test_conv_triplet(N, C, H, fs, F, pad, stride, groups):
    m = mc()
    rng = np.random.default_rng(11)
    M = _modules(H, fs, pad, stride)
    fc = C // groups
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    W = rng.standard_normal((fc * fs * fs, F)).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    dimg, dW, dhid = dev(img), dev(W), dev(hid)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)

    out = nans(F * M * M, N)
    m.call("mc_conv_filter_acts", ptr(dimg), ptr(dW), ptr(out), *geo, 0.0, 1.0, None)
    exp = G.filter_acts(img, W, *geo)
    assert relerr(host(out), exp) < TOL

    # accumulate form: target = 0.5*target + 2*conv
    prev = rng.standard_normal(exp.shape).astype(np.float32)
    out2 = dev(prev)
    m.call("mc_conv_filter_acts", ptr(dimg), ptr(dW), ptr(out2), *geo, 0.5, 2.0, None)
    assert relerr(host(out2), 0.5 * prev + 2.0 * exp) < TOL

    din = nans(C * H * H, N)
    m.call("mc_conv_img_acts", ptr(dhid), ptr(dW), ptr(din), *geo, 0.0, 1.0, None)
    assert relerr(host(din), G.img_acts(hid, W, *geo)) < TOL

    for ps in sorted({0, 1, M} | ({M * M // 2} if (M * M) % 2 == 0 else set())):
        if ps and (M * M) % ps:
            continue
        expw = G.weight_acts(img, hid, *geo, partial_sum=ps)
        dw = nans(*expw.shape)
        m.call("mc_conv_weight_acts", ptr(dimg), ptr(dhid), ptr(dw), *geo, ps, 0.0, 1.0, None)
        assert relerr(host(dw), expw) < TOL, ps


BAND_CASES = [
    # N, C, H, F, pad, modules (stride 1, 5x5): the shapes conv_band.hip serves, plus ragged edges
    (128, 3, 32, 32, 2, 32),     # layers-80sec conv1
    (128, 32, 16, 32, 2, 16),    # layers-80sec conv2
    (128, 3, 32, 192, 2, 32),    # NIN conv1
    (64, 3, 24, 64, 2, 24),      # cifar-cropped first layer, 64 filters
    (32, 16, 21, 32, 2, 21),     # odd map: the last 2x2 block hangs over the edge; two channel chunks
    (96, 8, 20, 32, 0, 16),      # no padding, one chunk, three image blocks
    (64, 3, 30, 32, 4, 34),      # more padding than the filter needs
]


@pytest.mark.parametrize("N,C,H,F,pad,M", BAND_CASES)
def test_conv_band_kernel_gives_the_engines_bits(N, C, H, F, pad, M):
    """conv_band.hip against the oracle, and bit for bit against the general engine (same MFMA, same k order, same epilogue):
    plain, with bias + relu, and with the relu bit mask."""
    import torch
    m = mc()
    rng = np.random.default_rng(21)
    fs = 5
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    W = rng.standard_normal((C * fs * fs, F)).astype(np.float32)
    b = rng.standard_normal((F,)).astype(np.float32)
    dimg, dW, db = dev(img), dev(W), dev(b)
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
    exp = G.filter_acts(img, W, *geo)
    words = m.lib().mc_relu_bits_words(F, M * M * N)
    got = {}
    try:
        for mode in (0, 1):
            m.call("mc_set_conv_band", mode)
            plain, fused, rb = nans(F * M * M, N), nans(F * M * M, N), nans(F * M * M, N)
            bits = torch.zeros(words, dtype=torch.int32, device="cuda")
            m.call("mc_conv_filter_acts", ptr(dimg), ptr(dW), ptr(plain), *geo, 0.0, 0.5, None)
            m.call("mc_conv_filter_acts_fused", ptr(dimg), ptr(dW), ptr(fused), ptr(db), m.NEURON_RELU, *geo, 0.0, 1.0, None)
            m.call("mc_conv_filter_acts_relu_bits", ptr(dimg), ptr(dW), ptr(rb), ptr(db), ptr(bits), *geo[:-1], None)
            got[mode] = (host(plain), host(fused), host(rb), bits.cpu().numpy())
    finally:
        m.call("mc_set_conv_band", -1)
    assert relerr(got[1][0], 0.5 * exp) < TOL
    assert relerr(got[1][1], np.maximum(exp + b.repeat(M * M)[:, None], 0)) < TOL
    for x, y, what in zip(got[0], got[1], ("plain", "bias + relu", "relu + bits: values", "relu + bits: mask")):
        assert np.array_equal(x, y), what


def test_conv_fused_bias_relu_is_bitwise_unfused():
    m = mc()
    rng = np.random.default_rng(12)
    N, C, H, fs, F, pad, stride = 128, 3, 32, 5, 32, 2, 1
    M = _modules(H, fs, pad, stride)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, 1)
    img, W = dev(rng.standard_normal((C * H * H, N))), dev(rng.standard_normal((C * fs * fs, F)))
    b = dev(rng.standard_normal((F,)))
    a = nans(F * M * M, N)
    m.call("mc_conv_filter_acts", ptr(img), ptr(W), ptr(a), *geo, 0.0, 1.0, None)
    m.call("mc_add_row_vector", ptr(a), ptr(b), F, M * M * N, 1.0, None)
    m.call("mc_neuron_activate", m.NEURONS["relu"], 0.0, 0.0, ptr(a), ptr(a), a.numel(), None)
    f = nans(F * M * M, N)
    m.call("mc_conv_filter_acts_fused", ptr(img), ptr(W), ptr(f), ptr(b), m.NEURON_RELU, *geo, 0.0, 1.0, None)
    assert np.array_equal(host(a), host(f))


LOCAL_CASES = [(128, 64, 8, 3, 32, 1, 1, 1), (16, 32, 8, 3, 32, 1, 1, 1), (6, 16, 7, 5, 32, 2, 3, 2), (5, 4, 6, 3, 6, 1, 1, 1)]


@pytest.mark.parametrize("N,C,H,fs,F,pad,stride,groups", LOCAL_CASES)
def test_local_triplet(N, C, H, fs, F, pad, stride, groups):
    m = mc()
    rng = np.random.default_rng(13)
    M = _modules(H, fs, pad, stride)
    fc = C // groups
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    W = rng.standard_normal((M * M * fc * fs * fs, F)).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    dimg, dW, dhid = dev(img), dev(W), dev(hid)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)
    out = nans(F * M * M, N)
    m.call("mc_local_filter_acts", ptr(dimg), ptr(dW), ptr(out), *geo, 0.0, 1.0, None)
    assert relerr(host(out), G.filter_acts(img, W, *geo, conv=False)) < TOL
    din = nans(C * H * H, N)
    m.call("mc_local_img_acts", ptr(dhid), ptr(dW), ptr(din), *geo, 0.0, 1.0, None)
    assert relerr(host(din), G.img_acts(hid, W, *geo, conv=False)) < TOL
    expw = G.weight_acts(img, hid, *geo, conv=False)
    prev = rng.standard_normal(expw.shape).astype(np.float32)
    dw = dev(prev)   # the reference accumulates straight into weightsInc with beta = momentum (layer.cu:578-591)
    m.call("mc_local_weight_acts", ptr(dimg), ptr(dhid), ptr(dw), *geo, 0.9, 0.01, None)
    assert relerr(host(dw), 0.9 * prev + 0.01 * expw) < TOL


GEMM_CASES = [  # m, n, k
    (160, 1024 * 16, 192), (96, 4096, 160), (10, 8192, 192), (64, 128, 1024), (10, 128, 64),
    (192, 192, 4096), (37, 53, 29), (2048, 128, 2560),
    # vector path (every extent a multiple of 4) with ragged tiles and a K tail inside the last 16-k step: the LDS-DMA producers'
    # out-of-range lanes (rows / columns past the edge, k past the end) must land as zeros, partial waves of weight rows stay off
    (200, 640, 100), (72, 336, 52), (132, 260, 36),
]


@pytest.mark.parametrize("m_,n_,k_", GEMM_CASES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm(m_, n_, k_, ta, tb):
    m = mc()
    rng = np.random.default_rng(14)
    A = rng.standard_normal((k_, m_) if ta else (m_, k_)).astype(np.float32)
    B = rng.standard_normal((n_, k_) if tb else (k_, n_)).astype(np.float32)
    exp = G.matmul(A.T if ta else A, B.T if tb else B)
    dA, dB = dev(A), dev(B)
    C = nans(m_, n_)
    ws_bytes = m.lib().mc_gemm_workspace_bytes(m_, n_, k_)
    ws = zeros(max(1, ws_bytes // 4))
    m.call("mc_gemm", ptr(dA), ptr(dB), ptr(C), m_, n_, k_, ta, tb, A.shape[1], B.shape[1], n_, 0.0, 1.0, None,
           m.NEURON_NONE, ptr(ws), ws_bytes, None)
    assert relerr(host(C), exp) < TOL
    # beta/alpha + bias + relu epilogue, no workspace
    prev = rng.standard_normal((m_, n_)).astype(np.float32)
    bias = rng.standard_normal((m_,)).astype(np.float32)
    C2, db = dev(prev), dev(bias)
    m.call("mc_gemm", ptr(dA), ptr(dB), ptr(C2), m_, n_, k_, ta, tb, A.shape[1], B.shape[1], n_, 0.25, 0.5, ptr(db),
           m.NEURON_RELU, None, 0, None)
    e2 = np.maximum(0.25 * prev + 0.5 * exp.astype(np.float64) + bias[:, None], 0)
    assert relerr(host(C2), e2) < TOL


POOL_CASES = [  # N, C, S, size, start, stride
    (128, 96, 32, 3, 0, 2), (128, 10, 8, 8, 0, 1), (16, 32, 16, 3, 0, 2), (128, 16, 8, 2, 1, 2),
    (6, 16, 6, 4, -2, 4), (5, 3, 7, 3, 0, 2),
    (128, 96, 55, 2, 1, 2), (128, 256, 27, 2, 1, 2),   # the imagenet net's pools (2x2, stride 2, start 1) at full size
    # size 4 / stride 2: an odd start puts three windows over a 2-pixel block per axis (the stride-2 undo kernel holds two:
    # such shapes must take the general kernel), an even start two
    (16, 8, 9, 4, -1, 2), (16, 8, 9, 4, 1, 2), (16, 8, 10, 4, 0, 2), (16, 8, 9, 4, -2, 2),
]


@pytest.mark.parametrize("N,C,S,size,start,stride", POOL_CASES)
@pytest.mark.parametrize("kind", ["max", "avg"])
def test_pool_and_undo(N, C, S, size, start, stride, kind):
    m = mc()
    rng = np.random.default_rng(15)
    outX = int(np.ceil((S - start - size) / float(stride))) + 1
    x = rng.standard_normal((C * S * S, N)).astype(np.float32)
    if kind == "max":   # ReLU-like zero plateaus exercise the equality (tie) routing
        x = np.maximum(x, 0)
    dx = dev(x)
    y = nans(C * outX * outX, N)
    m.call("mc_conv_local_pool", ptr(dx), ptr(y), N, C, S, size, start, stride, outX, m.POOL_MAX if kind == "max" else m.POOL_AVG, None)
    ey = G.pool(x, N, C, S, size, start, stride, outX, kind)
    if kind == "max":
        assert np.array_equal(host(y), ey)        # bit exact
    else:                                         # large windows are summed in slices: order differs from the oracle's
        np.testing.assert_allclose(host(y), ey, rtol=1e-6, atol=1e-6)
    g = rng.standard_normal(ey.shape).astype(np.float32)
    dg = dev(g)
    prev = rng.standard_normal(x.shape).astype(np.float32)
    for sT, sO in ((0.0, 1.0), (1.0, 1.0)):
        t = nans(*x.shape) if sT == 0 else dev(prev)
        if kind == "max":
            m.call("mc_conv_local_max_undo", ptr(dx), ptr(dg), ptr(y), ptr(t), N, C, S, size, start, stride, outX, sT, sO, None)
            e = G.max_undo(x, g, ey, N, C, S, size, start, stride, outX, target=prev.copy(), scale_targets=sT, scale_output=sO)
            assert np.array_equal(host(t), e)     # routing bit exact
        else:
            m.call("mc_conv_local_avg_undo", ptr(dg), ptr(t), N, C, S, size, start, stride, outX, sT, sO, None)
            e = G.avg_undo(g, N, C, S, size, start, stride, outX, target=prev.copy(), scale_targets=sT, scale_output=sO)
            np.testing.assert_allclose(host(t), e, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,C,S,size", [(128, 32, 16, 3), (16, 32, 8, 3), (6, 16, 5, 5), (5, 16, 6, 9), (128, 32, 8, 3), (12, 8, 7, 3)])
def test_rnorm_and_undo(N, C, S, size):
    m = mc()
    rng = np.random.default_rng(16)
    scale, pw = 0.001 / size ** 2, 0.75
    x = rng.standard_normal((C * S * S, N)).astype(np.float32) * 3
    dx, den, y = dev(x), nans(*x.shape), nans(*x.shape)
    m.call("mc_conv_response_norm", ptr(dx), ptr(den), ptr(y), N, C, S, size, scale, pw, None)
    ey, eden = G.rnorm(x, N, C, S, size, scale, pw)
    assert relerr(host(den), eden) < 1e-6 and relerr(host(y), ey) < TOL
    g = rng.standard_normal(x.shape).astype(np.float32)
    t = nans(*x.shape)
    acts = dev(ey)
    m.call("mc_conv_response_norm_undo", ptr(keep(dev(g))), ptr(keep(dev(eden))), ptr(dx), ptr(acts), ptr(t), N, C, S, size, scale, pw, 0.0, 1.0, None)
    eacts = ey.copy()
    e = G.rnorm_undo(g, eden, x, eacts, N, C, S, size, scale, pw)
    assert relerr(host(t), e) < TOL
    assert relerr(host(acts), eacts) < TOL          # acts is clobbered the same way (conv_util.cu:2236)


@pytest.mark.parametrize("N,C,S,size,blocked", [(128, 96, 6, 5, 0), (16, 32, 4, 5, 0), (6, 16, 3, 4, 1), (5, 16, 3, 3, 0),
                                                 (128, 96, 55, 5, 0), (128, 256, 27, 5, 0)])   # imagenet rnorm1 / rnorm2 at full size
def test_cmrnorm_and_undo(N, C, S, size, blocked):
    m = mc()
    rng = np.random.default_rng(17)
    scale, pw, k = 0.0001 / size, 0.75, 2.0
    x = rng.standard_normal((C * S * S, N)).astype(np.float32) * 5
    dx, den, y = dev(x), nans(*x.shape), nans(*x.shape)
    m.call("mc_conv_response_norm_cross_map", ptr(dx), ptr(den), ptr(y), N, C, S, size, scale, pw, blocked, k, None)
    ey, eden = G.cmrnorm(x, N, C, S, size, scale, pw, bool(blocked), k)
    assert relerr(host(den), eden) < 1e-6 and relerr(host(y), ey) < TOL
    g = rng.standard_normal(x.shape).astype(np.float32)
    prev = rng.standard_normal(x.shape).astype(np.float32)
    t = dev(prev)
    acts = dev(ey)
    m.call("mc_conv_response_norm_cross_map_undo", ptr(keep(dev(g))), ptr(keep(dev(eden))), ptr(dx), ptr(acts), ptr(t), N, C, S, size, scale, pw, blocked, 1.0, 1.0, None)
    e = G.cmrnorm_undo(g, eden, x, ey.copy(), N, C, S, size, scale, pw, bool(blocked), target=prev.copy(), scale_targets=1.0, scale_output=1.0)
    assert relerr(host(t), e) < TOL


NEURON_CASES = [("ident", {}), ("relu", {}), ("brelu", {"a": 0.7}), ("abs", {}), ("logistic", {}),
                ("tanh", {"a": 1.79, "b": -0.66}), ("softrelu", {}), ("square", {}), ("sqrt", {}), ("linear", {"a": 3.0, "b": 2.2})]


@pytest.mark.parametrize("kind,params", NEURON_CASES)
@pytest.mark.parametrize("n", [4096 * 3, 1001])
def test_neurons(kind, params, n):
    m = mc()
    rng = np.random.default_rng(18)
    x = rng.standard_normal(n).astype(np.float32) * 3
    if kind == "sqrt":
        x = np.abs(x) + 0.1
    a, b = params.get("a", 0.0), params.get("b", 0.0)
    dx, y = dev(x), nans(n)
    m.call("mc_neuron_activate", m.NEURONS[kind], a, b, ptr(dx), ptr(y), n, None)
    ey = G.neuron_fwd(kind, params, x)
    exact = kind in ("ident", "relu", "brelu", "abs", "square")   # linear: a*x+b is one FMA on the GPU
    if exact:
        assert np.array_equal(host(y), ey)
    else:
        np.testing.assert_allclose(host(y), ey, rtol=2e-6, atol=1e-6)
    g = rng.standard_normal(n).astype(np.float32)
    uses_input = kind in ("abs", "square")
    eg = G.neuron_bwd(kind, params, g, ey, x)
    for add in (0, 1):
        prev = rng.standard_normal(n).astype(np.float32)
        t = dev(prev)
        m.call("mc_neuron_input_grad", m.NEURONS[kind], a, b, ptr(keep(dev(g))), ptr(dx if uses_input else keep(dev(ey))), ptr(t), add, n, None)
        np.testing.assert_allclose(host(t), eg + (prev if add else 0), rtol=1e-5, atol=1e-5)
    # in-place activation (actsTarget >= 0 layers run the neuron in place, layer.cu:97)
    m.call("mc_neuron_activate", m.NEURONS[kind], a, b, ptr(dx), ptr(dx), n, None)
    np.testing.assert_allclose(host(dx), ey, rtol=2e-6, atol=1e-6)


def test_dropout_and_rng():
    m = mc()
    n = 128 * 96 * 16 * 16
    u = zeros(n)
    m.call("mc_rand_uniform", ptr(u), n, 99, 0, None)
    hu = host(u)
    assert hu.min() >= 0.0 and hu.max() < 1.0 and abs(hu.mean() - 0.5) < 2e-3 and abs((hu < 0.5).mean() - 0.5) < 2e-3
    # counter based: a sub-range regenerated with its offset is the same stream
    v = zeros(1000)
    m.call("mc_rand_uniform", ptr(v), 1000, 99, 4096, None)
    assert np.array_equal(host(v), hu[4096:5096])
    w = zeros(1000)
    m.call("mc_rand_uniform", ptr(w), 1000, 100, 4096, None)
    assert not np.array_equal(host(w), host(v))
    rng = np.random.default_rng(19)
    x = rng.standard_normal(n).astype(np.float32)
    y = nans(n)
    m.call("mc_dropout_apply", ptr(keep(dev(x))), ptr(u), ptr(y), 0.5, 0, n, None)
    assert np.array_equal(host(y), G.neuron_fwd("dropout", {"prob": 0.5}, x, train=True, mask=hu))


@pytest.mark.parametrize("n", [128 * 96 * 16 * 16, 70 * 192 * 64, 1001])
def test_dropout_counter_equals_stored_mask(n):
    """mc_dropout_counter regenerates the uniforms of mc_rand_uniform's stream in the kernel: forward, backward and the
    accumulate form equal mc_rand_uniform + mc_dropout_apply bit for bit (incl. a tail that is not a multiple of 4)."""
    m = mc()
    rng = np.random.default_rng(20)
    x, g, prev = (dev(rng.standard_normal(n)) for _ in range(3))
    seed, off = 1234567, 4 * 99991
    u = zeros(n)
    m.call("mc_rand_uniform", ptr(u), n, seed, off, None)
    for src, add in ((x, 0), (g, 0), (g, 1)):
        a, b = (prev.clone(), prev.clone()) if add else (nans(n), nans(n))
        m.call("mc_dropout_apply", ptr(src), ptr(u), ptr(a), 0.5, add, n, None)
        m.call("mc_dropout_counter", ptr(src), ptr(b), 0.5, add, n, seed, off, None)
        assert np.array_equal(host(a), host(b))
    # in place, as a neuron layer with actsTarget >= 0 runs it
    c = x.clone()
    m.call("mc_dropout_counter", ptr(c), ptr(c), 0.5, 0, n, seed, off, None)
    e = nans(n)
    m.call("mc_dropout_apply", ptr(x), ptr(u), ptr(e), 0.5, 0, n, None)
    assert np.array_equal(host(c), host(e))
    kept = float((host(c) != 0).mean())
    assert abs(kept - 0.5) < 0.05


@pytest.mark.parametrize("N,C,S,size,start,stride", [(128, 96, 32, 3, 0, 2), (128, 16, 8, 2, 1, 2), (16, 8, 9, 4, -1, 2), (6, 16, 6, 4, -2, 4),
                                                     (5, 3, 7, 3, 0, 2), (12, 8, 9, 3, 0, 2)])
def test_pool_with_fused_dropout(N, C, S, size, start, stride):
    """NIN's pool -> dropout pairs as one forward and one backward pass: mc_conv_local_pool_dropout writes the pooled tensor, its
    dropped-out copy and the keep bytes; mc_conv_local_max_undo_dropout applies the keep bytes to the gradient as it routes it.
    Bit-identical to the separate kernels (pool, mc_dropout_counter forward, mc_dropout_counter on the gradient, max-undo with and
    without the relu gradient) on the 3x3 / 2x2 fast paths, the general path, and cases whose images are not a multiple of 4."""
    import torch
    m = mc()
    rng = np.random.default_rng(61)
    outX = int(np.ceil((S - start - size) / float(stride))) + 1
    x = np.maximum(rng.standard_normal((C * S * S, N)).astype(np.float32), 0)       # a relu's output: zero plateaus, ties
    n = C * outX * outX * N
    p, seed, off = 0.5, 987654321, 4 * 12345
    dx = dev(x)
    y0, d0 = nans(C * outX * outX, N), nans(C * outX * outX, N)
    m.call("mc_conv_local_pool", ptr(dx), ptr(y0), N, C, S, size, start, stride, outX, m.POOL_MAX, None)
    m.call("mc_dropout_counter", ptr(y0), ptr(d0), p, 0, n, seed, off, None)
    y1, d1 = nans(C * outX * outX, N), nans(C * outX * outX, N)
    keep = torch.full((int(m.lib().mc_dropout_keep_bytes(n)),), 255, dtype=torch.uint8, device="cuda")
    m.call("mc_conv_local_pool_dropout", ptr(dx), ptr(y1), ptr(d1), keep.data_ptr(), N, C, S, size, start, stride, outX, m.POOL_MAX, p, seed, off, None)
    assert np.array_equal(host(y1), host(y0)) and np.array_equal(host(d1), host(d0))
    u = zeros(n)
    m.call("mc_rand_uniform", ptr(u), n, seed, off, None)
    bits = (host(u) >= p).astype(np.uint8)
    bits = np.concatenate([bits, np.zeros((-n) % 4, np.uint8)]).reshape(-1, 4)
    assert np.array_equal(host(keep), (bits * np.array([1, 2, 4, 8], np.uint8)).sum(axis=1).astype(np.uint8))
    g = dev(rng.standard_normal((C * outX * outX, N)))
    gd = nans(C * outX * outX, N)
    m.call("mc_dropout_counter", ptr(g), ptr(gd), p, 0, n, seed, off, None)
    prev = rng.standard_normal(x.shape).astype(np.float32)
    for relu in (0, 1):
        for sT in ((0.0,) if relu else (0.0, 1.0)):
            a = nans(*x.shape) if sT == 0 else dev(prev)
            b = nans(*x.shape) if sT == 0 else dev(prev)
            m.call("mc_conv_local_max_undo_relu_grad" if relu else "mc_conv_local_max_undo", ptr(dx), ptr(gd), ptr(y0), ptr(a), N, C, S, size, start,
                   stride, outX, sT, 1.0, None)
            m.call("mc_conv_local_max_undo_dropout", ptr(dx), ptr(g), ptr(y1), keep.data_ptr(), ptr(b), N, C, S, size, start, stride, outX, sT, 1.0, relu, None)
            assert np.array_equal(host(a), host(b)), (relu, sT)


@pytest.mark.parametrize("N,C,S,size,start,stride", [(128, 32, 32, 3, 0, 2), (128, 16, 8, 2, 1, 2), (16, 8, 9, 4, -1, 2), (6, 16, 6, 4, -2, 4),
                                                     (5, 3, 7, 3, 0, 2), (12, 8, 9, 3, 0, 2)])
def test_pool_with_fused_relu(N, C, S, size, start, stride):
    """`neuron=relu` on a max-pool layer (layers-80sec pool1) as one forward and one backward pass: mc_conv_local_pool_relu stores
    max(0, pooled); mc_conv_local_max_undo_relu_out applies the relu's backward step to every window's gradient as it routes it.
    Bit-identical to the separate kernels (pool, relu in place, relu gradient, max-undo with and without the relu gradient of the
    layer below, plain and accumulating) on the fast paths, the general path, and image counts that are not a multiple of 4."""
    m = mc()
    rng = np.random.default_rng(62)
    outX = int(np.ceil((S - start - size) / float(stride))) + 1
    n = C * outX * outX * N
    x = rng.standard_normal((C * S * S, N)).astype(np.float32) - 0.8     # many windows whose maximum is negative
    x[rng.random(x.shape) < 0.2] = 0.0                                   # and exact zeros, which tie with a relu'd maximum of 0
    dx = dev(x)
    for pooler in (m.POOL_MAX, m.POOL_AVG):
        y0, y1 = nans(C * outX * outX, N), nans(C * outX * outX, N)
        m.call("mc_conv_local_pool", ptr(dx), ptr(y0), N, C, S, size, start, stride, outX, pooler, None)
        r0, r1 = nans(C * outX * outX, N), nans(C * outX * outX, N)
        m.call("mc_neuron_activate", 1, 0.0, 0.0, ptr(y0), ptr(r0), n, None)
        m.call("mc_conv_local_pool_relu", ptr(dx), ptr(y1), ptr(r1), N, C, S, size, start, stride, outX, pooler, None)     # a matrix of its own
        assert np.array_equal(host(y0), host(y1)) and np.array_equal(host(r0), host(r1)), pooler
        m.call("mc_conv_local_pool_relu", ptr(dx), ptr(y1), ptr(y1), N, C, S, size, start, stride, outX, pooler, None)     # in place
        assert np.array_equal(host(r0), host(y1)), pooler
        if pooler == m.POOL_MAX:
            ymax = y1
    assert (host(ymax) == 0).mean() > 0.05
    g = dev(rng.standard_normal((C * outX * outX, N)))
    gr = nans(C * outX * outX, N)
    m.call("mc_neuron_input_grad", 1, 0.0, 0.0, ptr(g), ptr(ymax), ptr(gr), 0, n, None)
    prev = rng.standard_normal(x.shape).astype(np.float32)
    for relu in (0, 1):
        for sT in ((0.0,) if relu else (0.0, 1.0)):
            a = nans(*x.shape) if sT == 0 else dev(prev)
            b = nans(*x.shape) if sT == 0 else dev(prev)
            m.call("mc_conv_local_max_undo_relu_grad" if relu else "mc_conv_local_max_undo", ptr(dx), ptr(gr), ptr(ymax), ptr(a), N, C, S, size, start,
                   stride, outX, sT, 1.0, None)
            m.call("mc_conv_local_max_undo_relu_out", ptr(dx), ptr(g), ptr(ymax), ptr(b), N, C, S, size, start, stride, outX, sT, 1.0, relu, None)
            assert np.array_equal(host(a), host(b)), (relu, sT)


def test_bias_and_reductions():
    m = mc()
    rng = np.random.default_rng(20)
    for rows, cols in ((192, 16 * 16 * 128), (10, 64 * 16), (7, 33), (256, 75 * 192), (33, 20)):
        a = rng.standard_normal((rows, cols)).astype(np.float32)
        b = rng.standard_normal(rows).astype(np.float32)
        da = dev(a)
        m.call("mc_add_row_vector", ptr(da), ptr(keep(dev(b))), rows, cols, 1.0, None)
        assert np.array_equal(host(da), a + b[:, None])
        prev = rng.standard_normal(rows).astype(np.float32)
        o = dev(prev)
        m.call("mc_sum_rows", ptr(keep(dev(a))), ptr(o), rows, cols, 0.5, 0.01, None)
        np.testing.assert_allclose(host(o), 0.5 * prev + 0.01 * a.astype(np.float64).sum(1), rtol=1e-5, atol=1e-5)
        o2 = nans(cols)
        m.call("mc_sum_cols", ptr(keep(dev(a))), ptr(o2), rows, cols, 0.0, 1.0, None)
        np.testing.assert_allclose(host(o2), a.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    import torch
    v = rng.standard_normal(128).astype(np.float32)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    m.call("mc_sum_all", ptr(keep(dev(v))), 128, out.data_ptr(), None)
    torch.cuda.synchronize()
    assert abs(out.item() - v.astype(np.float64).sum()) < 1e-9


@pytest.mark.parametrize("N,K", [(128, 10), (16, 10), (128, 100), (5, 7), (2500, 10), (128, 1000), (64, 64), (130, 100), (256, 1000)])
def test_softmax_logreg(N, K):
    # (64 classes and more, whole float4 groups of cases: the class-parallel kernels; (130, 100) stays with the one-lane family)
    m = mc()
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((K, N)) * 4).astype(np.float32)
    labels = rng.integers(0, K, size=(1, N)).astype(np.float32)
    dx, dl = dev(x), dev(labels)
    p = nans(K, N)
    m.call("mc_softmax", ptr(dx), ptr(p), N, K, None)
    ep = G.softmax(x)
    np.testing.assert_allclose(host(p), ep, rtol=1e-5, atol=1e-7)
    hp = host(p)
    lp, cp = nans(N), nans(N)
    m.call("mc_compute_logreg_cost", ptr(dl), ptr(p), ptr(lp), ptr(cp), N, K, None)
    ec = G.logreg_cost(labels, hp)
    assert abs(-host(lp).astype(np.float64).sum() - ec[0]) < 1e-4 * max(1.0, abs(ec[0]))
    assert abs((N - host(cp).astype(np.float64).sum()) - ec[1]) < 1e-6
    # the fused form (values + both sums in one launch) equals the three separate calls bit for bit
    import torch
    lp2, cp2 = nans(N), nans(N)
    sums = torch.full((2,), float("nan"), dtype=torch.float64, device="cuda")
    sep = torch.full((2,), float("nan"), dtype=torch.float64, device="cuda")
    m.call("mc_compute_logreg_cost_sums", ptr(dl), ptr(p), ptr(lp2), ptr(cp2), ptr(sums), N, K, None)
    m.call("mc_sum_all", ptr(lp), N, sep.data_ptr(), None)
    m.call("mc_sum_all", ptr(cp), N, sep.data_ptr() + 8, None)
    assert np.array_equal(host(lp2), host(lp)) and np.array_equal(host(cp2), host(cp)) and np.array_equal(host(sums), host(sep))
    g = nans(K, N)
    m.call("mc_compute_logreg_softmax_grad", ptr(dl), ptr(p), ptr(g), 0, 0.1, N, K, None)
    np.testing.assert_allclose(host(g), G.logreg_softmax_grad(labels, hp, 0.1), rtol=1e-6, atol=1e-7)
    g2 = nans(K, N)
    m.call("mc_compute_logreg_grad", ptr(dl), ptr(p), ptr(g2), 0, 1.0, N, K, None)
    np.testing.assert_allclose(host(g2), G.logreg_grad(labels, hp, 1.0), rtol=1e-6)
    g3 = nans(K, N)
    m.call("mc_compute_softmax_grad", ptr(p), ptr(g2), ptr(g3), 0, N, K, None)
    np.testing.assert_allclose(host(g3), G.softmax_grad(hp, host(g2)), rtol=1e-4, atol=1e-5)
    # the whole classifier tail in one launch (mc_softmax_logreg): every output equals the separate kernels' bit for bit,
    # with and without the gradient, overwriting and accumulating
    pf, lpf, cpf, gf = nans(K, N), nans(N), nans(N), nans(K, N)
    sf = torch.full((2,), float("nan"), dtype=torch.float64, device="cuda")
    m.call("mc_softmax_logreg", ptr(dl), ptr(dx), ptr(pf), ptr(lpf), ptr(cpf), sf.data_ptr(), ptr(gf), 0, 0.1, N, K, None)
    assert np.array_equal(host(pf), hp) and np.array_equal(host(lpf), host(lp)) and np.array_equal(host(cpf), host(cp))
    assert np.array_equal(host(sf), host(sums)) and np.array_equal(host(gf), host(g))
    base = rng.standard_normal((K, N)).astype(np.float32)
    ga, gb = dev(base), dev(base)
    m.call("mc_softmax_logreg", ptr(dl), ptr(dx), ptr(pf), ptr(lpf), ptr(cpf), sf.data_ptr(), ptr(ga), 1, 0.1, N, K, None)
    m.call("mc_compute_logreg_softmax_grad", ptr(dl), ptr(p), ptr(gb), 1, 0.1, N, K, None)
    assert np.array_equal(host(ga), host(gb))
    pf2 = nans(K, N)
    m.call("mc_softmax_logreg", ptr(dl), ptr(dx), ptr(pf2), ptr(lpf), ptr(cpf), sf.data_ptr(), None, 0, 0.1, N, K, None)
    assert np.array_equal(host(pf2), hp) and np.array_equal(host(sf), host(sums))


def test_accumulate_f64_and_autotune_switch():
    """mc_accumulate_f64: dst += src in double (the device-side running cost total of captured training steps); mc_set_autotune:
    with the timed tile trial off a first-seen shape still gives the same bits (the rule's tile), and turning it back on works."""
    import torch
    m = mc()
    a = torch.tensor([1.5, -2.25, 1e-9], dtype=torch.float64, device="cuda")
    b = torch.tensor([1e9, 0.125, 3.0], dtype=torch.float64, device="cuda")
    m.call("mc_accumulate_f64", a.data_ptr(), b.data_ptr(), 3, None)
    m.call("mc_accumulate_f64", a.data_ptr(), b.data_ptr(), 3, None)
    torch.cuda.synchronize()
    assert a.cpu().tolist() == [1.5 + 2e9, -2.0, 1e-9 + 6.0]
    rng = np.random.default_rng(77)
    mm, nn, kk = 72, 1160, 136                      # a shape no other test uses: not in the tuner's cache yet
    A, B = dev(rng.standard_normal((mm, kk)).astype(np.float32)), dev(rng.standard_normal((kk, nn)).astype(np.float32))
    off, on = nans(mm, nn), nans(mm, nn)
    m.call("mc_set_autotune", 0)
    m.call("mc_gemm", ptr(A), ptr(B), ptr(off), mm, nn, kk, 0, 0, kk, nn, nn, 0.0, 1.0, None, m.NEURON_NONE, None, 0, None)
    m.call("mc_set_autotune", 1)
    m.call("mc_gemm", ptr(A), ptr(B), ptr(on), mm, nn, kk, 0, 0, kk, nn, nn, 0.0, 1.0, None, m.NEURON_NONE, None, 0, None)
    assert np.array_equal(host(off), host(on))
    assert relerr(host(on), G.matmul(host(A), host(B))) < TOL


def test_weights_update():
    m = mc()
    rng = np.random.default_rng(22)
    for n in (2400 * 192, 1001):
        w, inc, grad = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
        for wc in (0.0, 0.004):
            dw, di = dev(w), dev(inc)
            m.call("mc_weights_update", ptr(dw), ptr(di), ptr(keep(dev(grad))), n, 0.9, 0.001, wc, None)
            ew, ei = w.copy(), inc.copy()
            G.sgd_update(ew, ei, grad, 0.001, wc, 0.9)
            # mom*inc+grad contracts to one FMA on the GPU (as nvcc does for the reference): 1-ulp tolerance
            np.testing.assert_allclose(host(di), ei, rtol=2e-6, atol=1e-6)
            np.testing.assert_allclose(host(dw), ew, rtol=2e-6, atol=1e-6)


def test_rejected_calls_report():
    m = mc()
    with pytest.raises(m.McError):
        m.call("mc_conv_filter_acts", None, None, None, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0.0, 1.0, None)
    x = zeros(16)
    with pytest.raises(m.McError, match="partialSum"):
        m.call("mc_conv_weight_acts", ptr(x), ptr(x), ptr(x), 4, 1, 2, 2, 1, 1, 0, 1, 2, 2, 1, 3, 0.0, 1.0, None)
    with pytest.raises(m.McError, match="paddingStart"):
        m.call("mc_conv_filter_acts", ptr(x), ptr(x), ptr(x), 4, 1, 2, 2, 1, 1, 1, 1, 2, 2, 1, 0.0, 1.0, None)


def test_weights_update_spans_matches_per_tensor_update():
    """One launch over the arenas == Weights::update per tensor (weights.cuh:150-164); padding between
    tensors and tensors without spans (eps == 0) are left untouched."""
    import ctypes
    import torch
    m = mc()
    rng = np.random.default_rng(23)
    sizes = [(75 * 192, 0.9, 0.001, 0.004), (192, 0.9, 0.002, 0.0), (5000, 0.0, 0.1, 0.0), (37, 0.5, 0.0, 0.0), (192 * 160, 0.9, 1.0, 1e-5)]
    offs, total = [], 0
    for n, *_ in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    w, inc, grad = (rng.standard_normal(total).astype(np.float32) for _ in range(3))
    ew, ei = w.copy(), inc.copy()

    class Span(ctypes.Structure):
        _fields_ = [("start", ctypes.c_ulonglong), ("count", ctypes.c_uint), ("mom", ctypes.c_float), ("wceps", ctypes.c_float), ("pad", ctypes.c_uint)]
    spans = []
    for (n, mom, eps, wc), o in zip(sizes, offs):
        if eps <= 0:
            continue
        G.sgd_update(ew[o:o + n], ei[o:o + n], grad[o:o + n], eps, wc, mom)
        for d in range(0, n, 4096):
            spans.append(Span(o + d, min(4096, n - d), mom, -wc * eps if wc > 0 else 0.0, 0))
    arr = (Span * len(spans))(*spans)
    dspans = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).cuda()
    dw, di, dg = dev(w), dev(inc), dev(grad)
    m.call("mc_weights_update_spans", ptr(dw), ptr(di), ptr(dg), dspans.data_ptr(), len(spans), None)
    np.testing.assert_allclose(host(di), ei, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(dw), ew, rtol=2e-6, atol=1e-6)
    o, n = offs[3], sizes[3][0]
    assert np.array_equal(host(dw)[o:o + 64], w[o:o + 64]) and np.array_equal(host(di)[o:o + 64], inc[o:o + 64])


@pytest.mark.parametrize("m_,n_,k_", [(192, 2048, 160), (96, 640, 10), (37, 53, 29), (64, 128, 4096)])
def test_gemm_relu_grad_equals_gemm_then_relu_gradient(m_, n_, k_):
    """mc_gemm_relu_grad == mc_gemm followed by the relu layer's gradient g * (y > 0), bit for bit
    (CCCPLayer::bpropActs + neuron.cuh:173-178); y has zeros (relu output) and negative garbage is masked too."""
    m = mc()
    rng = np.random.default_rng(31)
    A = rng.standard_normal((k_, m_)).astype(np.float32)          # transA: dX = W^T . G
    B = rng.standard_normal((k_, n_)).astype(np.float32)
    y = np.maximum(rng.standard_normal((m_, n_)), 0).astype(np.float32)
    dA, dB, dy = dev(A), dev(B), dev(y)
    wsb = m.lib().mc_gemm_workspace_bytes(m_, n_, k_)
    ws = zeros(max(1, wsb // 4))
    ref = nans(m_, n_)
    m.call("mc_gemm", ptr(dA), ptr(dB), ptr(ref), m_, n_, k_, 1, 0, m_, n_, n_, 0.0, 1.0, None, m.NEURON_NONE, ptr(ws), wsb, None)
    m.call("mc_neuron_input_grad", m.NEURONS["relu"], 0.0, 0.0, ptr(ref), ptr(dy), ptr(ref), 0, ref.numel(), None)
    got = nans(m_, n_)
    m.call("mc_gemm_relu_grad", ptr(dA), ptr(dB), ptr(got), m_, n_, k_, 1, 0, m_, n_, n_, 0.0, 1.0, ptr(dy), n_, ptr(ws), wsb, None)
    assert np.array_equal(host(got), host(ref))
    exp = G.matmul(A.T, B) * (y > 0)
    assert relerr(host(got), exp) < TOL


def test_max_undo_relu_grad_equals_undo_then_relu_gradient():
    m = mc()
    rng = np.random.default_rng(32)
    for (N, C, S, size, start, stride) in ((128, 96, 32, 3, 0, 2), (6, 5, 7, 3, 0, 2), (16, 8, 8, 2, 1, 2)):
        outX = int(np.ceil((S - start - size) / float(stride))) + 1
        x = np.maximum(rng.standard_normal((C * S * S, N)), 0).astype(np.float32)     # relu output: zero plateaus
        dx = dev(x)
        y = nans(C * outX * outX, N)
        m.call("mc_conv_local_pool", ptr(dx), ptr(y), N, C, S, size, start, stride, outX, m.POOL_MAX, None)
        g = dev(rng.standard_normal((C * outX * outX, N)))
        ref = nans(*x.shape)
        m.call("mc_conv_local_max_undo", ptr(dx), ptr(g), ptr(y), ptr(ref), N, C, S, size, start, stride, outX, 0.0, 1.0, None)
        m.call("mc_neuron_input_grad", m.NEURONS["relu"], 0.0, 0.0, ptr(ref), ptr(dx), ptr(ref), 0, ref.numel(), None)
        got = nans(*x.shape)
        m.call("mc_conv_local_max_undo_relu_grad", ptr(dx), ptr(g), ptr(y), ptr(got), N, C, S, size, start, stride, outX, 0.0, 1.0, None)
        assert np.array_equal(host(got), host(ref))
        e = G.max_undo(x, host(g), host(y), N, C, S, size, start, stride, outX) * (x > 0)
        assert np.array_equal(host(got), e)


@pytest.mark.parametrize("N,C,S,size,start,stride", [(128, 32, 16, 3, 0, 2), (128, 10, 8, 8, 0, 1), (6, 16, 6, 4, -2, 4), (5, 3, 7, 3, 0, 2)])
def test_avg_undo_relu_grad_equals_undo_then_relu_gradient(N, C, S, size, start, stride):
    m = mc()
    rng = np.random.default_rng(44)
    outX = int(np.ceil((S - start - size) / float(stride))) + 1
    y = dev(np.maximum(rng.standard_normal((C * S * S, N)), 0))          # the relu output that was pooled
    g = dev(rng.standard_normal((C * outX * outX, N)))
    a, b = nans(C * S * S, N), nans(C * S * S, N)
    m.call("mc_conv_local_avg_undo", ptr(g), ptr(a), N, C, S, size, start, stride, outX, 0.0, 1.0, None)
    m.call("mc_neuron_input_grad", m.NEURONS["relu"], 0.0, 0.0, ptr(a), ptr(y), ptr(a), 0, a.numel(), None)
    m.call("mc_conv_local_avg_undo_relu_grad", ptr(g), ptr(y), ptr(b), N, C, S, size, start, stride, outX, 1.0, None)
    assert np.array_equal(host(a), host(b))


@pytest.mark.parametrize("N,C,H,fs,F,pad,stride,groups", [(128, 32, 16, 5, 32, 2, 1, 1), (16, 3, 32, 5, 32, 2, 1, 1), (32, 96, 8, 5, 192, 2, 1, 1),
                                                         (128, 192, 8, 3, 192, 1, 1, 1), (12, 8, 9, 3, 16, 1, 2, 2), (5, 3, 10, 5, 4, 0, 1, 1)])
def test_conv_weight_grad_free_chunking(N, C, H, fs, F, pad, stride, groups):
    """mc_conv_weight_grad = convWeightActs + reduction of its chunks (layer.cu:503-523) with module-unaligned chunks;
    run twice: the result must not depend on timing (deterministic chunking)."""
    m = mc()
    rng = np.random.default_rng(43)
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)
    expw = G.weight_acts(img, hid, *geo, partial_sum=0)
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, groups)
    ws = nans(max(1, wsb // 4))
    prev = rng.standard_normal(expw.shape).astype(np.float32)
    dimg, dhid = dev(img), dev(hid)
    outs = []
    for _ in range(2):
        dw = dev(prev)
        m.call("mc_conv_weight_grad", ptr(dimg), ptr(dhid), ptr(dw), *geo, 0.5, 0.01, ptr(ws), wsb, None)
        outs.append(host(dw))
    assert relerr(outs[0], 0.5 * prev + 0.01 * expw) < TOL
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("N,C,H,fs,F,pad,stride,groups", [(128, 32, 16, 5, 32, 2, 1, 1), (128, 3, 32, 5, 192, 2, 1, 1), (32, 96, 8, 5, 192, 2, 1, 1),
                                                         (128, 192, 8, 3, 192, 1, 1, 1), (12, 8, 9, 3, 16, 1, 2, 2), (5, 3, 10, 5, 4, 0, 1, 1),
                                                         (7, 16, 12, 3, 160, 1, 1, 1)])
def test_conv_weight_grad_with_bias_gradient(N, C, H, fs, F, pad, stride, groups):
    """mc_conv_weight_grad_bias: the weight gradient is bit-identical to mc_conv_weight_grad's, and the per-filter bias
    gradient equals ConvLayer::bpropBiases' row sum of hidActs viewed (numFilters, modules * images) (layer.cu:491-497)."""
    m = mc()
    rng = np.random.default_rng(47)
    M = _modules(H, fs, pad, stride)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    hid = (rng.standard_normal((F * M * M, N)) + 0.25).astype(np.float32)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, groups)
    dimg, dhid = dev(img), dev(hid)
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, groups)
    ws = nans(max(1, wsb // 4))
    wshape = ((C // groups) * fs * fs, F)
    ref = nans(*wshape)
    m.call("mc_conv_weight_grad", ptr(dimg), ptr(dhid), ptr(ref), *geo, 0.0, 0.01, ptr(ws), wsb, None)
    wsbb = m.lib().mc_conv_weight_grad_bias_workspace_bytes(N, C, F, fs, M * M, groups)
    assert (wsbb == 0) == (wsb == 0) and wsbb >= wsb
    wsB = nans(max(1, wsbb // 4))
    outs = []
    for _ in range(2):
        dw, dbias = nans(*wshape), nans(F, 1)
        m.call("mc_conv_weight_grad_bias", ptr(dimg), ptr(dhid), ptr(dw), ptr(dbias), *geo, 0.0, 0.01, 0.5, ptr(wsB), wsbb, None)
        outs.append((host(dw), host(dbias)))
    assert np.array_equal(outs[0][0], host(ref))
    expb = 0.5 * hid.astype(np.float64).reshape(F, M * M * N).sum(axis=1)
    assert relerr(outs[0][1][:, 0], expb) < TOL
    assert np.array_equal(outs[0][1], outs[1][1])                    # a fixed summation order: run-to-run identical
    # accumulating weights (a matrix shared by two layers) still overwrite the bias gradient
    prev = rng.standard_normal(wshape).astype(np.float32)
    dw, dbias = dev(prev), nans(F, 1)
    m.call("mc_conv_weight_grad_bias", ptr(dimg), ptr(dhid), ptr(dw), ptr(dbias), *geo, 1.0, 0.01, 0.5, ptr(wsB), wsbb, None)
    assert relerr(host(dw), prev + host(ref).astype(np.float64)) < TOL and np.array_equal(host(dbias), outs[0][1])


@pytest.mark.parametrize("m_,n_,k_", [(160, 192, 131072), (96, 160, 131072), (192, 192, 32768), (10, 192, 8192), (192, 96, 1000), (37, 53, 29),
                                      (64, 4096, 128), (1000, 2048, 128)])
def test_gemm_row_sums(m_, n_, k_):
    """mc_gemm_row_sums: C = A . B^T bit-identical to mc_gemm(transB) and rowSums = sumScale * row sums of A (the bias gradient
    of FCLayer / CCCPLayer, layer.cu:397-400, extralayers.cu:130-138), split-K (long k) and unsplit shapes."""
    m = mc()
    rng = np.random.default_rng(53)
    A = (rng.standard_normal((m_, k_)) + 0.125).astype(np.float32)
    B = rng.standard_normal((n_, k_)).astype(np.float32)
    dA, dB = dev(A), dev(B)
    wsb = m.lib().mc_gemm_workspace_bytes(m_, n_, k_)
    ws = nans(max(1, wsb // 4))
    ref = nans(m_, n_)
    m.call("mc_gemm", ptr(dA), ptr(dB), ptr(ref), m_, n_, k_, 0, 1, k_, k_, n_, 0.0, 0.5, None, m.NEURON_NONE, ptr(ws), wsb, None)
    wsr = m.lib().mc_gemm_row_sums_workspace_bytes(m_, n_, k_)
    assert (wsr == 0) == (wsb == 0)      # (mc_gemm's figure covers either orientation's chunk plan, this one the weight-gradient plan + its sums)
    wsR = nans(max(1, wsr // 4))
    outs = []
    for _ in range(2):
        C, sums = nans(m_, n_), nans(m_, 1)
        m.call("mc_gemm_row_sums", ptr(dA), ptr(dB), ptr(C), m_, n_, k_, k_, k_, n_, 0.0, 0.5, ptr(sums), 0.25, ptr(wsR), wsr, None)
        outs.append((host(C), host(sums)))
    assert np.array_equal(outs[0][0], host(ref))
    assert relerr(outs[0][1][:, 0], 0.25 * A.astype(np.float64).sum(axis=1)) < TOL
    assert np.array_equal(outs[0][1], outs[1][1])
    # without a workspace the unsplit product delivers the same values to tolerance
    C, sums = nans(m_, n_), nans(m_, 1)
    m.call("mc_gemm_row_sums", ptr(dA), ptr(dB), ptr(C), m_, n_, k_, k_, k_, n_, 0.0, 0.5, ptr(sums), 0.25, None, 0, None)
    assert relerr(host(C), host(ref)) < TOL and relerr(host(sums)[:, 0], 0.25 * A.astype(np.float64).sum(axis=1)) < TOL


# ---------------------------------------------------------------------------------------------------------
# relu bit masks (include/mc_kernels.h "relu bit masks"): written by forward products, read by backward products
# ---------------------------------------------------------------------------------------------------------
def _decode_relu_bits(words, rows):
    """(ceil(rows/32), cols) uint32 words -> (rows, cols) bool, following the documented layout."""
    out = np.zeros((rows, words.shape[1]), dtype=bool)
    for p in range(32):
        h, r = p >> 4, p & 15
        ro = (r & 3) + 8 * (r >> 2) + 4 * h
        for rb in range(words.shape[0]):
            i = rb * 32 + ro
            if i < rows:
                out[i] = (words[rb] >> np.uint32(p)) & np.uint32(1)
    return out


def _bits_buffer(m, rows, ld):
    import torch
    n = m.lib().mc_relu_bits_words(rows, ld)
    assert n == ((rows + 31) // 32) * ld
    return torch.full((n,), -1, dtype=torch.int32, device="cuda")   # poisoned: every valid word must be written


@pytest.mark.parametrize("mm,nn,kk", [(160, 4096, 192), (96, 2048, 160), (10, 8192, 192), (192, 1000, 192), (70, 333, 50)])
def test_gemm_relu_bits_forward_and_backward(mm, nn, kk):
    m = mc()
    rng = np.random.default_rng(91)
    W = rng.standard_normal((mm, kk)).astype(np.float32)
    x = rng.standard_normal((kk, nn)).astype(np.float32)
    b = rng.standard_normal((mm,)).astype(np.float32)
    dW, dx, db = dev(W), dev(x), dev(b)
    ref = nans(mm, nn)
    m.call("mc_gemm", ptr(dW), ptr(dx), ptr(ref), mm, nn, kk, 0, 0, kk, nn, nn, 0.0, 1.0, ptr(db), m.NEURON_RELU, None, 0, None)
    out = nans(mm, nn)
    bits = _bits_buffer(m, mm, nn)
    m.call("mc_gemm_relu_bits", ptr(dW), ptr(dx), ptr(out), mm, nn, kk, 0, 0, kk, nn, nn, ptr(db), ptr(bits), None)
    y = host(out)
    assert np.array_equal(y, host(ref))
    words = host(bits).view(np.uint32).reshape(-1, nn)
    assert np.array_equal(_decode_relu_bits(words, mm), y > 0)
    # bits rebuilt from the stored activation are the same words (valid rows)
    bits2 = _bits_buffer(m, mm, nn)
    m.call("mc_relu_bits_from_acts", ptr(out), ptr(bits2), mm, nn, nn, None)
    assert np.array_equal(_decode_relu_bits(host(bits2).view(np.uint32).reshape(-1, nn), mm), y > 0)
    # backward through the layer above: dIn(mm, nn) = V^T(mm, oo) * g(oo, nn), masked by this relu
    oo = 96
    V = dev(rng.standard_normal((oo, mm)))
    g = dev(rng.standard_normal((oo, nn)))
    a, c = nans(mm, nn), nans(mm, nn)
    m.call("mc_gemm_relu_grad", ptr(V), ptr(g), ptr(a), mm, nn, oo, 1, 0, mm, nn, nn, 0.0, 0.5, ptr(out), nn, None, 0, None)
    m.call("mc_gemm_relu_grad_bits", ptr(V), ptr(g), ptr(c), mm, nn, oo, 1, 0, mm, nn, nn, 0.5, ptr(bits), None)
    assert np.array_equal(host(a), host(c))
    exp = 0.5 * (host(V).astype(np.float64).T @ host(g).astype(np.float64)) * (y > 0)
    assert relerr(host(c), exp) < TOL


@pytest.mark.parametrize("N,C,H,fs,F,pad", [(128, 3, 32, 5, 192, 2), (64, 192, 8, 3, 192, 1), (16, 32, 8, 5, 64, 2), (6, 4, 7, 3, 10, 1)])
def test_conv_relu_bits(N, C, H, fs, F, pad):
    """conv + bias + relu + bit mask (the second shape takes the split-K path, whose reduction writes the bits)."""
    m = mc()
    rng = np.random.default_rng(92)
    M = _modules(H, fs, pad, 1)
    geo = (N, C, H, H, F, fs, -pad, 1, M, M)
    img, W, b = dev(rng.standard_normal((C * H * H, N))), dev(rng.standard_normal((C * fs * fs, F))), dev(rng.standard_normal((F,)))
    ref, out = nans(F * M * M, N), nans(F * M * M, N)
    m.call("mc_conv_filter_acts_fused", ptr(img), ptr(W), ptr(ref), ptr(b), m.NEURON_RELU, *geo, 1, 0.0, 1.0, None)
    bits = _bits_buffer(m, F, M * M * N)
    m.call("mc_conv_filter_acts_relu_bits", ptr(img), ptr(W), ptr(out), ptr(b), ptr(bits), *geo, None)
    y = host(out)
    assert np.array_equal(y, host(ref))
    words = host(bits).view(np.uint32).reshape(-1, M * M * N)
    assert np.array_equal(_decode_relu_bits(words, F), y.reshape(F, M * M * N) > 0)


def test_tile_choice_never_changes_a_bit():
    """The tile autotuner picks by timing, inside a library that promises run-to-run identical results: that is sound only if
    every workgroup tile adds the k terms of an output element in the same order.  One conv triple and dense products under
    each of the forced tiles (MC_IGEMM_TILE, autotune off) and under the autotuner: all outputs bit-identical -- including the
    16x16x4 tiles, whose matrix instruction takes four k values at a time in the k order of the 32x32x2 tiles (igemm_core.h)."""
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    tmp = tempfile.mkdtemp(prefix="tiles_")
    runs = {}
    # ("5,6" ... "9,9": the 16x16x4 tiles, 160x192 / 96x160 / 192x192 / 96x192 / 96x96, which exist for products with both operands
    # k-contiguous -- the weight gradients -- and fall back to 96x128 elsewhere)
    for tile in ["1,1", "2,1", "3,1", "5,1", "2,3", "3,2", "4,0", "4,1", "5,6", "3,5", "6,6", "3,6", "9,9", "auto"]:
        env = dict(os.environ)
        if tile == "auto":
            env.pop("MC_IGEMM_TILE", None)
            env["MC_IGEMM_AUTOTUNE"] = "1"
        else:
            env["MC_IGEMM_TILE"], env["MC_IGEMM_AUTOTUNE"] = tile, "0"
        out = os.path.join(tmp, tile.replace(",", "_") + ".npz")
        r = subprocess.run([sys.executable, os.path.join(here, "run_tile_case.py"), out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        runs[tile] = dict(np.load(out))
    ref, ref16 = runs["3,1"], runs["5,6"]
    assert np.array_equal(ref["dense_bwd_mask"], ref["dense_bwd_bits"])
    mf = ("5,6", "3,5", "6,6", "3,6", "9,9")
    for tile, res in runs.items():
        for k in ref:
            if k.endswith("_sums") or k.endswith("bgrad"):
                # the bias-gradient by-product: the 32x32 tiles add it up in the consumers, the 16x16x4 tiles in the producers --
                # two fixed orders.  Identical within a family; a launch that asks for sums gets its family by rule (a pure
                # function of the shape: launch_rows_cols_epi), so the autotuned run equals one of the two, and both are the sum
                want = ref16 if tile in mf else ref
                if tile == "auto":
                    assert np.array_equal(res[k], ref[k]) or np.array_equal(res[k], ref16[k]), (tile, k)
                else:
                    assert np.array_equal(res[k], want[k]), (tile, k)
                np.testing.assert_allclose(res[k], ref[k], rtol=2e-5, atol=1e-4)
            else:
                assert np.array_equal(res[k], ref[k]), (tile, k)
