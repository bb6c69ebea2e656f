"""GPU parity at BASELINE.json's FULL sizes (bs = 128), where test_kernels_gpu.py uses reduced shapes:
 * the three contractions of every NIN cifar-10 conv layer and the largest cccp product, directly against the
   CPU oracle (its OpenMP C contractions finish a 30 GFLOP product in about a second);
 * the imagenet-224 conv1 / conv2 shapes (114.7 GFLOP per product) through a size-independent property: forward,
   backward-data and weight-gradient are the three adjoints of one trilinear form,
       <filterActs(x, W), h> = <x, imgActs(h, W)> = <W, weightActs(x, h)>,
   so each kernel is checked against the other two without an oracle pass; plus linearity of the forward product.
Tolerance: rel 1e-4 (north_star's fp32 contraction tolerance)."""
import numpy as np
import pytest

from oracle import golden as G
from gpu_util import dev, host, nans, ptr, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4


def mc():
    import mckernels
    return mckernels


def _modules(img, fs, pad, stride):
    return 1 + int(np.ceil((2 * pad + img - fs) / float(stride)))


NIN_CONVS = [("conv1", 3, 32, 5, 192, 2), ("conv2", 96, 16, 5, 192, 2), ("conv3", 192, 8, 3, 192, 1)]


@pytest.mark.parametrize("name,C,H,fs,F,pad", NIN_CONVS)
def test_nin_conv_layers_full_size_vs_oracle(name, C, H, fs, F, pad):
    m = mc()
    N = 128
    rng = np.random.default_rng(21)
    M = _modules(H, fs, pad, 1)
    img = rng.standard_normal((C * H * H, N)).astype(np.float32)
    W = (rng.standard_normal((C * fs * fs, F)) * 0.05).astype(np.float32)
    hid = rng.standard_normal((F * M * M, N)).astype(np.float32)
    bias = rng.standard_normal((F,)).astype(np.float32)
    dimg, dW, dhid, db = dev(img), dev(W), dev(hid), dev(bias)
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)

    out = nans(F * M * M, N)      # the call the runtime makes: bias + relu in the product's store
    m.call("mc_conv_filter_acts_fused", ptr(dimg), ptr(dW), ptr(out), ptr(db), m.NEURON_RELU, *geo, 0.0, 1.0, None)
    lin = G.filter_acts(img, W, *geo).astype(np.float64).reshape(F, -1) + bias[:, None]
    exp = np.maximum(lin, 0).reshape(F * M * M, N)
    assert np.abs(host(out) - exp).max() < TOL * np.sqrt((lin ** 2).mean())

    din = nans(C * H * H, N)
    m.call("mc_conv_img_acts", ptr(dhid), ptr(dW), ptr(din), *geo, 0.0, 1.0, None)
    assert relerr(host(din), G.img_acts(hid, W, *geo)) < TOL

    dw = nans(C * fs * fs, F)     # the runtime's weight-gradient entry (device-sized chunks), eps/N scale
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, 1)
    ws = nans(max(1, wsb // 4))
    m.call("mc_conv_weight_grad", ptr(dimg), ptr(dhid), ptr(dw), *geo, 0.0, 0.01 / N, ptr(ws), wsb, None)
    expw = G.weight_acts(img, hid, *geo, partial_sum=0).astype(np.float64) * (0.01 / N)
    assert relerr(host(dw), expw) < TOL


def test_nin_cccp1_full_size_vs_oracle():
    """cccp1 of NIN at bs128: (160 x 192) . (192 x 131072), forward with bias + relu, backward, weight gradient."""
    m = mc()
    rng = np.random.default_rng(22)
    out_, in_, cols = 160, 192, 1024 * 128
    Wt = (rng.standard_normal((out_, in_)) * 0.05).astype(np.float32)
    X = rng.standard_normal((in_, cols)).astype(np.float32)
    Gd = rng.standard_normal((out_, cols)).astype(np.float32)
    bias = rng.standard_normal((out_,)).astype(np.float32)
    dW, dX, dG, db = dev(Wt), dev(X), dev(Gd), dev(bias)
    Y = nans(out_, cols)
    m.call("mc_gemm", ptr(dW), ptr(dX), ptr(Y), out_, cols, in_, 0, 0, in_, cols, cols, 0.0, 1.0, ptr(db), m.NEURON_RELU, None, 0, None)
    lin = G.matmul(Wt, X).astype(np.float64) + bias[:, None]
    assert np.abs(host(Y) - np.maximum(lin, 0)).max() < TOL * np.sqrt((lin ** 2).mean())
    dIn = nans(in_, cols)
    m.call("mc_gemm", ptr(dW), ptr(dG), ptr(dIn), in_, cols, out_, 1, 0, in_, cols, cols, 0.0, 1.0, None, m.NEURON_NONE, None, 0, None)
    assert relerr(host(dIn), G.matmul(Wt.T, Gd)) < TOL
    ws_bytes = m.lib().mc_gemm_workspace_bytes(out_, in_, cols)
    import torch
    ws = torch.zeros(max(1, ws_bytes // 4), dtype=torch.float32, device="cuda")
    dWg = nans(out_, in_)
    m.call("mc_gemm", ptr(dG), ptr(dX), ptr(dWg), out_, in_, cols, 0, 1, cols, cols, in_, 0.0, 1.0, None, m.NEURON_NONE, ptr(ws), ws_bytes, None)
    assert relerr(host(dWg), Gd.astype(np.float64) @ X.astype(np.float64).T) < TOL


INET = [("conv1", 3, 224, 11, 96, 0, 4), ("conv2", 96, 27, 5, 256, 2, 1)]


@pytest.mark.parametrize("name,C,H,fs,F,pad,stride", INET)
def test_imagenet_conv_adjoint_identities_full_size(name, C, H, fs, F, pad, stride):
    import torch
    m = mc()
    N = 128
    M = _modules(H, fs, pad, stride)
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.randn(C * H * H, N, device="cuda", generator=g)
    x2 = torch.randn(C * H * H, N, device="cuda", generator=g)
    W = torch.randn(C * fs * fs, F, device="cuda", generator=g) * 0.05
    h = torch.randn(F * M * M, N, device="cuda", generator=g)
    geo = (N, C, H, H, F, fs, -pad, stride, M, M, 1)
    y = nans(F * M * M, N)
    m.call("mc_conv_filter_acts", ptr(x), ptr(W), ptr(y), *geo, 0.0, 1.0, None)
    dx = nans(C * H * H, N)
    m.call("mc_conv_img_acts", ptr(h), ptr(W), ptr(dx), *geo, 0.0, 1.0, None)
    dw = nans(C * fs * fs, F)
    wsb = m.lib().mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, 1)
    ws = nans(max(1, wsb // 4))
    m.call("mc_conv_weight_grad", ptr(x), ptr(h), ptr(dw), *geo, 0.0, 1.0, ptr(ws), wsb, None)
    torch.cuda.synchronize()
    assert not (torch.isnan(y).any() or torch.isnan(dx).any() or torch.isnan(dw).any())

    def dot(a, b):
        return float((a.double() * b.double()).sum())

    a, b, c = dot(y, h), dot(x, dx), dot(W, dw)
    # |<y, h>| of independent tensors is about ||y|| ||h|| / sqrt(n): the natural scale of the comparison
    scale = float(y.double().norm() * h.double().norm()) / np.sqrt(y.numel())
    print(name, a, b, c, scale)
    assert abs(a - b) < TOL * scale and abs(a - c) < TOL * scale

    # linearity of the forward product in the image: conv(x + 2 x2) = conv(x) + 2 conv(x2)
    y2, y12 = nans(F * M * M, N), nans(F * M * M, N)
    m.call("mc_conv_filter_acts", ptr(x2), ptr(W), ptr(y2), *geo, 0.0, 1.0, None)
    xs = x + 2 * x2
    m.call("mc_conv_filter_acts", ptr(xs), ptr(W), ptr(y12), *geo, 0.0, 1.0, None)
    torch.cuda.synchronize()
    err = float((y12 - (y + 2 * y2)).abs().max()) / float(y12.double().pow(2).mean().sqrt())
    assert err < TOL
