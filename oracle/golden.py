"""oracle/golden.py -- TEST INFRASTRUCTURE ONLY (see oracle/golden.c header).

ctypes bindings for the C oracle plus NumPy restatements of the cheap ops
(neurons, softmax, logreg cost, dense products, SGD update).  Every function
cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

All arrays are float32, C-contiguous, shaped (features, numCases) -- the
reference's universal layout (Kernel/src/conv/filter_acts.cu:39-43).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/golden.c -> oracle/libgolden.so (gcc, a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libgolden.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgolden.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.c_void_p)


_i, _f = ctypes.c_int, ctypes.c_float


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------
# conv / local contractions (C oracle)
# --------------------------------------------------------------------------
def filter_acts(images, filters, N, C, imgY, imgX, F, fs, pad_start, stride, mY, mX, groups=1,
                targets=None, scale_targets=0.0, scale_output=1.0, conv=True):
    """convFilterActs / localFilterActs (Kernel/src/conv/filter_acts.cu:2206-2224)."""
    if targets is None:
        targets = np.zeros((F * mY * mX, N), np.float32)
    lib().gold_filter_acts(_p(images), _p(filters), _p(targets), _i(N), _i(C), _i(imgY), _i(imgX), _i(F),
                           _i(fs), _i(pad_start), _i(stride), _i(mY), _i(mX), _i(groups),
                           _f(scale_targets), _f(scale_output), _i(int(conv)))
    return targets


def img_acts(hid, filters, N, C, imgY, imgX, F, fs, pad_start, stride, mY, mX, groups=1,
             targets=None, scale_targets=0.0, scale_output=1.0, conv=True):
    """convImgActs / localImgActs (Kernel/src/conv/img_acts.cu:2040-2056)."""
    if targets is None:
        targets = np.zeros((C * imgY * imgX, N), np.float32)
    lib().gold_img_acts(_p(hid), _p(filters), _p(targets), _i(N), _i(C), _i(imgY), _i(imgX), _i(F),
                        _i(fs), _i(pad_start), _i(stride), _i(mY), _i(mX), _i(groups),
                        _f(scale_targets), _f(scale_output), _i(int(conv)))
    return targets


def weight_acts(images, hid, N, C, imgY, imgX, F, fs, pad_start, stride, mY, mX, groups=1, partial_sum=0,
                targets=None, scale_targets=0.0, scale_output=1.0, conv=True):
    """convWeightActs / localWeightActs (Kernel/src/conv/weight_acts.cu:1092-1108)."""
    M = mY * mX
    ps = 1 if not conv else (M if partial_sum <= 0 else partial_sum)
    rows = (M // ps) * (C // groups) * fs * fs
    if targets is None:
        targets = np.zeros((rows, F), np.float32)
    lib().gold_weight_acts(_p(images), _p(hid), _p(targets), _i(N), _i(C), _i(imgY), _i(imgX), _i(F),
                           _i(fs), _i(pad_start), _i(stride), _i(mY), _i(mX), _i(groups), _i(partial_sum),
                           _f(scale_targets), _f(scale_output), _i(int(conv)))
    return targets


# --------------------------------------------------------------------------
# pooling / normalisation (C oracle)
# --------------------------------------------------------------------------
def pool(imgs, N, C, img_size, size_x, start, stride, outputs_x, kind):
    """convLocalPool (Kernel/include/conv/conv_util.cuh:335-524)."""
    out = np.zeros((C * outputs_x * outputs_x, N), np.float32)
    lib().gold_pool(_p(imgs), _p(out), _i(N), _i(C), _i(img_size), _i(size_x), _i(start), _i(stride),
                    _i(outputs_x), _i(int(kind == "max")))
    return out


def max_undo(imgs, max_grads, max_acts, N, C, img_size, size_x, start, stride, outputs_x,
             target=None, scale_targets=0.0, scale_output=1.0):
    """convLocalMaxUndo (Kernel/src/conv/conv_util.cu:1879-1968)."""
    if target is None:
        target = np.zeros((C * img_size * img_size, N), np.float32)
    lib().gold_max_undo(_p(imgs), _p(max_grads), _p(max_acts), _p(target), _i(N), _i(C), _i(img_size),
                        _i(size_x), _i(start), _i(stride), _i(outputs_x), _f(scale_targets), _f(scale_output))
    return target


def avg_undo(avg_grads, N, C, img_size, size_x, start, stride, outputs_x,
             target=None, scale_targets=0.0, scale_output=1.0):
    """convLocalAvgUndo (Kernel/src/conv/conv_util.cu:1978-2072)."""
    if target is None:
        target = np.zeros((C * img_size * img_size, N), np.float32)
    lib().gold_avg_undo(_p(avg_grads), _p(target), _i(N), _i(C), _i(img_size), _i(size_x), _i(start),
                        _i(stride), _i(outputs_x), _f(scale_targets), _f(scale_output))
    return target


def rnorm(imgs, N, C, img_size, size, add_scale, pow_scale, mean_diffs=None):
    """convResponseNorm / convContrastNorm (Kernel/src/conv/conv_util.cu:2074-2222)."""
    denoms = np.zeros_like(imgs)
    out = np.zeros_like(imgs)
    md = imgs if mean_diffs is None else mean_diffs
    lib().gold_cnorm(_p(imgs), _p(md), _p(denoms), _p(out), _i(N), _i(C), _i(img_size), _i(size),
                     _f(add_scale), _f(pow_scale))
    return out, denoms


def rnorm_undo(out_grads, denoms, inputs, acts, N, C, img_size, size, add_scale, pow_scale,
               target=None, scale_targets=0.0, scale_output=1.0):
    """convResponseNormUndo (Kernel/src/conv/conv_util.cu:2238-2424).  `acts` is clobbered."""
    if target is None:
        target = np.zeros_like(inputs)
    lib().gold_rnorm_undo(_p(out_grads), _p(denoms), _p(inputs), _p(acts), _p(target), _i(N), _i(C),
                          _i(img_size), _i(size), _f(add_scale), _f(pow_scale), _f(scale_targets), _f(scale_output))
    return target


def cmrnorm(imgs, N, C, img_size, size, add_scale, pow_scale, blocked=False, k=1.0):
    """convResponseNormCrossMap (Kernel/src/conv/conv_util.cu:2753-2809, 2902-2904)."""
    denoms = np.zeros_like(imgs)
    out = np.zeros_like(imgs)
    lib().gold_cmrnorm(_p(imgs), _p(denoms), _p(out), _i(N), _i(C), _i(img_size), _i(size), _f(add_scale),
                       _f(pow_scale), _i(int(blocked)), _f(k))
    return out, denoms


def cmrnorm_undo(out_grads, denoms, inputs, acts, N, C, img_size, size, add_scale, pow_scale, blocked=False,
                 target=None, scale_targets=0.0, scale_output=1.0):
    """convResponseNormCrossMapUndo (Kernel/src/conv/conv_util.cu:2811-2900).  `acts` is clobbered."""
    if target is None:
        target = np.zeros_like(inputs)
    lib().gold_cmrnorm_undo(_p(out_grads), _p(denoms), _p(inputs), _p(acts), _p(target), _i(N), _i(C),
                            _i(img_size), _i(size), _f(add_scale), _f(pow_scale), _i(int(blocked)),
                            _f(scale_targets), _f(scale_output))
    return target


# --------------------------------------------------------------------------
# dense products (FC / cccp), NumPy in float64 then one rounding
# --------------------------------------------------------------------------
def matmul(a, b):
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)


# --------------------------------------------------------------------------
# neurons (Kernel/include/layers/neuron.cuh:65-462,
#          PluginsSrc/include/extralayers.cuh:34-102)
# --------------------------------------------------------------------------
def neuron_fwd(kind, params, x, train=True, mask=None):
    x = _f32(x)
    if kind == "ident":
        return x.copy()
    if kind == "relu":                                   # neuron.cuh:166-171
        return np.where(x < 0, np.float32(0), x)
    if kind == "brelu":                                  # :207-216
        a = np.float32(params["a"])
        return np.where(x < 0, np.float32(0), np.where(x > a, a, x))
    if kind == "abs":
        return np.abs(x)
    if kind == "logistic":
        return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    if kind == "tanh":                                   # :289-298  a*(2/(1+e^(-2bx)) - 1)
        a, b = params["a"], params["b"]
        return (a * (2.0 / (1.0 + np.exp(x.astype(np.float64) * (-2 * b))) - 1.0)).astype(np.float32)
    if kind == "softrelu":                               # :338-345
        xd = x.astype(np.float64)
        return np.where(xd > 4.0, xd, np.log1p(np.exp(np.minimum(xd, 4.0)))).astype(np.float32)
    if kind == "square":
        return x * x
    if kind == "sqrt":
        return np.sqrt(x)
    if kind == "linear":                                 # :449-456
        return (np.float32(params["a"]) * x + np.float32(params["b"])).astype(np.float32)
    if kind == "dropout":                                # extralayers.cuh:39-56, 81-96
        p = np.float32(params["prob"])
        if train:
            assert mask is not None, "dropout mask must be injected (reference RNG is wall-clock seeded)"
            return np.where(mask < p, np.float32(0), x)
        return (x * np.float32(1 - p)).astype(np.float32)
    raise ValueError("unknown neuron " + kind)


def neuron_bwd(kind, params, g, y, x, train=True, mask=None):
    """Input gradient.  `y` = the neuron's output, `x` = its input (only abs/square use x)."""
    g = _f32(g)
    if kind == "ident":
        return g.copy()
    if kind == "relu":                                   # neuron.cuh:173-178
        return g * (y > 0)
    if kind == "brelu":                                  # :218-227
        return g * (y > 0) * (y < np.float32(params["a"]))
    if kind == "abs":                                    # :255-260
        return g * np.where(x > 0, np.float32(1), np.float32(-1))
    if kind == "logistic":                               # :134-139
        return g * y * (np.float32(1) - y)
    if kind == "tanh":                                   # :300-310
        a, b = params["a"], params["b"]
        t = (1.0 - y.astype(np.float64) / a) / 2.0
        return (g * (-4 * a * b) * (t * (t - 1.0))).astype(np.float32)
    if kind == "softrelu":                               # :347-356 (fed the OUTPUT, upstream quirk)
        yd = y.astype(np.float64)
        f = np.exp(np.minimum(yd, 4.0))
        return np.where(yd > 4.0, g, g * (f / (1.0 + f))).astype(np.float32)
    if kind == "square":                                 # :384-389
        return g * np.float32(2) * x
    if kind == "sqrt":                                   # :416-421
        return g / (np.float32(2) * y)
    if kind == "linear":                                 # :440-442
        return g * np.float32(params["a"])
    if kind == "dropout":                                # extralayers.cuh:58-78
        p = np.float32(params["prob"])
        if train:
            return np.where(mask < p, np.float32(0), g)
        return (g * np.float32(1 - p)).astype(np.float32)
    raise ValueError("unknown neuron " + kind)


# --------------------------------------------------------------------------
# softmax + logistic-regression cost
# --------------------------------------------------------------------------
def softmax(x):
    """SoftmaxLayer::fpropActs (Kernel/src/layers/layer.cu:612-622); x is (classes, N)."""
    xd = x.astype(np.float64)
    e = np.exp(xd - xd.max(axis=0, keepdims=True))
    return (e / e.sum(axis=0, keepdims=True)).astype(np.float32)


def logreg_cost(labels, probs):
    """kLogregCost (Kernel/src/layers/layer_kernels.cu:41-74) + LogregCostLayer::fpropActs
    (layer.cu:993-1005): returns [-sum log p_label, N - sum correctProb]."""
    n = probs.shape[1]
    lab = labels.reshape(-1).astype(np.int64)
    pl = probs[lab, np.arange(n)]
    mx = probs.max(axis=0)
    num_max = (probs == mx[None, :]).sum(axis=0)
    correct = np.where(pl == mx, 1.0 / num_max, 0.0)
    return [float(-np.log(pl.astype(np.float64)).sum()), float(n - correct.sum())]


def logreg_softmax_grad(labels, probs, coeff):
    """kLogregSoftmaxGrad (layer_kernels.cu:137-152): coeff * (1[j==label] - y_j)."""
    n = probs.shape[1]
    onehot = np.zeros_like(probs)
    onehot[labels.reshape(-1).astype(np.int64), np.arange(n)] = 1
    return (np.float32(coeff) * (onehot - probs)).astype(np.float32)


def logreg_grad(labels, probs, coeff):
    """kLogregCostGrad (layer_kernels.cu:84-101): coeff * 1[j==label] / y_j."""
    n = probs.shape[1]
    g = np.zeros_like(probs)
    lab = labels.reshape(-1).astype(np.int64)
    g[lab, np.arange(n)] = np.float32(coeff) / probs[lab, np.arange(n)]
    return g


def softmax_grad(y, g):
    """kSoftmaxGrad (layer_kernels.cu:109-128): y_j * (g_j - sum_k g_k y_k)."""
    yd, gd = y.astype(np.float64), g.astype(np.float64)
    return (yd * (gd - (gd * yd).sum(axis=0, keepdims=True))).astype(np.float32)


# --------------------------------------------------------------------------
# SGD update (Kernel/include/layers/weights.cuh:150-164)
# --------------------------------------------------------------------------
def sgd_update(w, inc, grad, eps, wc, mom):
    inc[...] = (np.float32(mom) * inc + grad).astype(np.float32)
    if wc > 0:
        inc += (np.float32(-wc * eps) * w).astype(np.float32)
    w += inc


# --------------------------------------------------------------------------
# image layers and element-wise max (NumPy)
# --------------------------------------------------------------------------
def _blend(target, value, scale_targets, scale_output):
    value = np.float32(scale_output) * value.astype(np.float32)
    if scale_targets == 0:
        return value.astype(np.float32)
    return (np.float32(scale_targets) * target + value).astype(np.float32)


def bed_of_nails(imgs, N, C, img_size, start, stride, target=None, scale_targets=0.0, scale_output=1.0):
    """convBedOfNails (Kernel/src/conv/conv_util.cu:667-670; kernel :463-537, forward branch):
    output (oy, ox) = pixel (start + oy*stride, start + ox*stride), outputsX = DIVUP(imgSize, stride) (:554)."""
    assert stride > 1                                   # conv_util.cu:552
    ox = (img_size + stride - 1) // stride
    x = imgs.reshape(C, img_size, img_size, N)
    nails = x[:, start:start + (ox - 1) * stride + 1:stride, start:start + (ox - 1) * stride + 1:stride, :]
    assert nails.shape[1] == ox and nails.shape[2] == ox
    nails = nails.reshape(C * ox * ox, N)
    return _blend(target, nails, scale_targets, scale_output)


def bed_of_nails_undo(acts_grad, N, C, img_size, start, stride, target=None, scale_targets=0.0, scale_output=1.0):
    """convBedOfNailsUndo (conv_util.cu:672-676; kernel reverse branch :502-533).  scaleTargets == 0: the target is
    zeroed first (:563-565); otherwise only the nail pixels are updated, the rest of the target is left as is."""
    assert stride > 1
    ox = (img_size + stride - 1) // stride
    g = acts_grad.reshape(C, ox, ox, N)
    if scale_targets == 0:
        out = np.zeros((C, img_size, img_size, N), np.float32)
    else:
        out = target.reshape(C, img_size, img_size, N).copy()
    sl = (slice(None), slice(start, start + (ox - 1) * stride + 1, stride), slice(start, start + (ox - 1) * stride + 1, stride))
    out[sl] = _blend(out[sl], g, scale_targets, scale_output)
    return out.reshape(C * img_size * img_size, N)


def _fma32(a, b, c):
    """float32 fused multiply-add emulated in float64 (the product of two float32 is exact in float64)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def gaussian_blur(imgs, filt, N, C, img_size, horiz, target=None, scale_targets=0.0, scale_output=1.0):
    """convGaussianBlur (conv_util.cu:684-728; kernel :358-443): one direction of the separable blur,
    out[x] = sum_{d=-r..r} in[x+d] * filter[r+d] with zeros outside the line; the kernel adds the terms in
    ascending x+d, each as a multiply-add.  The reference's filters are symmetric (layer.py:443-446), which
    is the only case in which its prologue (filter[r+d]) and main loop (filter[r-d]) agree."""
    f = np.asarray(filt, np.float32).reshape(-1)
    r = f.size // 2
    assert f.size == 2 * r + 1 and 1 <= r <= 4 and img_size >= 2 * r + 1      # conv_util.cu:692-693
    x = imgs.reshape(C, img_size, img_size, N)
    axis = 2 if horiz else 1
    pad = [(0, 0)] * 4
    pad[axis] = (r, r)
    xp = np.pad(x, pad)
    acc = np.zeros_like(x)
    for k in range(2 * r + 1):
        sl = [slice(None)] * 4
        sl[axis] = slice(k, k + img_size)
        acc = _fma32(f[k], xp[tuple(sl)], acc)
    return _blend(target, acc.reshape(C * img_size * img_size, N), scale_targets, scale_output)


def resize_bilinear(imgs, N, C, img_size, tgt_size, scale):
    """convResizeBilinear (conv_util.cu:2426-2474; kernel :290-341)."""
    f32 = np.float32
    center_scale = f32(img_size * 0.5 - tgt_size * 0.5 * float(f32(scale)))      # double, then float (:2439-2441)
    px = np.arange(tgt_size, dtype=np.float32)
    src = np.maximum(f32(0), np.minimum(f32(img_size) - f32(1.01), _fma32(px, f32(scale), center_scale)))   # :314-315
    fl = np.floor(src)
    u = (np.floor(src + f32(1)) - src).astype(np.float32)      # weight of the left pixel (:317)
    w = (src - fl).astype(np.float32)                          # weight of the bottom row (:318)
    i0 = fl.astype(np.int64)
    x = imgs.reshape(C, img_size, img_size, N)
    iy, ix = i0[:, None], i0[None, :]
    v0, v1 = x[:, iy, ix, :], x[:, iy, ix + 1, :]
    v2, v3 = x[:, iy + 1, ix, :], x[:, iy + 1, ix + 1, :]
    uu = u[None, None, :, None]
    ww = w[None, :, None, None]
    c0 = _fma32(uu, v0 - v1, v1)                               # :334
    c1 = _fma32(uu, v2 - v3, v3)                               # :335
    out = _fma32(ww, c1 - c0, c0)                              # :337
    return out.reshape(C * tgt_size * tgt_size, N)


def rgb_to_yuv(imgs, N, img_pixels):
    """convRGBToYUV (conv_util.cu:2480-2524; kernel :172-194)."""
    x = imgs.reshape(3, img_pixels, N).astype(np.float64)
    m = np.array([[0.2126, 0.7152, 0.0722], [-0.09991, -0.33609, 0.436], [0.615, -0.55861, -0.05639]], np.float32).astype(np.float64)
    return np.einsum("oc,cpn->opn", m, x).astype(np.float32).reshape(3 * img_pixels, N)


def rgb_to_lab(imgs, N, img_pixels, center=False):
    """convRGBToLAB (conv_util.cu:2526-2601; kernel :237-268, labf :196-201).  As in the kernel, the XYZ matrix
    is applied WITHOUT the 5.6506753 factor its comment mentions, and the cube root is x ** 0.3333."""
    x = imgs.reshape(3, img_pixels, N).astype(np.float64)
    m = np.array([[0.49, 0.31, 0.2], [0.17697, 0.8124, 0.01063], [0.0, 0.01, 0.99]], np.float32).astype(np.float64)
    xyz = np.einsum("oc,cpn->opn", m, x).astype(np.float32)

    def labf(v):
        v64 = v.astype(np.float64)
        return np.where(v > np.float32(0.0088564517), np.power(np.maximum(v64, 1e-30), float(np.float32(0.3333))),
                        float(np.float32(7.787037)) * v64 + float(np.float32(0.13793103))).astype(np.float32)

    lx, ly, lz = labf(xyz[0]), labf(xyz[1]), labf(xyz[2])
    L = np.float32(116.0) * ly - np.float32(16.0) - np.float32(50.0 if center else 0.0)
    a = np.float32(500.0) * (lx - ly)
    b = np.float32(200.0) * (ly - lz)
    return np.stack([L, a, b]).astype(np.float32).reshape(3 * img_pixels, N)


def eltwise_max(inputs):
    """EltwiseMaxLayer::fpropActs (Kernel/src/layers/layer.cu:670-676)."""
    out = np.maximum(inputs[0], inputs[1])
    for x in inputs[2:]:
        out = np.maximum(out, x)
    return out.astype(np.float32)


def eltwise_max_grad(act_grad, inp, out, target=None, add=False):
    """computeEltwiseMaxGrad (Kernel/src/layers/layer_kernels.cu:155-186): grad * (output == input)."""
    g = (act_grad * (out == inp)).astype(np.float32)
    return (target + g).astype(np.float32) if add else g


def group_sparsity_in_label_cost(labels, acts, channels):
    """GroupSparsityInLabelCostLayer::fpropActs (PluginsSrc/src/extralayers.cu:10-32) with
    CalculateSqrtSumSquareMatrix (PluginsSrc/src/extrautils.cu:128-158): per channel and per distinct label value,
    S = sqrt(count * sum over that label's cases and all pixels of acts^2); returns (cost = sum S, S per case
    (channels, N), count per case (N,))."""
    N = acts.shape[1]
    lab = labels.reshape(-1)
    q = (acts.reshape(channels, -1, N).astype(np.float64) ** 2).sum(axis=1)          # (channels, N)
    same = lab[:, None] == lab[None, :]                                              # (N, N)
    counts = same.sum(axis=1)
    S = np.sqrt(counts[None, :] * (q @ same.astype(np.float64)))                     # S of case n's label, (channels, N)
    first = np.array([not same[n, :n].any() for n in range(N)])
    return float(S[:, first].sum()), S.astype(np.float32), counts.astype(np.float32)


def group_sparsity_in_label_grad(labels, acts, channels, coeff, target=None, scale_targets=0.0):
    """bpropActs (extralayers.cu:34-38) with kCalculateGradient (extrautils.cu:167-199):
    actsGrad = scaleTargets*actsGrad - coeff * count * acts / (S + 2.5e-7)."""
    N = acts.shape[1]
    _, S, counts = group_sparsity_in_label_cost(labels, acts, channels)
    a = acts.reshape(channels, -1, N).astype(np.float64)
    g = (counts[None, None, :] * a / (S[:, None, :].astype(np.float64) + 2.5e-7)).reshape(acts.shape)
    out = -float(coeff) * g
    if scale_targets != 0:
        out = scale_targets * target.astype(np.float64) + out
    return out.astype(np.float32)


# --------------------------------------------------------------------------
# random sparse connectivity (randSparse conv / local layers)
# --------------------------------------------------------------------------
def gather_channels(imgs, conns, C, N):
    """The channel indirection of the *Sparse kernels (Kernel/src/conv/filter_acts.cu:259-413: filter group g reads
    channel colorIndices[g*numFilterColors + c]) as a tensor: out channel k = input channel conns[k]."""
    x = imgs.reshape(C, -1, N)
    return _f32(x[np.asarray(conns, np.int64)].reshape(-1, N))


def scatter_channels(gathered, conns, C, N, target=None, scale_targets=0.0, scale_output=1.0):
    """Backward of gather_channels: every real channel sums the gradients of its copies, in ascending position
    (convImgActsSparse + the overSample sum of Kernel/src/layers/layer.cu:526-535)."""
    conns = np.asarray(conns, np.int64)
    g = gathered.reshape(len(conns), -1, N)
    out = np.zeros((C, g.shape[1], N), np.float32)
    for k, c in enumerate(conns):
        out[c] = out[c] + g[k]
    return _blend(target, out.reshape(-1, N), scale_targets, scale_output)
