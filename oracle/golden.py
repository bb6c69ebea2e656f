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
