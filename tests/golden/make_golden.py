"""Generates tests/golden/*.npz -- small input/expected-output vectors for the hot path.

Provenance (be exact about what pins what):
  * datagen_*.npz  re-create the ONLY reference-authored test cases for this path, the MATLAB generators
    Kernel/unittest/{filter_acts,img_acts,weight_acts}/DataGen.m (sizes: 10x10 images, 2 colours, 10
    images, 5x5 filters, 4 / 30 / 2 filters, padding 0, stride 1; the reference stores no vectors and
    MATLAB/Octave are not available here).  Their expected values are what the generators compute:
    convn(img, flipped filter, 'valid') = cross-correlation (filter_acts/DataGen.m:50), full
    convn(act, filter) (img_acts/DataGen.m:52) and the sum over images of valid correlations
    (weight_acts/DataGen.m:50) -- computed here with scipy.signal in float64, INDEPENDENTLY of oracle/.
  * ops_*.npz      seeded inputs with the CPU oracle's outputs for the ops no reference-authored data covers
    (pool / undo, norms, softmax + logreg, neurons, SGD update, cccp).  These are regression pins of the
    oracle itself ("parity unpinned by reference data"): they keep oracle and HIP
    kernels from drifting together.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.signal import convolve2d, correlate2d

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import golden as G  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(os.path.join(HERE, name)) / 1024.0))


def datagen():
    N, C, H, fs, M = 10, 2, 10, 5, 6
    rng = np.random.default_rng(20111)
    # filter_acts/DataGen.m: 4 filters
    F = 4
    img = rng.standard_normal((C, H, H, N)).astype(np.float32)
    W = rng.standard_normal((C, fs, fs, F)).astype(np.float32)
    exp = np.zeros((F, M, M, N))
    for f in range(F):
        for n in range(N):
            for c in range(C):
                exp[f, :, :, n] += correlate2d(img[c, :, :, n].astype(np.float64), W[c, :, :, f].astype(np.float64), mode="valid")
    save("datagen_filter_acts.npz", images=img.reshape(-1, N), filters=W.reshape(-1, F), expected=exp.reshape(-1, N),
         geom=np.array([N, C, H, H, F, fs, 0, 1, M, M, 1]))
    # img_acts/DataGen.m: 30 filters
    F = 30
    hid = rng.standard_normal((F, M, M, N)).astype(np.float32)
    W = rng.standard_normal((C, fs, fs, F)).astype(np.float32)
    exp = np.zeros((C, H, H, N))
    for c in range(C):
        for n in range(N):
            for f in range(F):
                exp[c, :, :, n] += convolve2d(hid[f, :, :, n].astype(np.float64), W[c, :, :, f].astype(np.float64), mode="full")
    save("datagen_img_acts.npz", hidacts=hid.reshape(-1, N), filters=W.reshape(-1, F), expected=exp.reshape(-1, N),
         geom=np.array([N, C, H, H, F, fs, 0, 1, M, M, 1]))
    # weight_acts/DataGen.m: 2 filters, partialSum = 1 then sum over modules (test.cu:62-66)
    F = 2
    img = rng.standard_normal((C, H, H, N)).astype(np.float32)
    hid = rng.standard_normal((F, M, M, N)).astype(np.float32)
    exp = np.zeros((C, fs, fs, F))
    for c in range(C):
        for f in range(F):
            for n in range(N):
                exp[c, :, :, f] += correlate2d(img[c, :, :, n].astype(np.float64), hid[f, :, :, n].astype(np.float64), mode="valid")
    save("datagen_weight_acts.npz", images=img.reshape(-1, N), hidacts=hid.reshape(-1, N), expected=exp.reshape(-1, F),
         geom=np.array([N, C, H, H, F, fs, 0, 1, M, M, 1]))


def datagen_group_sparsity():
    """Kernel/unittest/groupSparsity/DataGen.m:1-46, statement by statement in NumPy float64 (16 x 16 x 3 inputs, 12 cases,
    labels 1..10; MATLAB's rand -> a seeded generator): per case and channel the sum of squares, summed per distinct label,
    times that label's case count, square root; cost = the sum of those roots; gradient = count * data / root.  INDEPENDENT of
    oracle/ (no import from it is used here).  test.cu:74-80 compares CalculateSqrtSumSquareMatrix / CalculateGradient to it."""
    channels, cases, imgSize, labelnum = 3, 12, 16, 10
    rng = np.random.default_rng(20113)
    pix = imgSize * imgSize
    data = rng.random((pix * channels, cases)).astype(np.float32).astype(np.float64)   # the values the fp32 kernels see
    labels = rng.integers(1, labelnum + 1, size=cases)
    sq = data ** 2
    compact = np.zeros((channels, cases))
    for i in range(cases):
        for j in range(channels):
            compact[j, i] += sq[j * pix:(j + 1) * pix, i].sum()
    a, c = np.unique(labels, return_inverse=True)
    comcompact = np.zeros((channels, len(a)))
    for i in range(cases):
        comcompact[:, c[i]] += compact[:, i]
    for i in range(len(a)):
        comcompact[:, i] *= (labels == a[i]).sum()
    comcompact = np.sqrt(comcompact)
    cost = comcompact.sum()
    gradient = np.zeros_like(data)
    for i in range(cases):
        for j in range(channels):
            index = int(np.where(a == labels[i])[0][0])
            gradient[j * pix:(j + 1) * pix, i] = (labels == labels[i]).sum() * data[j * pix:(j + 1) * pix, i] / comcompact[j, index]
    save("datagen_group_sparsity.npz", acts=data.astype(np.float32), labels=labels.astype(np.float32).reshape(1, cases),
         expected_grad=gradient, expected_cost=np.array(cost), geom=np.array([cases, channels, pix]))


def ops():
    rng = np.random.default_rng(20112)
    N = 12
    # pooling + undo: NIN pool1-like (max 3/2), global avg (8/1), negative start (layers.gc.cfg), start=1 (imagenet)
    pool = {}
    for i, (kind, C, S, size, start, stride) in enumerate((("max", 4, 8, 3, 0, 2), ("avg", 3, 8, 8, 0, 1), ("avg", 4, 6, 4, -2, 4), ("max", 4, 8, 2, 1, 2))):
        outX = int(np.ceil((S - start - size) / float(stride))) + 1
        x = rng.standard_normal((C * S * S, N)).astype(np.float32)
        if kind == "max":
            x = np.maximum(x, 0)                       # ReLU plateaus: ties route the gradient to every zero
        y = G.pool(x, N, C, S, size, start, stride, outX, kind)
        g = rng.standard_normal(y.shape).astype(np.float32)
        d = G.max_undo(x, g, y, N, C, S, size, start, stride, outX) if kind == "max" else G.avg_undo(g, N, C, S, size, start, stride, outX)
        pool.update({"p%d_x" % i: x, "p%d_y" % i: y, "p%d_g" % i: g, "p%d_d" % i: d,
                     "p%d_geom" % i: np.array([N, C, S, size, start, stride, outX, int(kind == "max")])})
    save("ops_pool.npz", **pool)
    # response norms (layers-18pct rnorm size 3; imagenet cmrnorm size 5, k = 2 here to exercise k)
    C, S = 16, 5
    x = (rng.standard_normal((C * S * S, N)) * 3).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    y, den = G.rnorm(x, N, C, S, 3, 0.001 / 9, 0.75)
    acts = y.copy()
    d = G.rnorm_undo(g, den, x, acts, N, C, S, 3, 0.001 / 9, 0.75)
    y2, den2 = G.cmrnorm(x, N, C, S, 5, 0.0001 / 5, 0.75, False, 2.0)
    acts2 = y2.copy()
    d2 = G.cmrnorm_undo(g, den2, x, acts2, N, C, S, 5, 0.0001 / 5, 0.75, False)
    save("ops_norm.npz", x=x, g=g, rnorm_y=y, rnorm_den=den, rnorm_dx=d, rnorm_acts_after=acts,
         cmr_y=y2, cmr_den=den2, cmr_dx=d2, geom=np.array([N, C, S]))
    # softmax + logreg (ties included: two equal maxima in column 0) + neurons + SGD update + cccp
    K = 10
    logits = (rng.standard_normal((K, N)) * 3).astype(np.float32)
    logits[3, 0] = logits[7, 0] = logits[:, 0].max() + 1
    labels = rng.integers(0, K, size=(1, N)).astype(np.float32)
    labels[0, 0] = 3
    p = G.softmax(logits)
    cost = np.array(G.logreg_cost(labels, p))
    grad = G.logreg_softmax_grad(labels, p, 0.1)
    xs = (rng.standard_normal(257) * 2).astype(np.float32)
    gs = rng.standard_normal(257).astype(np.float32)
    neu = {}
    for kind, prm in (("relu", {}), ("tanh", {"a": 1.79, "b": -0.66}), ("logistic", {}), ("abs", {}), ("linear", {"a": 3.0, "b": 2.2})):
        yv = G.neuron_fwd(kind, prm, xs)
        neu[kind + "_y"] = yv
        neu[kind + "_dx"] = G.neuron_bwd(kind, prm, gs, yv, xs)
    mask = rng.random(257, dtype=np.float32)
    neu["dropout_y"] = G.neuron_fwd("dropout", {"prob": 0.5}, xs, train=True, mask=mask)
    neu["dropout_test_y"] = G.neuron_fwd("dropout", {"prob": 0.5}, xs, train=False)
    w, inc, gr = (rng.standard_normal(300).astype(np.float32) for _ in range(3))
    w2, inc2 = w.copy(), inc.copy()
    G.sgd_update(w2, inc2, gr, 0.001, 0.004, 0.9)
    Wc = rng.standard_normal((7, 9)).astype(np.float32)           # cccp: out = W(o x i) . x(i x P*N)
    xc = rng.standard_normal((9, 5 * N)).astype(np.float32)
    save("ops_misc.npz", logits=logits, labels=labels, probs=p, cost=cost, grad=grad, xs=xs, gs=gs, mask=mask,
         w=w, inc=inc, gr=gr, w_after=w2, inc_after=inc2, cccp_w=Wc, cccp_x=xc, cccp_y=G.matmul(Wc, xc), **neu)


if __name__ == "__main__":
    datagen()
    datagen_group_sparsity()
    ops()
