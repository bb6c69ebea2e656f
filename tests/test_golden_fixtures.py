"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

datagen_*: the reference's own unit-test cases (Kernel/unittest/*/DataGen.m) with expected values from
scipy.signal -- independent of oracle/ -- checked against BOTH the CPU oracle (not gpu) and the HIP kernels
through the C-ABI (gpu).  ops_*: regression pins of the oracle for the ops no reference data covers; the
HIP kernels are checked against the stored values on the GPU.
Tolerances: contractions rel 1e-4 of the tensor RMS (north_star); routing / relu / dropout bit-exact."""
import os

import numpy as np
import pytest

from oracle import golden as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def relerr(got, exp):
    exp = exp.astype(np.float64)
    return float(np.abs(got.astype(np.float64) - exp).max() / max(float(np.sqrt((exp ** 2).mean())), 1e-30))


# ------------------------------------------------------------------ CPU: oracle vs fixtures
def test_oracle_matches_datagen_vectors():
    d = load("datagen_filter_acts.npz")
    geo = [int(v) for v in d["geom"]]
    assert relerr(G.filter_acts(d["images"], d["filters"], *geo), d["expected"]) < 1e-5
    d = load("datagen_img_acts.npz")
    geo = [int(v) for v in d["geom"]]
    assert relerr(G.img_acts(d["hidacts"], d["filters"], *geo), d["expected"]) < 1e-5
    d = load("datagen_weight_acts.npz")
    geo = [int(v) for v in d["geom"]]
    M = geo[8] * geo[9]
    part = G.weight_acts(d["images"], d["hidacts"], *geo, partial_sum=1)      # test.cu:62-66: partialSum 1, then sum(0)
    assert relerr(part.reshape(M, -1, geo[4]).sum(axis=0), d["expected"]) < 1e-5


def test_oracle_regression_pins():
    d = load("ops_pool.npz")
    for i in range(4):
        N, C, S, size, start, stride, outX, is_max = (int(v) for v in d["p%d_geom" % i])
        kind = "max" if is_max else "avg"
        y = G.pool(d["p%d_x" % i], N, C, S, size, start, stride, outX, kind)
        assert np.array_equal(y, d["p%d_y" % i])
        if is_max:
            u = G.max_undo(d["p%d_x" % i], d["p%d_g" % i], y, N, C, S, size, start, stride, outX)
        else:
            u = G.avg_undo(d["p%d_g" % i], N, C, S, size, start, stride, outX)
        assert np.array_equal(u, d["p%d_d" % i])
    d = load("ops_misc.npz")
    p = G.softmax(d["logits"])
    np.testing.assert_allclose(p, d["probs"], rtol=1e-6)
    np.testing.assert_allclose(G.logreg_cost(d["labels"], p), d["cost"], rtol=1e-6)


# ------------------------------------------------------------------ GPU: HIP kernels vs fixtures
def _gpu():
    import mckernels
    from gpu_util import dev, host, nans, ptr, keep
    return mckernels, dev, host, nans, ptr, keep


@pytest.mark.gpu
def test_hip_conv_matches_datagen_vectors():
    m, dev, host, nans, ptr, keep = _gpu()
    d = load("datagen_filter_acts.npz")
    geo = [int(v) for v in d["geom"]]
    out = nans(*d["expected"].shape)
    m.call("mc_conv_filter_acts", ptr(keep(dev(d["images"]))), ptr(keep(dev(d["filters"]))), ptr(out), *geo, 0.0, 1.0, None)
    assert relerr(host(out), d["expected"]) < TOL
    d = load("datagen_img_acts.npz")
    geo = [int(v) for v in d["geom"]]
    out = nans(*d["expected"].shape)
    m.call("mc_conv_img_acts", ptr(keep(dev(d["hidacts"]))), ptr(keep(dev(d["filters"]))), ptr(out), *geo, 0.0, 1.0, None)
    assert relerr(host(out), d["expected"]) < TOL
    d = load("datagen_weight_acts.npz")
    geo = [int(v) for v in d["geom"]]
    M, F = geo[8] * geo[9], geo[4]
    rows = d["expected"].shape[0]
    part = nans(M * rows, F)
    m.call("mc_conv_weight_acts", ptr(keep(dev(d["images"]))), ptr(keep(dev(d["hidacts"]))), ptr(part), *geo, 1, 0.0, 1.0, None)
    tot = nans(rows * F)
    m.call("mc_sum_cols", ptr(part), ptr(tot), M, rows * F, 0.0, 1.0, None)     # the reference's sum(0) over modules
    assert relerr(host(tot).reshape(rows, F), d["expected"]) < TOL


@pytest.mark.gpu
def test_hip_pool_norm_cost_match_pins():
    m, dev, host, nans, ptr, keep = _gpu()
    d = load("ops_pool.npz")
    for i in range(4):
        N, C, S, size, start, stride, outX, is_max = (int(v) for v in d["p%d_geom" % i])
        x, g = keep(dev(d["p%d_x" % i])), keep(dev(d["p%d_g" % i]))
        y = nans(*d["p%d_y" % i].shape)
        m.call("mc_conv_local_pool", ptr(x), ptr(y), N, C, S, size, start, stride, outX, m.POOL_MAX if is_max else m.POOL_AVG, None)
        if is_max:
            assert np.array_equal(host(y), d["p%d_y" % i])
        else:
            np.testing.assert_allclose(host(y), d["p%d_y" % i], rtol=1e-6, atol=1e-6)
        t = nans(*d["p%d_d" % i].shape)
        if is_max:
            m.call("mc_conv_local_max_undo", ptr(x), ptr(g), ptr(y), ptr(t), N, C, S, size, start, stride, outX, 0.0, 1.0, None)
            assert np.array_equal(host(t), d["p%d_d" % i])            # equality routing, bit exact
        else:
            m.call("mc_conv_local_avg_undo", ptr(g), ptr(t), N, C, S, size, start, stride, outX, 0.0, 1.0, None)
            np.testing.assert_allclose(host(t), d["p%d_d" % i], rtol=1e-6, atol=1e-6)
    d = load("ops_norm.npz")
    N, C, S = (int(v) for v in d["geom"])
    x, g = keep(dev(d["x"])), keep(dev(d["g"]))
    den, y = nans(*d["x"].shape), nans(*d["x"].shape)
    m.call("mc_conv_response_norm", ptr(x), ptr(den), ptr(y), N, C, S, 3, 0.001 / 9, 0.75, None)
    assert relerr(host(den), d["rnorm_den"]) < 1e-6 and relerr(host(y), d["rnorm_y"]) < TOL
    t = nans(*d["x"].shape)
    m.call("mc_conv_response_norm_undo", ptr(g), ptr(den), ptr(x), ptr(y), ptr(t), N, C, S, 3, 0.001 / 9, 0.75, 0.0, 1.0, None)
    assert relerr(host(t), d["rnorm_dx"]) < TOL and relerr(host(y), d["rnorm_acts_after"]) < TOL
    den, y = nans(*d["x"].shape), nans(*d["x"].shape)
    m.call("mc_conv_response_norm_cross_map", ptr(x), ptr(den), ptr(y), N, C, S, 5, 0.0001 / 5, 0.75, 0, 2.0, None)
    assert relerr(host(den), d["cmr_den"]) < 1e-6 and relerr(host(y), d["cmr_y"]) < TOL
    t = nans(*d["x"].shape)
    m.call("mc_conv_response_norm_cross_map_undo", ptr(g), ptr(den), ptr(x), ptr(y), ptr(t), N, C, S, 5, 0.0001 / 5, 0.75, 0, 0.0, 1.0, None)
    assert relerr(host(t), d["cmr_dx"]) < TOL
    d = load("ops_misc.npz")
    K, N = d["logits"].shape
    lab = keep(dev(d["labels"]))
    p = nans(K, N)
    m.call("mc_softmax", ptr(keep(dev(d["logits"]))), ptr(p), N, K, None)
    np.testing.assert_allclose(host(p), d["probs"], rtol=1e-5, atol=1e-7)
    p = keep(dev(d["probs"]))                      # cost + gradient on the pinned probabilities (tie in column 0)
    lp, cp = nans(N), nans(N)
    m.call("mc_compute_logreg_cost", ptr(lab), ptr(p), ptr(lp), ptr(cp), N, K, None)
    assert abs(-host(lp).astype(np.float64).sum() - d["cost"][0]) < 1e-4 * abs(d["cost"][0])
    assert abs((N - host(cp).astype(np.float64).sum()) - d["cost"][1]) < 1e-6
    gr = nans(K, N)
    m.call("mc_compute_logreg_softmax_grad", ptr(lab), ptr(p), ptr(gr), 0, 0.1, N, K, None)
    np.testing.assert_allclose(host(gr), d["grad"], rtol=1e-6, atol=1e-7)
    n = d["xs"].size
    xs, gs = keep(dev(d["xs"])), keep(dev(d["gs"]))
    for kind, a, b, exact in (("relu", 0, 0, True), ("tanh", 1.79, -0.66, False), ("logistic", 0, 0, False), ("abs", 0, 0, True), ("linear", 3.0, 2.2, False)):
        yv = nans(n)
        m.call("mc_neuron_activate", m.NEURONS[kind], a, b, ptr(xs), ptr(yv), n, None)
        if exact:
            assert np.array_equal(host(yv), d[kind + "_y"])
        else:
            np.testing.assert_allclose(host(yv), d[kind + "_y"], rtol=2e-6, atol=1e-6)
        t = nans(n)
        m.call("mc_neuron_input_grad", m.NEURONS[kind], a, b, ptr(gs), ptr(xs if kind == "abs" else keep(dev(d[kind + "_y"]))), ptr(t), 0, n, None)
        np.testing.assert_allclose(host(t), d[kind + "_dx"], rtol=1e-5, atol=1e-5)
    yv = nans(n)
    m.call("mc_dropout_apply", ptr(xs), ptr(keep(dev(d["mask"]))), ptr(yv), 0.5, 0, n, None)
    assert np.array_equal(host(yv), d["dropout_y"])
    w, inc = dev(d["w"]), dev(d["inc"])
    m.call("mc_weights_update", ptr(w), ptr(inc), ptr(keep(dev(d["gr"]))), d["w"].size, 0.9, 0.001, 0.004, None)
    np.testing.assert_allclose(host(inc), d["inc_after"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(w), d["w_after"], rtol=2e-6, atol=1e-6)
    o, i = d["cccp_w"].shape
    cols = d["cccp_x"].shape[1]
    yv = nans(o, cols)
    m.call("mc_gemm", ptr(keep(dev(d["cccp_w"]))), ptr(keep(dev(d["cccp_x"]))), ptr(yv), o, cols, i, 0, 0, i, cols, cols, 0.0, 1.0, None, 0, None, 0, None)
    assert relerr(host(yv), d["cccp_y"]) < TOL
