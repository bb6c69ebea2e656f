"""Child of test_tile_choice_never_changes_a_bit: runs one conv forward / backward-data / weight-gradient triple and two dense
products through the C-ABI under whatever MC_IGEMM_TILE / MC_IGEMM_AUTOTUNE the parent set, and saves the raw outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mckernels as m  # noqa: E402
from gpu_util import dev, host, nans, ptr  # noqa: E402

rng = np.random.default_rng(5)
N, C, H, fs, F, pad = 64, 32, 8, 3, 64, 1
M = H
geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
img, W, hid = dev(rng.standard_normal((C * H * H, N))), dev(rng.standard_normal((C * fs * fs, F))), dev(rng.standard_normal((F * M * M, N)))
b = dev(rng.standard_normal((F,)))
res = {}
out = nans(F * M * M, N)
m.call("mc_conv_filter_acts_fused", ptr(img), ptr(W), ptr(out), ptr(b), m.NEURON_RELU, *geo, 0.0, 1.0, None)
res["fwd"] = host(out)
din = nans(C * H * H, N)
m.call("mc_conv_img_acts", ptr(hid), ptr(W), ptr(din), *geo, 0.0, 1.0, None)
res["bwd"] = host(din)
dw = nans(C * fs * fs, F)
m.call("mc_conv_weight_acts", ptr(img), ptr(hid), ptr(dw), *geo, 0, 0.0, 1.0, None)   # one chunk: the chunk plan is not under test
res["wgrad"] = host(dw)
# dense: forward orientation with bias + relu, backward orientation with a float mask (the general epilogue) and with bits
mm, nn, kk = 160, 1024, 192
A, B, bias = dev(rng.standard_normal((mm, kk))), dev(rng.standard_normal((kk, nn))), dev(rng.standard_normal((mm,)))
y = nans(mm, nn)
m.call("mc_gemm", ptr(A), ptr(B), ptr(y), mm, nn, kk, 0, 0, kk, nn, nn, 0.0, 1.0, ptr(bias), m.NEURON_RELU, None, 0, None)
res["dense_fwd"] = host(y)
import torch
bits = torch.zeros(m.lib().mc_relu_bits_words(mm, nn), dtype=torch.int32, device="cuda")
m.call("mc_relu_bits_from_acts", ptr(y), ptr(bits), mm, nn, nn, None)
V, g = dev(rng.standard_normal((96, mm))), dev(rng.standard_normal((96, nn)))
d1, d2 = nans(mm, nn), nans(mm, nn)
m.call("mc_gemm_relu_grad", ptr(V), ptr(g), ptr(d1), mm, nn, 96, 1, 0, mm, nn, nn, 0.0, 1.0, ptr(y), nn, None, 0, None)
m.call("mc_gemm_relu_grad_bits", ptr(V), ptr(g), ptr(d2), mm, nn, 96, 1, 0, mm, nn, nn, 1.0, ptr(bits), None)
res["dense_bwd_mask"], res["dense_bwd_bits"] = host(d1), host(d2)
# weight-gradient orientation (both operands k-contiguous: the products the 16x16x4 tiles exist for), split K with the shape's own
# chunk plan and the bias gradient as a by-product: a dense one whose whole output is one 160x192 tile, one of 96x160, and a
# first-layer conv weight gradient (75 rows = 3 channels x 5x5 taps, 192 filters)
for tag, (wm_, wn_, wk_) in (("wg160x192", (160, 192, 4096)), ("wg96x160", (96, 160, 3072)), ("wg192x192", (192, 192, 2048)), ("wg10x192", (10, 192, 2048))):
    Ag, Bg = dev(rng.standard_normal((wm_, wk_))), dev(rng.standard_normal((wn_, wk_)))
    wsb = m.lib().mc_gemm_row_sums_workspace_bytes(wm_, wn_, wk_)
    ws, cg_, rs_ = nans(max(1, wsb // 4)), nans(wm_, wn_), nans(wm_)
    m.call("mc_gemm_row_sums", ptr(Ag), ptr(Bg), ptr(cg_), wm_, wn_, wk_, wk_, wk_, wn_, 0.0, 0.5, ptr(rs_), 0.25, ptr(ws), wsb, None)
    res[tag], res[tag + "_sums"] = host(cg_), host(rs_)
N1, C1, H1, fs1, F1, pad1 = 32, 3, 12, 5, 192, 2
geo1 = (N1, C1, H1, H1, F1, fs1, -pad1, 1, H1, H1, 1)
img1, hid1 = dev(rng.standard_normal((C1 * H1 * H1, N1))), dev(rng.standard_normal((F1 * H1 * H1, N1)))
wsb = m.lib().mc_conv_weight_grad_bias_workspace_bytes(N1, C1, F1, fs1, H1 * H1, 1)
ws, dw1, db1 = nans(max(1, wsb // 4)), nans(C1 * fs1 * fs1, F1), nans(F1)
m.call("mc_conv_weight_grad_bias", ptr(img1), ptr(hid1), ptr(dw1), ptr(db1), *geo1, 0.0, 1.0, 0.5, ptr(ws), wsb, None)
res["conv1_wgrad"], res["conv1_bgrad"] = host(dw1), host(db1)
np.savez(sys.argv[1], **res)
