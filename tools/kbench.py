"""Per-kernel timing of the C-ABI on the NIN / 80sec shapes (N = 128): TFLOP/s vs the 157.3 TF
fp32-MFMA peak for contractions, GB/s vs 8 TB/s for the HBM-bound kernels."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m

def t_ms(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def R(*s): return torch.randn(*s, device="cuda")
P = lambda t: t.data_ptr()
N = 128
rows = []
def conv(name, C, H, fs, F, pad):
    M = H + 2 * pad - fs + 1
    img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
    out, din = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda")
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
    fl = 2.0 * C*fs*fs * F * M*M * N
    ms = t_ms(lambda: m.call("mc_conv_filter_acts", P(img), P(W), P(out), *geo, 0.0, 1.0, None))
    rows.append((name+" fwd", ms, fl/ms/1e9))
    if C > 3:
        ms = t_ms(lambda: m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None))
        rows.append((name+" bwd-data", ms, fl/ms/1e9))
    ps = m.lib().mc_conv_weight_acts_pick_partial_sum(N, C, F, fs, M*M, 1)
    tmp = torch.empty((M*M//ps)*C*fs*fs, F, device="cuda"); dw = torch.empty(C*fs*fs, F, device="cuda")
    def wg():
        m.call("mc_conv_weight_acts", P(img), P(hid), P(tmp), *geo, ps, 0.0, 1.0, None)
        m.call("mc_sum_cols", P(tmp), P(dw), M*M//ps, C*fs*fs*F, 0.0, 1.0, None)
    ms = t_ms(wg)
    rows.append((name+" bwd-wts(ps=%d)" % ps, ms, fl/ms/1e9))
def cccp(name, i, o, P_):
    W, x, g = R(o, i), R(i, P_*N), R(o, P_*N)
    y, dx, dW = torch.empty(o, P_*N, device="cuda"), torch.empty(i, P_*N, device="cuda"), torch.empty(o, i, device="cuda")
    fl = 2.0 * i * o * P_ * N
    ms = t_ms(lambda: m.call("mc_gemm", P(W), P(x), P(y), o, P_*N, i, 0, 0, i, P_*N, P_*N, 0.0, 1.0, None, 0, None, 0, None))
    rows.append((name+" fwd", ms, fl/ms/1e9))
    ms = t_ms(lambda: m.call("mc_gemm", P(W), P(g), P(dx), i, P_*N, o, 1, 0, i, P_*N, P_*N, 0.0, 1.0, None, 0, None, 0, None))
    rows.append((name+" bwd-data", ms, fl/ms/1e9))
    wsb = m.lib().mc_gemm_workspace_bytes(o, i, P_*N); ws = torch.empty(max(1, wsb//4), device="cuda")
    ms = t_ms(lambda: m.call("mc_gemm", P(g), P(x), P(dW), o, i, P_*N, 0, 1, P_*N, P_*N, i, 0.0, 1.0, None, 0, P(ws), wsb, None))
    rows.append((name+" bwd-wts", ms, fl/ms/1e9))
which = sys.argv[1] if len(sys.argv) > 1 else "nin"
if which == "nin":
    conv("conv1 3->192 5x5 32^2", 3, 32, 5, 192, 2)
    conv("conv2 96->192 5x5 16^2", 96, 16, 5, 192, 2)
    conv("conv3 192->192 3x3 8^2", 192, 8, 3, 192, 1)
    cccp("cccp1 192->160 P1024", 192, 160, 1024)
    cccp("cccp2 160->96 P1024", 160, 96, 1024)
    cccp("cccp3 192->192 P256", 192, 192, 256)
    cccp("cccp5 192->192 P64", 192, 192, 64)
    cccp("cccp6 192->10 P64", 192, 10, 64)
else:
    conv("80s conv1 3->32", 3, 32, 5, 32, 2); conv("80s conv2 32->32", 32, 16, 5, 32, 2); conv("80s conv3 32->64", 32, 8, 5, 64, 2)
tot = 0
for n, ms, gf in rows:
    tot += ms
    print("%-34s %8.3f ms %8.1f TF/s  %5.1f%% of 157.3" % (n, ms, gf, gf/1.573))
print("total %.3f ms" % tot)
