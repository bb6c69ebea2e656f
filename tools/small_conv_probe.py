"""Times the conv forward / backward-data / weight-gradient products of the small BASELINE nets (layers-80sec, conv-local) through
the C-ABI; MC_KERNELS_LIB selects a variant build (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
L = m.lib(); N = 128
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
def conv(name, C, H, fs, F, pad):
    M = H + 2 * pad - fs + 1
    img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
    out, din, dw = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda"), torch.empty(C*fs*fs, F, device="cuda")
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
    wsb = L.mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, 1); ws = torch.empty(max(1, wsb // 4), device="cuda")
    f = timeit(lambda: m.call("mc_conv_filter_acts", P(img), P(W), P(out), *geo, 0.0, 1.0, None))
    b = timeit(lambda: m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None))
    w = timeit(lambda: m.call("mc_conv_weight_grad", P(img), P(hid), P(dw), *geo, 0.0, 1.0, P(ws), wsb, None))
    fl = 2.0 * C * fs * fs * F * M * M * N
    print("%-16s fwd %6.1f us (%4.1f%%)  bwd %6.1f us (%4.1f%%)  wgrad %6.1f us (%4.1f%%)" % (name, f, fl / f / 157.3e4, b, fl / b / 157.3e4, w, fl / w / 157.3e4))
conv("80s conv1", 3, 32, 5, 32, 2); conv("80s conv2", 32, 16, 5, 32, 2); conv("80s conv3", 32, 8, 5, 64, 2)
conv("cl conv1", 3, 32, 5, 64, 2); conv("cl conv2", 64, 16, 5, 64, 2)
