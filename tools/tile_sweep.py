"""Times every contraction of the NIN step (N = 128) under a forced wave tile; run once per tile:
   MC_IGEMM_TILE=3,1 python tools/tile_sweep.py [chunks...]   (prints 'shape tile ms')"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
def t_ms(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr(); N = 128
tile = os.environ.get("MC_IGEMM_TILE", "auto"); ch = os.environ.get("MC_GEMM_CHUNKS", "auto")
only = sys.argv[1] if len(sys.argv) > 1 else ""
def out(name, ms, fl): print("%-18s tile=%-5s chunks=%-5s %8.4f ms %7.1f TF/s" % (name, tile, ch, ms, fl / ms / 1e9), flush=True)
def conv(name, C, H, fs, F, pad, pss):
    M = H + 2 * pad - fs + 1
    img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
    o, din = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda")
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1); fl = 2.0 * C*fs*fs * F * M*M * N
    if only in ("", "f"): out(name + "/fwd", t_ms(lambda: m.call("mc_conv_filter_acts", P(img), P(W), P(o), *geo, 0.0, 1.0, None)), fl)
    if C > 3 and only in ("", "b"): out(name + "/bwd", t_ms(lambda: m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None)), fl)
    if only in ("", "w"):
        for ps in pss:
            tmp = torch.empty((M*M//ps)*C*fs*fs, F, device="cuda"); dw = torch.empty(C*fs*fs, F, device="cuda")
            def wg():
                m.call("mc_conv_weight_acts", P(img), P(hid), P(tmp), *geo, ps, 0.0, 1.0, None)
                m.call("mc_sum_cols", P(tmp), P(dw), M*M//ps, C*fs*fs*F, 0.0, 1.0, None)
            out(name + "/wgrad ps=%d" % ps, t_ms(wg), fl)
def cccp(name, i, o, P_):
    W, x, g = R(o, i), R(i, P_*N), R(o, P_*N)
    y, dx, dW = torch.empty(o, P_*N, device="cuda"), torch.empty(i, P_*N, device="cuda"), torch.empty(o, i, device="cuda")
    fl = 2.0 * i * o * P_ * N
    if only in ("", "f"): out(name + "/fwd", t_ms(lambda: m.call("mc_gemm", P(W), P(x), P(y), o, P_*N, i, 0, 0, i, P_*N, P_*N, 0.0, 1.0, None, 0, None, 0, None)), fl)
    if only in ("", "b"): out(name + "/bwd", t_ms(lambda: m.call("mc_gemm", P(W), P(g), P(dx), i, P_*N, o, 1, 0, i, P_*N, P_*N, 0.0, 1.0, None, 0, None, 0, None)), fl)
    if only in ("", "w"):
        wsb = m.lib().mc_gemm_workspace_bytes(o, i, P_*N); ws = torch.empty(max(1, wsb//4), device="cuda")
        out(name + "/wgrad", t_ms(lambda: m.call("mc_gemm", P(g), P(x), P(dW), o, i, P_*N, 0, 1, P_*N, P_*N, i, 0.0, 1.0, None, 0, P(ws), wsb, None)), fl)
conv("conv1", 3, 32, 5, 192, 2, (16, 32, 64))
conv("conv2", 96, 16, 5, 192, 2, (4, 8, 16))
conv("conv3", 192, 8, 3, 192, 1, (2, 4, 8))
cccp("cccp1", 192, 160, 1024); cccp("cccp2", 160, 96, 1024); cccp("cccp3", 192, 192, 256)
cccp("cccp5", 192, 192, 64); cccp("cccp6", 192, 10, 64)
