"""Diagnostic (MC_STAMP build): per-K-step cycle shares of consumer wave 0 and producer wave 4, and the cycles a consumer
spends storing its tiles.   MC_KERNELS_LIB=build/variants/libmck_stamp.so python tools/stamp_probe.py [conv|gemm]"""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import numpy as np, torch
import mckernels as m
m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr(); N = 128
L = m.lib()
def run(name, fn, tu="gemm"):
    rd = getattr(L, "mc_debug_read_stamps_" + tu); rd.argtypes = [ctypes.c_void_p, ctypes.c_int]; clr = getattr(L, "mc_debug_clear_stamps_" + tu)
    for _ in range(3): fn()
    torch.cuda.synchronize(); assert clr() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    assert rd(buf.ctypes.data, 8192 * 8) == 0
    raw = buf.reshape(-1, 8); raw = raw[raw[:, 3] != 0]
    rt = (raw[:, 2] >> np.uint64(32)).astype(np.float64)          # loop duration in 100 MHz ticks
    accS = (raw[:, 3] >> np.uint64(32)).astype(np.float64); nk = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
    raw[:, 2] = raw[:, 2] & np.uint64(0xFFFFFFFF)
    bb = raw.astype(np.float64)
    clk = float(np.median(bb[:, 0] / rt * 0.1))
    print("%-28s %7.1f us  wgs %4d  nk %5.1f  clk %.2f GHz | consumer: loop %7.0f cyc = %5.1f us (per K-step: mfma %5.0f, barrier %4.0f), stores %6.0f cyc | producer/K-step: commit %4.0f fetch %5.0f barrier %5.0f"
          % (name, a.elapsed_time(b) * 1e3, len(raw), nk.mean(), clk, bb[:, 0].mean(), bb[:, 0].mean() / clk / 1e3, (bb[:, 1] / nk).mean(), (bb[:, 2] / nk).mean(),
             accS.mean(), (bb[:, 4] / nk).mean(), (bb[:, 5] / nk).mean(), (bb[:, 7] / nk).mean()))
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
def gemm(mm, nn, kk, ta, tb, name):
    A = R(kk, mm) if ta else R(mm, kk); B = R(nn, kk) if tb else R(kk, nn); C = torch.empty(mm, nn, device="cuda")
    wsb = L.mc_gemm_workspace_bytes(mm, nn, kk); ws = torch.empty(max(1, wsb // 4), device="cuda")
    run(name, lambda: m.call("mc_gemm", P(A), P(B), P(C), mm, nn, kk, ta, tb, A.shape[1], B.shape[1], nn, 0.0, 1.0, None, 0, P(ws), wsb, None))
def conv(name, C, H, fs, F, pad, which="fbw"):
    M = H + 2 * pad - fs + 1
    img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
    out, din = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda")
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
    if "f" in which: run(name + " fwd", lambda: m.call("mc_conv_filter_acts", P(img), P(W), P(out), *geo, 0.0, 1.0, None), "fwd")
    if "b" in which: run(name + " bwd", lambda: m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None), "bwd")
    if "w" in which:
        wsb = L.mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, 1); ws = torch.empty(max(1, wsb // 4), device="cuda"); dw = torch.empty(C*fs*fs, F, device="cuda")
        run(name + " wgrad", lambda: m.call("mc_conv_weight_grad", P(img), P(hid), P(dw), *geo, 0.0, 1.0, P(ws), wsb, None), "wgrad")
if which == "gemm":
    gemm(160, 131072, 192, 0, 0, "cccp1 fwd"); gemm(192, 131072, 160, 1, 0, "cccp1 bwd"); gemm(160, 192, 131072, 0, 1, "cccp1 wgrad")
    gemm(96, 131072, 160, 0, 0, "cccp2 fwd"); gemm(160, 131072, 96, 1, 0, "cccp2 bwd"); gemm(96, 160, 131072, 0, 1, "cccp2 wgrad")
    gemm(192, 32768, 192, 0, 0, "cccp3 fwd"); gemm(192, 32768, 192, 1, 0, "cccp3 bwd"); gemm(192, 192, 32768, 0, 1, "cccp3 wgrad")
    gemm(192, 8192, 192, 0, 0, "cccp5 fwd"); gemm(192, 8192, 192, 1, 0, "cccp5 bwd"); gemm(192, 192, 8192, 0, 1, "cccp5 wgrad")
elif which == "small":   # layers-80sec: 3 -> 32 on 32x32, 32 -> 32 on 16x16, 32 -> 64 on 8x8, all 5x5 pad 2
    conv("80s conv1", 3, 32, 5, 32, 2, "fw"); conv("80s conv2", 32, 16, 5, 32, 2); conv("80s conv3", 32, 8, 5, 64, 2)
else:
    conv("conv1", 3, 32, 5, 192, 2, "fw"); conv("conv2", 96, 16, 5, 192, 2); conv("conv3", 192, 8, 3, 192, 1)
