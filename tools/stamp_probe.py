"""Diagnostic (MC_STAMP build): per-K-step cycle shares of consumer wave 0 and producer wave 4.
   MC_KERNELS_LIB=build/variants/libmck_stamp.so python tools/stamp_probe.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import numpy as np, torch
import mckernels as m
m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr(); N = 128
def report(name, nblocks):
    torch.cuda.synchronize()
    buf = np.zeros(nblocks * 8, dtype=np.uint64)
    L = m.lib(); L.mc_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.mc_debug_read_stamps(buf.ctypes.data, nblocks * 8) == 0
    raw = buf.reshape(-1, 8)
    rt = (raw[:, 2] >> np.uint64(32)).astype(np.float64)          # loop duration in 100 MHz ticks
    raw[:, 2] = raw[:, 2] & np.uint64(0xFFFFFFFF)
    b = raw.astype(np.float64)
    print("   shader clock during the K loop: %.2f GHz (median over workgroups)" % float(np.median(b[:, 0] / rt * 0.1)))
    nk = b[:, 3].mean()
    print("%-26s blocks %4d nk %4d | consumer/K-step: total %6.0f  mfma-section %6.0f  barrier-wait %6.0f | producer/K-step: commit(+load wait) %6.0f  fetch-issue %6.0f  lds-drain %5.0f  barrier-wait %6.0f"
          % (name, nblocks, nk, (b[:, 0] / b[:, 3]).mean(), (b[:, 1] / b[:, 3]).mean(), (b[:, 2] / b[:, 3]).mean(),
             (b[:, 4] / b[:, 3]).mean(), (b[:, 5] / b[:, 3]).mean(), (b[:, 6] / b[:, 3]).mean(), (b[:, 7] / b[:, 3]).mean()))
C, H, fs, F, pad = 96, 16, 5, 192, 2; M = 16
img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
out, din = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda")
geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
for _ in range(3): m.call("mc_conv_filter_acts", P(img), P(W), P(out), *geo, 0.0, 1.0, None)
report("conv2 fwd", 512)
for _ in range(3): m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None)
report("conv2 bwd-data", 256)
ps = m.lib().mc_conv_weight_acts_pick_partial_sum(N, C, F, fs, M * M, 1)
tmp = torch.empty((M*M//ps)*C*fs*fs, F, device="cuda")
for _ in range(3): m.call("mc_conv_weight_acts", P(img), P(hid), P(tmp), *geo, ps, 0.0, 1.0, None)
report("conv2 wgrad ps=%d" % ps, 512)
C3, H3, M3 = 192, 8, 8
img3, W3, hid3 = R(C3*H3*H3, N), R(C3*9, 192), R(192*M3*M3, N)
out3, din3 = torch.empty(192*M3*M3, N, device="cuda"), torch.empty(C3*H3*H3, N, device="cuda")
geo3 = (N, C3, H3, H3, 192, 3, -1, 1, M3, M3, 1)
for _ in range(3): m.call("mc_conv_filter_acts", P(img3), P(W3), P(out3), *geo3, 0.0, 1.0, None)
report("conv3 fwd", 192)
for _ in range(3): m.call("mc_conv_img_acts", P(hid3), P(W3), P(din3), *geo3, 0.0, 1.0, None)
report("conv3 bwd-data", 192)
Wc, xc, yc = R(160, 192), R(192, 1024 * N), torch.empty(160, 1024 * N, device="cuda")
for _ in range(3): m.call("mc_gemm", P(Wc), P(xc), P(yc), 160, 1024 * N, 192, 0, 0, 192, 1024 * N, 1024 * N, 0.0, 1.0, None, 0, None, 0, None)
report("cccp1 fwd 160x131072x192", 1024)
A, B, Cm = R(2400, 192), R(2400, 32768), torch.empty(192, 32768, device="cuda")
for _ in range(3): m.call("mc_gemm", P(A), P(B), P(Cm), 192, 32768, 2400, 1, 0, 192, 32768, 32768, 0.0, 1.0, None, 0, None, 0, None)
report("gemm 192x32768x2400 tn", 512)
