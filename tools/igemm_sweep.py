"""Times the contraction kernels of NIN's conv2 (and a square GEMM) through the C-ABI; the wave tile
can be forced with MC_IGEMM_TILE=MT,NT.  python tools/igemm_sweep.py [what]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])

def t_ms(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

R = lambda *s: torch.randn(*s, device="cuda")
P = lambda t: t.data_ptr()
N = 128
what = sys.argv[1] if len(sys.argv) > 1 else "all"
tile = os.environ.get("MC_IGEMM_TILE", "auto") + ":" + os.path.basename(os.environ.get("MC_KERNELS_LIB", "default"))[:18]
def conv(name, C, H, fs, F, pad, which="fbw"):
    M = H + 2 * pad - fs + 1
    img, W, hid = R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N)
    out, din = torch.empty(F*M*M, N, device="cuda"), torch.empty(C*H*H, N, device="cuda")
    geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
    fl = 2.0 * C*fs*fs * F * M*M * N
    if "f" in which:
        ms = t_ms(lambda: m.call("mc_conv_filter_acts", P(img), P(W), P(out), *geo, 0.0, 1.0, None))
        print("%-22s %-28s fwd      %8.3f ms %7.1f TF/s" % (tile, name, ms, fl/ms/1e9))
    if "b" in which:
        ms = t_ms(lambda: m.call("mc_conv_img_acts", P(hid), P(W), P(din), *geo, 0.0, 1.0, None))
        print("%-22s %-28s bwd-data %8.3f ms %7.1f TF/s" % (tile, name, ms, fl/ms/1e9))
    if "w" in which:
        ps = m.lib().mc_conv_weight_acts_pick_partial_sum(N, C, F, fs, M*M, 1)
        tmp = torch.empty((M*M//ps)*C*fs*fs, F, device="cuda")
        ms = t_ms(lambda: m.call("mc_conv_weight_acts", P(img), P(hid), P(tmp), *geo, ps, 0.0, 1.0, None))
        print("%-22s %-28s bwd-wts(ps=%d) %8.3f ms %7.1f TF/s" % (tile, name, ps, ms, fl/ms/1e9))
def gemm(mm, nn, kk, ta=0, tb=0):
    A = R(kk, mm) if ta else R(mm, kk); B = R(nn, kk) if tb else R(kk, nn); C = torch.empty(mm, nn, device="cuda")
    wsb = m.lib().mc_gemm_workspace_bytes(mm, nn, kk); ws = torch.empty(max(1, wsb // 4), device="cuda")
    ms = t_ms(lambda: m.call("mc_gemm", P(A), P(B), P(C), mm, nn, kk, ta, tb, A.shape[1], B.shape[1], nn, 0.0, 1.0, None, 0, P(ws), wsb, None))
    print("%-22s gemm %dx%dx%d ta=%d tb=%d %8.3f ms %7.1f TF/s" % (tile, mm, nn, kk, ta, tb, ms, 2.0*mm*nn*kk/ms/1e9))
if what in ("all", "conv2"): conv("conv2 96->192 5x5 16^2", 96, 16, 5, 192, 2)
if what in ("all", "conv2f"): conv("conv2 96->192 5x5 16^2", 96, 16, 5, 192, 2, "f")
if what in ("all", "gemm"):
    gemm(4096, 4096, 4096, 1, 0); gemm(4096, 4096, 4096, 0, 0)
    gemm(160, 131072, 192, 0, 0); gemm(192, 131072, 160, 1, 0)
if what in ("all", "cccp"):
    for (i, o, pp) in ((192, 160, 1024), (160, 96, 1024), (192, 192, 256), (192, 192, 64), (192, 10, 64)):
        gemm(o, pp * N, i, 0, 0); gemm(i, pp * N, o, 1, 0); gemm(o, i, pp * N, 0, 1)
if what in ("all", "conv13"):
    conv("conv1 3->192 5x5 32^2", 3, 32, 5, 192, 2, "fw"); conv("conv3 192->192 3x3 8^2", 192, 8, 3, 192, 1)
if what == "exp":   # conv2 beside plain products of the same extents (cost of the gather addressing)
    conv("conv2 96->192 5x5 16^2", 96, 16, 5, 192, 2)
    gemm(192, 32768, 2400, 1, 0); gemm(96, 32768, 4800, 0, 0); gemm(2400, 192, 32768, 0, 1)
    gemm(160, 131072, 192, 0, 0); gemm(192, 131072, 160, 1, 0); gemm(160, 192, 131072, 0, 1)
