import os, sys
sys.path.insert(0, "/root/repo/cuda-convnet_amd")
import torch
import mckernels as m
def t_ms(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr(); N = 128
for (i, o, P_) in ((192, 160, 1024), (160, 96, 1024), (192, 192, 256)):
    W, x, bias = R(o, i), R(i, P_*N), R(o)
    y = torch.empty(o, P_*N, device="cuda")
    a = t_ms(lambda: m.call("mc_gemm", P(W), P(x), P(y), o, P_*N, i, 0, 0, i, P_*N, P_*N, 0.0, 1.0, None, 0, None, 0, None))
    b = t_ms(lambda: m.call("mc_gemm", P(W), P(x), P(y), o, P_*N, i, 0, 0, i, P_*N, P_*N, 0.0, 1.0, P(bias), 1, None, 0, None))
    c = t_ms(lambda: m.call("mc_gemm", P(W), P(x), P(y), o, P_*N, i, 0, 0, i, P_*N, P_*N, 0.0, 1.0, None, 1, None, 0, None))
    print("cccp %dx%dx%d: plain %.4f  bias+relu %.4f  relu only %.4f ms" % (o, P_*N, i, a, b, c))
