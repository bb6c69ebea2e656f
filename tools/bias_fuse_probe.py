"""Times mc_gemm vs mc_gemm_row_sums (and the conv pair) on the NIN weight-gradient shapes; run under rocprofv3 --kernel-trace
--stats to see which kernel the difference is in."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])

def ptr(t): return t.data_ptr()
dev = "cuda"
for (mm, nn, kk) in ((160, 192, 131072), (96, 160, 131072), (192, 192, 32768)):
    A = torch.randn(mm, kk, device=dev); B = torch.randn(nn, kk, device=dev); C = torch.empty(mm, nn, device=dev); s = torch.empty(mm, device=dev)
    wsb = m.lib().mc_gemm_row_sums_workspace_bytes(mm, nn, kk)
    ws = torch.empty(max(1, wsb // 4), device=dev)
    wsb0 = m.lib().mc_gemm_workspace_bytes(mm, nn, kk)
    for name in ("plain", "sums"):
        def run():
            if name == "plain":
                m.call("mc_gemm", ptr(A), ptr(B), ptr(C), mm, nn, kk, 0, 1, kk, kk, nn, 0.0, 1.0, None, m.NEURON_NONE, ptr(ws), wsb0, None)
            else:
                m.call("mc_gemm_row_sums", ptr(A), ptr(B), ptr(C), mm, nn, kk, kk, kk, nn, 0.0, 1.0, ptr(s), 1.0, ptr(ws), wsb, None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print("%4d x %4d x %6d  %-6s %.1f us" % (mm, nn, kk, name, e0.elapsed_time(e1) * 1000 / 20))
