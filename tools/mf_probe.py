"""Times mc_gemm_row_sums (dense weight gradient + bias gradient, split K + reduction) on one shape:
   python tools/mf_probe.py m n k [iters]      (tile / chunk / interleave choices come from the MC_* environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
M, N, K = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
C, rs = torch.empty(M, N, device="cuda"), torch.empty(M, device="cuda")
wsb = m.lib().mc_gemm_row_sums_workspace_bytes(M, N, K)
ws = torch.empty(max(1, wsb // 4), device="cuda")
P = lambda t: t.data_ptr()
f = lambda: m.call("mc_gemm_row_sums", P(A), P(B), P(C), M, N, K, K, K, N, 0.0, 1.0, P(rs), 1.0, P(ws), wsb, None)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
ref = (A.double() @ B.double().t())
err = float((C.double() - ref).abs().max() / ref.pow(2).mean().sqrt())
print("m %d n %d k %d  tile %s chunks %s interleave %s: %.1f us  %.1f TFLOP/s  (max err / rms %.1e)" % (M, N, K, os.environ.get("MC_IGEMM_TILE", "auto"),
      os.environ.get("MC_GEMM_CHUNKS", "rule"), "yes" if os.environ.get("MC_IGEMM_KINTERLEAVE") else "no", ms * 1e3, 2.0 * M * N * K / ms / 1e9, err))
