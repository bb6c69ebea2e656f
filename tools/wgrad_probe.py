"""Times mc_conv_weight_grad_bias (conv weight gradient + shared-bias gradient, split K + reduction) on one layer shape:
   python tools/wgrad_probe.py N C H fs F pad [iters]     (tile / chunk choices come from MC_IGEMM_TILE / MC_WGRAD_CHUNKS)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
N, C, H, fs, F, pad = (int(x) for x in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 30
geo = (N, C, H, H, F, fs, -pad, 1, H, H, 1)
img, hid = torch.randn(C * H * H, N, device="cuda"), torch.randn(F * H * H, N, device="cuda")
wsb = m.lib().mc_conv_weight_grad_bias_workspace_bytes(N, C, F, fs, H * H, 1)
ws, dw, db = torch.empty(max(1, wsb // 4), device="cuda"), torch.empty(C * fs * fs, F, device="cuda"), torch.empty(F, device="cuda")
P = lambda t: t.data_ptr()
f = lambda: m.call("mc_conv_weight_grad_bias", P(img), P(hid), P(dw), P(db), *geo, 0.0, 1.0, 1.0, P(ws), wsb, None)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
ref = torch.nn.functional.unfold(img.view(C, H, H, N).permute(3, 0, 1, 2).double(), fs, padding=pad)     # N, C*fs*fs, H*H
want = torch.einsum("nrp,fpn->rf", ref, hid.view(F, H * H, N).double())
err = float((dw.double() - want).abs().max() / want.pow(2).mean().sqrt())
print("N %d C %d H %d fs %d F %d  tile %s chunks %s: %.1f us  %.1f TFLOP/s  (max err / rms %.1e)" % (N, C, H, fs, F, os.environ.get("MC_IGEMM_TILE", "auto"),
      os.environ.get("MC_WGRAD_CHUNKS", "rule"), ms * 1e3, 2.0 * C * fs * fs * F * H * H * N / ms / 1e9, err))
