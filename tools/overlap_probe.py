"""Do two small conv products overlap on the device when issued on two streams?  Times 2n launches of the layers-80sec conv2
backward-data product on one stream against n launches on each of two streams, and the same for a backward-data /
weight-gradient pair (what a forked step graph would run side by side)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
L = m.lib(); N = 128
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr()
C, H, fs, F, pad = 32, 16, 5, 32, 2
M = H
geo = (N, C, H, H, F, fs, -pad, 1, M, M, 1)
def bufs():
    return R(C*H*H, N), R(C*fs*fs, F), R(F*M*M, N), torch.empty(C*H*H, N, device="cuda"), torch.empty(C*fs*fs, F, device="cuda")
A, B = bufs(), bufs()
wsb = L.mc_conv_weight_grad_workspace_bytes(N, C, F, fs, M * M, 1)
wsA, wsB = torch.empty(max(1, wsb // 4), device="cuda"), torch.empty(max(1, wsb // 4), device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def bwd(b, st): m.call("mc_conv_img_acts", P(b[2]), P(b[1]), P(b[3]), *geo, 0.0, 1.0, st.cuda_stream)
def wg(b, ws, st): m.call("mc_conv_weight_grad", P(b[0]), P(b[2]), P(b[4]), *geo, 0.0, 1.0, P(ws), wsb, st.cuda_stream)
for _ in range(5): bwd(A, s1); wg(A, wsA, s1); bwd(B, s2); wg(B, wsB, s2)
torch.cuda.synchronize()
def wall(fn, n=200):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e6 / n
def one_bwd(n):
    for _ in range(n): bwd(A, s1); bwd(A, s1)
def two_bwd(n):
    for _ in range(n): bwd(A, s1); bwd(B, s2)
def one_mix(n):
    for _ in range(n): bwd(A, s1); wg(A, wsA, s1)
def two_mix(n):
    for _ in range(n): bwd(A, s1); wg(B, wsB, s2)
for name, fn in (("bwd+bwd one stream", one_bwd), ("bwd+bwd two streams", two_bwd), ("bwd+wgrad one stream", one_mix), ("bwd+wgrad two streams", two_mix)):
    fn(50); print("%-24s %7.1f us per pair" % (name, wall(fn)))
