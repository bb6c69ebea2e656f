"""HBM-bound kernels of the NIN step against a plain device copy of the same byte count:
   python tools/pool_bench.py    (prints us, algorithmic GB/s, fraction of 8 TB/s)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
def t_us(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
R = lambda *s: torch.randn(*s, device="cuda"); P = lambda t: t.data_ptr(); N = 128
def line(name, us, nbytes): print("%-28s %8.1f us %8.0f GB/s  %5.1f%% of 8 TB/s" % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80.0), flush=True)
def copy_ref(nbytes):
    a = torch.empty(nbytes // 8, device="cuda"); b = torch.empty_like(a)
    line("  (device copy, same bytes)", t_us(lambda: b.copy_(a)), nbytes)
def pool(name, C, S, size, start, stride, kind):
    outX = -(-(S - start - size) // stride) + 1
    x = torch.relu(R(C*S*S, N)); y = torch.empty(C*outX*outX, N, device="cuda"); g = R(C*outX*outX, N); d = torch.empty_like(x)
    pk = m.POOL_MAX if kind == "max" else m.POOL_AVG
    nb = 4 * (x.numel() + y.numel())
    line(name + " fwd", t_us(lambda: m.call("mc_conv_local_pool", P(x), P(y), N, C, S, size, start, stride, outX, pk, None)), nb); copy_ref(nb)
    if kind == "max":
        nb = 4 * (2 * x.numel() + 2 * y.numel())
        line(name + " max-undo", t_us(lambda: m.call("mc_conv_local_max_undo", P(x), P(g), P(y), P(d), N, C, S, size, start, stride, outX, 0.0, 1.0, None)), nb); copy_ref(nb)
    else:
        nb = 4 * (x.numel() + y.numel())
        line(name + " avg-undo", t_us(lambda: m.call("mc_conv_local_avg_undo", P(g), P(d), N, C, S, size, start, stride, outX, 0.0, 1.0, None)), nb)
pool("pool1 96x32^2 max3/2", 96, 32, 3, 0, 2, "max")
pool("pool2 192x16^2 max3/2", 192, 16, 3, 0, 2, "max")
pool("poolg 10x8^2 avg8/1", 10, 8, 8, 0, 1, "avg")
pool("80s pool2 32x16^2 avg3/2", 32, 16, 3, 0, 2, "avg")
n = 192 * 32 * 32 * N
a, b, c = R(n), torch.relu(R(n)), torch.empty(n, device="cuda")
line("relu bwd 192x32^2", t_us(lambda: m.call("mc_neuron_input_grad", m.NEURONS["relu"], 0.0, 0.0, P(a), P(b), P(c), 0, n, None)), 12 * n); copy_ref(12 * n)
o = torch.empty(192, device="cuda")
line("bias grad sum_rows 192", t_us(lambda: m.call("mc_sum_rows", P(a), P(o), 192, n // 192, 0.0, 1.0, None)), 4 * n)
