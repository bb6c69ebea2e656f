"""Times the classifier tail (softmax, logreg cost + sums, gradient) through the C-ABI for wide classifiers (100 / 1000 classes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch
import mckernels as m
if os.environ.get("MC_KERNELS_LIB"): m._LIB_PATH = os.path.abspath(os.environ["MC_KERNELS_LIB"])
L = m.lib(); P = lambda t: t.data_ptr()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for C, N in ((10, 128), (100, 128), (1000, 128), (1000, 256)):
    x = torch.randn(C, N, device="cuda"); lab = torch.randint(0, C, (1, N), device="cuda").float()
    pr, g = torch.empty_like(x), torch.empty_like(x)
    llp, cp = torch.empty(1, N, device="cuda"), torch.empty(1, N, device="cuda"); sums = torch.zeros(2, device="cuda", dtype=torch.float64)
    t1 = timeit(lambda: m.call("mc_softmax", P(x), P(pr), N, C, None))
    t2 = timeit(lambda: m.call("mc_compute_logreg_cost_sums", P(lab), P(pr), P(llp), P(cp), P(sums), N, C, None))
    t3 = timeit(lambda: m.call("mc_compute_logreg_softmax_grad", P(lab), P(pr), P(g), 0, 1.0, N, C, None))
    ref = torch.softmax(x.double(), 0)
    print("%4d classes x %3d cases: softmax %7.1f us, cost+sums %7.1f us, grad %6.1f us; max rel err of probs %.2e" % (C, N, t1, t2, t3, float(((pr.double() - ref).abs() / ref).max())))
