"""bench.py -- NIN CIFAR-10 training throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full training step (fprop + cost + bprop + SGD update, plus the RCCL gradient
all-reduce when N > 1) of the NIN cifar-10 net on one minibatch of 128 synthetic CIFAR-shaped
images per GPU, through the product path: the `libkernel` extension (C++ runtime) calling the HIP
kernels of libmckernels.so.  Inputs are resident in HBM before the timed region (the batch is
uploaded by the warm-up startBatch; timed steps re-run resident minibatches).  Prints ONE JSON line.

The `roofline` object is the dominant kernel, conv2's forward contraction (30.2 GFLOP of the
167.1 GFLOP step): its launches inside the timed region are bracketed by HIP events on the
runtime's compute stream.  `cpu_baseline` is the CPU oracle (oracle/golden_net.py, OpenMP C +
NumPy) timed on one 128-image step on this host's cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))

MB = 128
NET = "nin"
FP32_MFMA_PEAK = 157.3e12       # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
CONV2_FLOPS = 2.0 * 2400 * 192 * 256 * MB     # 2 * K * F * modules * images
TRAIN_GFLOP_PER_STEP = 167.09
CONV2_BYTES = 4 * ((96 * 256 + 192 * 256) * MB + 2400 * 192)   # input + output + filters, each touched once


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--resident-minibatches", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", default="", help="also write a per-kernel timing table (JSON) here")
    a = ap.parse_args()

    import torch          # first: libkernel must share torch's HIP runtime (same soname)
    import dp
    rank, world, local_rank = dp.env()
    if world != a.gpus:
        if a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = dp.init_control_group() if world > 1 else None

    import modelutil
    import libkernel
    layers, dims, classes = modelutil.load_net(NET, seed=42)       # identical weights on every rank
    libkernel.initModel(layers, MB, local_rank)
    dp.connect(libkernel, rank, world)
    libkernel.setSeed(99 + rank)                                    # dropout masks differ per rank

    nres = a.resident_minibatches
    data = modelutil.synthetic_batch(dims, classes, nres * MB, seed_data=1234 + rank, seed_labels=5678 + rank)
    # upload + one pass over the resident minibatches (counts as warm-up work)
    libkernel.startBatch(data, 0)
    libkernel.finishBatch()
    if a.warmup > 0:
        libkernel.startResident(0, 0, a.warmup)
        libkernel.finishBatch()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    libkernel.setProfile("conv2/fwd:mc_conv_filter_acts_fused")
    barrier()
    t0 = time.perf_counter()
    libkernel.startResident(0, a.warmup % nres, a.steps)
    costs, ncases = libkernel.finishBatch()                         # blocks until the stream is drained
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    prof = libkernel.getProfile()
    libkernel.setProfile("")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # optional full per-kernel table (extra, untimed pass; every rank takes part in its collectives)
    table = None
    if a.kernel_table:
        libkernel.setProfile("*")
        libkernel.startResident(0, 0, 10)
        libkernel.finishBatch()
        table = libkernel.getProfile()
        libkernel.setProfile("")

    if rank != 0:
        return
    if table is not None:
        rows = sorted(((k, n, ms / n) for k, (n, ms) in table.items()), key=lambda r: -r[2] * r[1])
        with open(a.kernel_table, "w") as f:
            json.dump([{"kernel": k, "launches_per_step": n / 10.0, "avg_ms": ms} for k, n, ms in rows], f, indent=1)

    # HBM traffic of the same launch: PMC counters cannot be read from inside this process, so the
    # value is the one measured by the separate rocprofv3 --pmc passes (tools/profile_round.sh) and
    # committed under profiles/ together with the counter summaries it comes from.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    n_launch, total_ms = prof.get("conv2/fwd:mc_conv_filter_acts_fused", (0, 0.0))
    avg_ms = total_ms / max(n_launch, 1)
    achieved = CONV2_FLOPS / (avg_ms * 1e-3) / 1e12 if n_launch else None
    out = {
        "metric": "train images/sec, NIN CIFAR-10 bs128/GPU", "value": world * MB * a.steps / dt, "unit": "images/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "NIN cifar-10 (configs[2]: conv/cccp/pool/dropout/softmax, 966,986 params) full training step, "
                               "128 images per GPU, global batch %d" % (world * MB),
                   "global_batch": world * MB, "parallelism": "dp%d" % world, "gflop_per_step_per_gpu": TRAIN_GFLOP_PER_STEP,
                   "contraction_tflops_per_gpu": TRAIN_GFLOP_PER_STEP / (dt / a.steps) / 1e3,
                   "logprob_per_case": costs["logprob"][0] / max(ncases, 1)},
        "roofline": {"bound": "mfma", "kernel": "conv2 forward (mc_igemm, implicit GEMM 192 x 32768 x 2400)", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": (achieved / (FP32_MFMA_PEAK / 1e12)) if achieved else None,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)",
                     "algorithmic_bytes": CONV2_BYTES, "avg_launch_ms": avg_ms, "launches_timed": n_launch},
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(modelutil)
    print(json.dumps(out))


def cpu_baseline(modelutil):
    """One NIN training step of 128 images on the CPU oracle (the reported baseline, not a target)."""
    sys.path.insert(0, ROOT)
    from oracle.golden_net import GoldenNet
    import numpy as np
    layers, dims, classes = modelutil.load_net(NET, seed=42)
    data = modelutil.synthetic_batch(dims, classes, MB)
    net = GoldenNet(layers)
    rng = np.random.default_rng(99)
    masks = dict((nm, rng.random((outs, MB), dtype=np.float32)) for nm, outs in modelutil.dropout_layers(layers))
    net.fprop(data, masks=masks)            # untimed: page in libgolden.so, spin up the OpenMP team
    steps, t0 = 0, time.perf_counter()
    while steps < 12 and (steps == 0 or time.perf_counter() - t0 < 12.0):
        net.train_step(data, masks=masks)
        steps += 1
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or os.cpu_count()
    return {"value": steps * MB / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d full NIN training steps of 128 synthetic images (%.1f s): OpenMP C contractions/pool/norm + NumPy" % (steps, dt)}


if __name__ == "__main__":
    main()
