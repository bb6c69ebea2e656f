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
runtime's compute stream.  An extra, untimed pass with every launch bracketed by events gives
`roofline_worst` (the contraction furthest below the fp32-MFMA peak among ALL layers' products, the six small
launches of cccp5 / cccp6 counted as one row), `roofline_hbm` (the HBM-bound kernel that takes the most time, bytes per SURVEY 8d over its
in-net event time) and `kernels` (every launch over 1 % of the step with its fraction of its roofline).
`cpu_baseline` is the CPU oracle (oracle/golden_net.py, OpenMP C + NumPy) timed on one 128-image step
on this host's cores (rank 0, N = 1 only).
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

    libkernel.setProfile("conv2/fwd:mc_conv_filter_acts_relu_bits")
    libkernel.getProfile()                                          # clears the communicator's counters: they cover the timed steps only
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

    # full per-kernel table (extra, untimed pass; every rank takes part in its collectives)
    libkernel.setProfile("*")
    libkernel.startResident(0, 0, 10)
    libkernel.finishBatch()
    table = libkernel.getProfile()
    libkernel.setProfile("")
    # counters that ride in the same dict are not kernels
    counters = dict((k, prof.pop(k)) for k in list(prof) if k.startswith(("dp/", "net/step_")))
    for k in [k for k in table if k.startswith(("dp/", "net/step_"))]:
        table.pop(k)

    if rank != 0:
        return
    import rooflines
    import mckernels
    rows = rooflines.attribute(table, rooflines.step_work(layers, MB))
    if a.kernel_table:
        alls = sorted(((k, n, ms / n) for k, (n, ms) in table.items()), key=lambda r: -r[2] * r[1])
        byk = dict((r["kernel"], r) for r in rows)
        with open(a.kernel_table, "w") as f:
            json.dump([dict({"kernel": k, "launches_per_step": n / 10.0, "avg_ms": ms},
                            **({"bound": byk[k]["bound"], "achieved": byk[k]["achieved"], "unit": byk[k]["unit"], "frac_of_peak": byk[k]["frac"]} if k in byk else {}))
                       for k, n, ms in alls], f, indent=1)
    step_us = dt / a.steps * 1e6
    brief = lambda r: {"kernel": r["kernel"], "bound": r["bound"], "achieved": round(r["achieved"], 2), "peak": r["peak"], "unit": r["unit"],
                       "frac": round(r["frac"], 4), "avg_launch_us": round(r["avg_us"], 2), "share_of_step": round(r["avg_us"] / step_us, 4)}
    big = [r for r in rows if r["avg_us"] >= 0.01 * step_us]
    # the weakest contraction of ANY named layer.  The six launches of the last mlpconv block's 1x1 layers (cccp5 / cccp6: 8 192
    # columns, 0.6 and 0.03 GFLOP a product) are a few microseconds of pipeline fill and latency each; they count as one row, their
    # flops over their summed time, so that they are neither hidden nor allowed to stand for the kernels
    mf = [r for r in rows if r["bound"] == "mfma" and not r["kernel"].startswith(("cccp5/", "cccp6/"))]
    tail = [r for r in rows if r["bound"] == "mfma" and r["kernel"].startswith(("cccp5/", "cccp6/"))]
    if tail:
        us, fl = sum(r["avg_us"] for r in tail), sum(r["work"] for r in tail)
        mf.append({"kernel": "cccp5+cccp6 (%d launches)" % len(tail), "bound": "mfma", "avg_us": us, "work": fl, "achieved": fl / (us * 1e-6) / 1e12,
                   "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / (us * 1e-6) / FP32_MFMA_PEAK})
    hb = [r for r in big if r["bound"] == "hbm"]
    worst = min(mf, key=lambda r: r["frac"]) if mf else None
    hbm_dom = max(hb, key=lambda r: r["avg_us"]) if hb else None

    # HBM traffic of the conv2 forward launch: PMC counters cannot be read from inside this process, so the value is
    # the one measured by the separate rocprofv3 --pmc passes (tools/profile_round.sh) and committed under profiles/
    # with the counter summaries it comes from.  It is only reported while the kernel this run timed IS the kernel
    # those passes saw: the library describes the instantiation + grid it launches for conv2's shape right now
    # (mc_describe_conv_filter_acts: the autotuner's choice of this process) and that must equal the recorded one.
    traffic, traffic_note = None, None
    buf = __import__("ctypes").create_string_buffer(256)
    mckernels.call("mc_describe_conv_filter_acts", buf, 256, 1, MB, 96, 16, 16, 192, 5, -2, 1, 16, 16, 1)
    launched = buf.value.decode()
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        known = pmc.get("launches", {pmc.get("launch"): pmc})      # one entry per launch the autotuner may pick
        if launched in known:
            traffic = known[launched]["hbm_bytes_per_launch"]
        else:
            traffic_note = "STALE profiles/pmc_traffic.json: counters were collected on %s, this run launched [%s]" % (sorted(str(k) for k in known), launched)
            sys.stderr.write("bench.py: " + traffic_note + " -- roofline.traffic withheld; re-run tools/profile_round.sh\n")
    except (OSError, KeyError, ValueError) as e:
        traffic_note = "profiles/pmc_traffic.json unreadable: %s" % e
    # the same counters for the other two kernels this line names (roofline_worst, roofline_hbm): profiles/pmc_traffic.json "kernels"
    # holds, per profile key, the bytes of the launch the single-stream PMC passes saw and how long it ran there; they are reported
    # only while this run's event time for the key agrees with the event time of that single-stream run within 20 % (another tile or kernel would not)
    def key_traffic(row):
        try:
            e = pmc["kernels"][row["kernel"]]
        except (NameError, KeyError, TypeError):
            return {"traffic": None}
        if abs(e["event_avg_us_single_stream"] - row["avg_us"]) > 0.2 * row["avg_us"]:
            return {"traffic": None, "traffic_note": "profiles/pmc_traffic.json timed %s at %.1f us (events), this run %.1f us: withheld" % (e["launch"], e["event_avg_us_single_stream"], row["avg_us"])}
        return {"traffic": e["hbm_bytes_per_launch"], "algorithmic_bytes": e.get("algorithmic_bytes_per_launch"), "launch": e["launch"]}

    n_launch, total_ms = prof.get("conv2/fwd:mc_conv_filter_acts_relu_bits", (0, 0.0))
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
                     "algorithmic_bytes": CONV2_BYTES, "avg_launch_ms": avg_ms, "launches_timed": n_launch, "launch": launched},
        "roofline_worst": dict(brief(worst), among="the contractions of every layer; cccp5 + cccp6 (six launches) as one row", **key_traffic(worst)) if worst else None,
        "roofline_hbm": dict(brief(hbm_dom), frac_of_copy_ceiling=round(hbm_dom["achieved"] * 1e9 / rooflines.HBM_COPY, 4), **key_traffic(hbm_dom)) if hbm_dom else None,
        "kernels": [brief(r) for r in sorted(big, key=lambda r: -r["avg_us"])],
    }
    if traffic_note:
        out["roofline"]["traffic_note"] = traffic_note
    if "dp/allreduce:bytes" in counters:   # gradient all-reduce of the timed steps (rank 0's communicator)
        calls, nbytes = counters["dp/allreduce:bytes"]
        wsteps, wait_ms = counters.get("dp/wait:ms", (0, 0.0))
        out["dp"] = {"allreduce_calls_per_step": calls / float(a.steps), "allreduce_bytes_per_step": nbytes / float(a.steps),
                     "exposed_wait_ms_per_step": wait_ms / max(wsteps, 1),
                     "note": "buckets of >= 1 MiB of completed gradients reduced on a side stream during back-propagation; the wait is how long the "
                             "compute stream stalled on them before the weight update (HIP events)"}
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
