"""Algorithmic work of every launch of a training step, keyed like libkernel's profile table ("<layer>/<phase>"), so that
event times become fractions of a roofline: fp32 flops for the contractions (2 K F M N), bytes that
must cross HBM once for the pool / norm / neuron / reduction kernels (4 B x elements read once + written once).
Used by bench.py and tools/net_bench.py; nothing here touches the GPU."""

FP32_MFMA_PEAK = 157.3e12     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12             # HBM3E spec
HBM_COPY = 6.29e12            # measured device-copy ceiling (same guide)


def step_work(layers, mb):
    """{profile-key-prefix: ("mfma", flops) | ("hbm", bytes)} for one minibatch of `mb` cases."""
    by_name = dict((l["name"], l) for l in layers)
    outs = lambda l: l["outputs"]
    work = {}
    for l in layers:
        n, t = l["name"], l["type"]
        ins = [layers[i] for i in l.get("inputs", [])]
        if t in ("conv", "local"):
            fl = 0.0
            for i in range(len(ins)):
                fl += 2.0 * l["filterChannels"][i] * l["filterPixels"][i] * l["filters"] * l["modules"] * mb
            work[n + "/fwd"] = ("mfma", fl)
            work[n + "/bwd"] = ("mfma", fl)
            work[n + "/wgrad:mc_conv_weight_grad"] = ("mfma", fl)
            work[n + "/wgrad:mc_local_weight_acts"] = ("mfma", fl)
            work[n + "/wgrad:mc_sum_rows"] = ("hbm", 4.0 * outs(l) * mb)
        elif t == "fc":
            fl = sum(2.0 * outs(i) * outs(l) * mb for i in ins)
            work[n + "/fwd"] = ("mfma", fl)
            work[n + "/bwd"] = ("mfma", fl)
            work[n + "/wgrad:mc_gemm"] = ("mfma", fl)
            work[n + "/wgrad:mc_sum_rows"] = ("hbm", 4.0 * outs(l) * mb)
        elif t == "cccp":
            pix = outs(ins[0]) // (l["channels"] * l["in_nodes"])
            fl = 2.0 * l["in_nodes"] * l["out_nodes"] * l["channels"] * pix * mb
            work[n + "/fwd"] = ("mfma", fl)
            work[n + "/bwd"] = ("mfma", fl)
            work[n + "/wgrad:mc_gemm"] = ("mfma", fl)
            work[n + "/wgrad:mc_sum_rows"] = ("hbm", 4.0 * outs(l) * mb)
        elif t == "pool":
            i, o = outs(ins[0]) * mb, outs(l) * mb
            work[n + "/fwd"] = ("hbm", 4.0 * (i + o))
            # with the dropout layer above fused in: + its output, + a keep byte per four elements (written forward, read backward)
            work[n + "/fwd:mc_conv_local_pool_dropout"] = ("hbm", 4.0 * (i + 2 * o) + o / 4.0)
            if l["pool"] == "max":
                work[n + "/bwd"] = ("hbm", 4.0 * (2 * i + 2 * o))         # images + maxActs + maxGrads + target
                work[n + "/bwd:mc_conv_local_max_undo_dropout"] = ("hbm", 4.0 * (2 * i + 2 * o) + o / 4.0)
            else:
                work[n + "/bwd:mc_conv_local_avg_undo_relu_grad"] = ("hbm", 4.0 * (2 * i + o))   # grads + relu acts + target
                work[n + "/bwd:mc_conv_local_avg_undo"] = ("hbm", 4.0 * (i + o))
        elif t in ("rnorm", "cnorm", "cmrnorm"):
            e = outs(l) * mb
            work[n + "/fwd"] = ("hbm", 4.0 * 3 * e)                        # in + denoms + out
            work[n + "/bwd"] = ("hbm", 4.0 * 6 * e)                        # outGrad, denoms, in, acts r + acts w + dIn
        elif t == "neuron":
            e = outs(l) * mb
            work[n + "/fwd"] = ("hbm", 4.0 * 2 * e)
            work[n + "/bwd"] = ("hbm", 4.0 * (2 if l["neuron"]["type"] == "dropout" else 3) * e)
    return work


def attribute(table, work):
    """table: {profile key: (launches, total_ms)} -> rows {key, kind, avg_us, per-launch work, achieved, frac} for every key
    the work table knows (longest matching prefix)."""
    rows = []
    for key, (n, ms) in table.items():
        best = None
        for pre in work:
            if key.startswith(pre) and (best is None or len(pre) > len(best)):
                best = pre
        if best is None or n == 0:
            continue
        kind, amount = work[best]
        avg_s = ms / n * 1e-3
        if kind == "mfma":
            ach, peak, unit = amount / avg_s / 1e12, FP32_MFMA_PEAK / 1e12, "TFLOP/s"
        else:
            ach, peak, unit = amount / avg_s / 1e9, HBM_PEAK / 1e9, "GB/s"
        rows.append({"kernel": key, "bound": kind, "avg_us": avg_s * 1e6, "work": amount, "achieved": ach, "peak": peak, "unit": unit,
                     "frac": ach / peak})
    return rows
