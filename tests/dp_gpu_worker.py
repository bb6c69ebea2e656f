"""One rank of the 2-GPU data-parallel test: libkernel on its own device, RCCL all-reduce of the gradient buckets inside the
native module, rendezvous through dp.connect.  Trains on its share of a ragged batch and saves the weights."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))
import torch  # noqa: E402,F401  (before libkernel: one HIP runtime)
import dp  # noqa: E402
import modelutil  # noqa: E402


def main():
    out, net_name, cases, mini = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rank, world, local = dp.env()
    import libkernel
    layers, dims, classes = modelutil.load_net(net_name, seed=11)
    libkernel.initModel(layers, mini, local)
    libkernel.setSeed(7 + rank)
    dp.connect(libkernel, rank, world)
    data = modelutil.synthetic_batch(dims, classes, cases)
    cols = dp.shard_columns(cases, mini, rank, world)
    mine = [np.require(d[:, cols], dtype=np.float32, requirements="C") for d in data]
    libkernel.startBatch(mine, 0, dp.global_cases(cases, mini, world))
    cost = libkernel.finishBatch()
    prof = libkernel.getProfile()
    libkernel.syncWithHost()
    res = {"cases": np.array([cost[1]]), "allreduce": np.array(prof.get("dp/allreduce:bytes", (0, 0.0)))}
    whole, seen = dp.sum_costs(cost, world)          # what convnet.ConvNet.finish_batch hands the driver on every rank
    res["globalCases"] = np.array([seen])
    for name, vals in whole.items():
        res["cost:" + name] = np.array(vals, dtype=np.float64)
    for l in layers:
        if "weights" in l:
            for i, w in enumerate(l["weights"]):
                res["W%d:%s" % (i, l["name"])] = np.array(w)
            res["B:" + l["name"]] = np.array(l["biases"])
    np.savez(out + ".%d.npz" % rank, **res)
    dp.init_control_group().barrier()


if __name__ == "__main__":
    main()
