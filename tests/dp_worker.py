"""One rank of the 2-process CPU rehearsal of the data-parallel path (gloo): the same control flow
bench.py / convnet.py use -- dp.connect() rendezvous, column sharding, per-rank gradients scaled by
eps / (N_local * world), sum all-reduce, identical update on every rank -- with the CPU oracle net
standing in for the GPU runtime and gloo for RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dp  # noqa: E402
import modelutil  # noqa: E402
from oracle.golden_net import GoldenNet  # noqa: E402


class FakeLib(object):
    """Records what dp.connect would hand to libkernel."""
    def commUniqueId(self):
        return bytes(range(128))
    def initComm(self, rank, world, uid, force=0):
        self.joined = (rank, world, uid)


def main():
    out = sys.argv[1]
    rank, world, _ = dp.env()
    lib = FakeLib()
    dp.connect(lib, rank, world)
    assert lib.joined == (rank, world, bytes(range(128)))
    dist = dp.init_control_group()

    mini, cases = 8, 43          # global minibatch 16: two full ones + a ragged tail of 11 (6 + 5)
    layers, dims, classes = modelutil.load_net("gc-small", seed=11)
    data = modelutil.synthetic_batch(dims, classes, cases)
    cols = dp.shard_columns(cases, mini, rank, world)
    mine = [d[:, cols] for d in data]
    net = GoldenNet(layers, world=world)
    total = dp.global_cases(cases, mini, world)
    shard_costs = {}
    for i, s in enumerate(range(0, mine[0].shape[1], mini)):
        net.global_cases = min(mini * world, total - i * mini * world)     # what libkernel derives from startBatch's third argument
        for name, vals in net.fprop([d[:, s:s + mini] for d in mine]).items():
            shard_costs[name] = [a + float(b) for a, b in zip(shard_costs.get(name, [0.0] * len(vals)), vals)]
        net.bprop()
        for store in (net.wgrads, net.bgrads):
            for k in sorted(store, key=str):
                t = torch.from_numpy(store[k])
                dist.all_reduce(t)           # sum, in place (what ncclAllReduce does on the GPU path)
        net.update()
    # what convnet.ConvNet.finish_batch reports: this rank's (costs, cases) summed over the ranks
    costs, seen = dp.sum_costs((shard_costs, mine[0].shape[1]), world)
    res = {"cases": np.array(seen)}
    for name, vals in costs.items():
        res["cost:" + name] = np.array(vals, dtype=np.float64)
    for l in layers:
        if "weights" in l:
            res["W:" + l["name"]] = net.weight(l["name"], 0)
            res["B:" + l["name"]] = net.B[l["name"]]
    np.savez(out + ".%d.npz" % rank, **res)
    dist.barrier()


if __name__ == "__main__":
    main()
