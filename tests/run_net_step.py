"""Child process of the runtime GPU tests: builds a net in libkernel (one model per process, as in
the reference), trains / tests on one synthetic batch with injected dropout masks and dumps costs,
final weights and a few activations to an .npz the parent compares with the CPU oracle."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))
import modelutil  # noqa: E402


def make_masks(layers, cases, mini, seed=99):
    """[minibatch][dropout layer] uniform masks, in the order fprop consumes them."""
    rng = np.random.default_rng(seed)
    out = []
    for s in range(0, cases, mini):
        n = min(mini, cases - s)
        out.append([rng.random((outs, n), dtype=np.float32) for _, outs in modelutil.dropout_layers(layers)])
    return out


def multiview_batch(dims, classes, cases, views):
    """`views` different synthetic views of `cases` images, view-major as the cropped provider lays them
    out (convdata.py:68-80), every view carrying the same labels."""
    data = modelutil.synthetic_batch(dims, classes, cases * views)
    data[1] = np.ascontiguousarray(np.tile(data[1][:, :cases], (1, views)))
    return data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", required=True)
    ap.add_argument("--cases", type=int, default=256)
    ap.add_argument("--mini", type=int, default=128)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--test", type=int, default=0)
    ap.add_argument("--single-rank-comm", type=int, default=0)
    ap.add_argument("--check-grads", type=int, default=0)
    ap.add_argument("--acts", default="")
    ap.add_argument("--multiview", type=int, default=0, help="number of views: startMultiviewTest on cases*views columns")
    ap.add_argument("--features", default="", help="layer name: startFeatureWriter into a (cases, outputs) matrix")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import libkernel
    layers, dims, classes = modelutil.load_net(a.net, seed=a.seed)
    data = multiview_batch(dims, classes, a.cases, a.multiview) if a.multiview else modelutil.synthetic_batch(dims, classes, a.cases)
    libkernel.initModel(layers, a.mini, 0)
    if a.single_rank_comm:
        libkernel.initComm(0, 1, libkernel.commUniqueId(), 1)
    if a.check_grads:
        failures = libkernel.checkGradients(data)
        np.savez(a.out, failures=failures)
        return
    if a.multiview:
        logreg = [i for i, l in enumerate(layers) if l["type"] == "cost.logreg"][0]
        libkernel.startMultiviewTest(data, a.multiview, logreg)
        costs, ncases = libkernel.finishBatch()
        np.savez(a.out, numCases=ncases, **{"cost:" + k: np.array(v) for k, v in costs.items()})
        return
    if a.features:
        idx = [i for i, l in enumerate(layers) if l["name"] == a.features][0]
        ftrs = np.full((a.cases, layers[idx]["outputs"]), np.nan, dtype=np.float32)
        libkernel.startFeatureWriter(data + [ftrs], idx)
        costs, ncases = libkernel.finishBatch()
        np.savez(a.out, numCases=ncases, ftrs=ftrs, **{"cost:" + k: np.array(v) for k, v in costs.items()})
        return
    if not a.test:
        libkernel.setDropoutMasks([m for mb in make_masks(layers, a.cases, a.mini) for m in mb])
    libkernel.startBatch(data, a.test)
    costs, ncases = libkernel.finishBatch()
    res = {"numCases": ncases}
    for k, v in costs.items():
        res["cost:" + k] = np.array(v)
    for name in [x for x in a.acts.split(",") if x]:
        res["acts:" + name] = libkernel.getActs(name)
    libkernel.syncWithHost()
    for l in layers:
        if "weights" in l:
            for i, w in enumerate(l["weights"]):
                res["W:%s:%d" % (l["name"], i)] = np.array(w)
                res["Winc:%s:%d" % (l["name"], i)] = np.array(l["weightsInc"][i])
            res["B:" + l["name"]] = np.array(l["biases"])
            res["Binc:" + l["name"]] = np.array(l["biasesInc"])
    np.savez(a.out, **res)


if __name__ == "__main__":
    main()
