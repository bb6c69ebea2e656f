"""Bit equality of forked (two-stream) and single-stream eager training steps on the full-size BASELINE nets, ten minibatches
each: a race between the weight-gradient branch and the main stream would show as a differing weight, bias or cost.
   python tools/fork_stress.py        (GPU; ~2 minutes)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_runtime_gpu as t
for net, cases, mini in (("nin", 10 * 128, 128), ("imagenet", 6 * 128, 128), ("18pct", 10 * 128 + 50, 128), ("conv-local", 10 * 128, 128)):
    t0 = time.time()
    kw = dict(net=net, cases=cases, mini=mini, init_w=0.0 if net == "imagenet" else 0.05)   # (imagenet blows up to NaN at 0.05)
    if net == "nin": kw["philox_seed"] = 11
    a, _ = t.run_child(_env={"MC_NO_STEP_GRAPH": "1"}, **kw)
    b, _ = t.run_child(_env={"MC_NO_STEP_GRAPH": "1", "MC_NO_STEP_FORK": "1"}, **kw)
    bad = [k for k in b if (k.startswith("cost:") or k[0] in "WB") and not (np.array_equal(a[k], b[k]) and np.isfinite(b[k]).all())]
    print(net, "forked steps", int(a["forkedSteps"]), "vs", int(b["forkedSteps"]), "mismatching tensors:", bad, "%.0fs" % (time.time() - t0), flush=True)
    assert not bad
    # the same through the one-rank RCCL path: gradient buckets are reduced behind the branch stream (or both streams)
    c, _ = t.run_child(single_rank_comm=1, **kw)
    bad = [k for k in b if (k.startswith("cost:") or k[0] in "WB") and not np.array_equal(c[k], b[k])]
    print(net, "with a communicator: forked steps", int(c["forkedSteps"]), "mismatching tensors:", bad, flush=True)
    assert not bad
print("ok")
