"""Per-layer activation / gradient comparison of libkernel vs the CPU oracle after one minibatch
(debugging aid; run on the GPU box): python tools/debug_parity.py <net> <cases>"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import modelutil
from oracle.golden_net import GoldenNet
from run_net_step import make_masks

net_name, cases = sys.argv[1], int(sys.argv[2])
mini = int(sys.argv[3]) if len(sys.argv) > 3 else cases
IW = float(os.environ.get("INIT_W", "0"))
layers, dims, classes = modelutil.load_net(net_name, init_w=IW)
names = [l["name"] for l in layers if not l["type"].startswith("cost.")]
out = os.path.join(tempfile.mkdtemp(), "r.npz")
code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import modelutil, libkernel
from run_net_step import make_masks
layers, dims, classes = modelutil.load_net(%r, init_w=%r)
data = modelutil.synthetic_batch(dims, classes, %d)
libkernel.initModel(layers, %d, 0)
libkernel.setDropoutMasks([m for mb in make_masks(layers, %d, %d) for m in mb])
libkernel.startBatch(data, 0); print(libkernel.finishBatch())
res = {}
for n in %r:
    res["a:" + n] = libkernel.getActs(n)
    try: res["g:" + n] = libkernel.getActs(n, 1)
    except Exception as e: pass
libkernel.syncWithHost()
for l in layers:
    if "weights" in l:
        res["W:" + l["name"]] = np.array(l["weightsInc"][0]); res["B:" + l["name"]] = np.array(l["biasesInc"])
np.savez(%r, **res)
''' % (os.path.join(ROOT, "cuda-convnet_amd"), os.path.join(ROOT, "tests"), net_name, IW, cases, mini, cases, mini, names, out)
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
print(r.stdout[-500:], r.stderr[-2000:])
got = dict(np.load(out))
data = modelutil.synthetic_batch(dims, classes, cases)
g = GoldenNet(layers)
m = make_masks(layers, cases, mini)
dn = [n for n, _ in modelutil.dropout_layers(layers)]
starts = list(range(0, cases, mini))
for b, s0 in enumerate(starts):
    mb = [d[:, s0:s0 + mini] for d in data]
    print(g.fprop(mb, masks=dict(zip(dn, m[b]))))
    g.bprop()
    if b + 1 < len(starts):
        g.update()
def rel(a, b):
    b = b.astype(np.float64); s = max(np.sqrt((b**2).mean()), 1e-30)
    return np.abs(a - b).max() / s
for n in names:
    ea = rel(got["a:" + n], g.acts[n]) if got["a:" + n].shape == g.acts[n].shape else -1
    eg = rel(got["g:" + n], g.grads[n]) if ("g:" + n in got and n in g.grads and g.grads[n] is not None and got["g:" + n].shape == g.grads[n].shape) else -1
    print("%-16s acts %.2e   grad %.2e" % (n, ea, eg))

g.update()
for l in layers:
    if "weights" in l:
        print("%-10s Winc %.2e  Binc %.2e" % (l["name"], rel(got["W:" + l["name"]], g.Winc[(l["name"], 0)]), rel(got["B:" + l["name"]], g.Binc[l["name"]])))
