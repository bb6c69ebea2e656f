"""Training-step throughput of any BASELINE config through the product path (libkernel -> libmckernels):
   python tools/net_bench.py <net> [minibatch] [steps]     net in modelutil.NETS (80sec, 18pct, conv-local, imagenet, nin)
Prints images/sec, ms/step and the ten heaviest launches (HIP-event timed)."""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cuda-convnet_amd"))
import torch
import modelutil, libkernel
net = sys.argv[1]; mb = int(sys.argv[2]) if len(sys.argv) > 2 else 128; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
layers, dims, classes = modelutil.load_net(net, seed=42)
libkernel.initModel(layers, mb, 0)
libkernel.setSeed(99)
nres = 4
data = modelutil.synthetic_batch(dims, classes, nres * mb)
libkernel.startBatch(data, 0); libkernel.finishBatch()
libkernel.startResident(0, 0, 10); libkernel.finishBatch()
torch.cuda.synchronize(); t0 = time.perf_counter()
libkernel.startResident(0, 0, steps); costs, n = libkernel.finishBatch()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
counters = dict((k, v[0]) for k, v in libkernel.getProfile().items() if k.startswith("net/step_"))
libkernel.setProfile("*"); libkernel.startResident(0, 0, 5); libkernel.finishBatch(); tab = libkernel.getProfile(); libkernel.setProfile("")
rows = sorted(((k, c / 5.0, ms / c) for k, (c, ms) in tab.items() if c and ms), key=lambda r: -r[1] * r[2])
print("%s mb=%d: %.0f images/s, %.3f ms/step, cost/case %.4f; kernel time/step %.3f ms" % (net, mb, mb * steps / dt, dt / steps * 1e3, list(costs.values())[0][0] / n, sum(r[1] * r[2] for r in rows)))
print("   steps replayed as graphs %d, eager steps with the weight-gradient branch %d" % (counters.get("net/step_graph:replays", 0), counters.get("net/step_fork:steps", 0)))
for k, c, ms in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 10]: print("   %-52s x%.0f %8.4f ms" % (k, c, ms))
if len(sys.argv) > 5:   # the whole event table (every launch timed, one stream: the kernels' own times) as JSON
    json.dump({"net": net, "minibatch": mb, "images_per_s": mb * steps / dt, "ms_per_step": dt / steps * 1e3,
               "launches": [{"name": k, "per_step": c, "avg_ms": ms} for k, c, ms in rows]}, open(sys.argv[5], "w"), indent=1)
