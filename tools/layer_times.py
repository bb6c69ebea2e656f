"""Per-layer HIP-event timings of one net through libkernel, filtered by a substring:
   python tools/layer_times.py <net> <substring> [minibatch]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-convnet_amd"))
import torch, modelutil, libkernel
net, sub = sys.argv[1], sys.argv[2]
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 128
layers, dims, classes = modelutil.load_net(net, seed=42)
libkernel.initModel(layers, mb, 0); libkernel.setSeed(99)
data = modelutil.synthetic_batch(dims, classes, 2 * mb)
libkernel.startBatch(data, 0); libkernel.finishBatch()
libkernel.startResident(0, 0, 6); libkernel.finishBatch()
libkernel.setProfile("*"); libkernel.startResident(0, 0, 10); libkernel.finishBatch(); tab = libkernel.getProfile()
for k, (c, ms) in sorted(tab.items()):
    if sub in k: print("%-55s %8.1f us" % (k, ms / c * 1e3))
