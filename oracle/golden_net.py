"""oracle/golden_net.py -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of one cuda-convnet training step over a parsed layer list: fprop -> cost ->
bprop -> SGD update, built from the per-op oracle in golden.py / golden.c.  It follows the layer
semantics of the reference's Kernel/src/layers/layer.cu (cited per branch below) and the cccp /
dropout plugin (PluginsSrc/src/extralayers.cu:58-174, PluginsSrc/include/extralayers.cuh:34-102).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: the conv / pool / norm ops underneath are pinned to the reference's DataGen.m
cases through SciPy (tests/test_oracle_pin.py); the graph plumbing here is pinned by the
reference's own integration check, finite differences on layers.gc.cfg-style nets
(tests/test_golden_net.py), since the reference ships no stored outputs for a whole step.
"""
import numpy as np

from . import golden as G

F32 = np.float32


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


class GoldenNet(object):
    def __init__(self, layers, world=1):
        """`layers`: the list LayerParser.parse_layers returns.  Weight arrays are copied."""
        self.layers = layers
        self.world = world
        self.W, self.Winc, self.B, self.Binc = {}, {}, {}, {}
        for li, l in enumerate(layers):
            if "weights" not in l:
                continue
            for i, w in enumerate(l["weights"]):
                src = l["weightSourceLayerIndices"][i]
                if src >= 0:      # shared: alias the owner's entry
                    key = (layers[src]["name"] if src < len(layers) and src != li else l["name"], l["weightSourceMatrixIndices"][i])
                    self.W[(l["name"], i)] = key
                else:
                    self.W[(l["name"], i)] = np.array(w, dtype=F32, order="C")
                    self.Winc[(l["name"], i)] = np.array(l["weightsInc"][i], dtype=F32, order="C")
            self.B[l["name"]] = np.array(l["biases"], dtype=F32, order="C")
            self.Binc[l["name"]] = np.array(l["biasesInc"], dtype=F32, order="C")
        self.acts, self.grads, self.aux = {}, {}, {}

    # -- weight access ----------------------------------------------------------------------
    def _owner(self, key):
        while not isinstance(self.W[key], np.ndarray):
            key = self.W[key]
        return key

    def weight(self, name, i=0):
        return self.W[self._owner((name, i))]

    # -- forward ------------------------------------------------------------------------------
    def fprop(self, data, train=True, gc=False, masks=None):
        """data: list of (dims, N) float32 arrays.  masks: dict layer name -> uniform mask (dropout)."""
        L, A = self.layers, self.acts
        N = data[0].shape[1]
        self.N = N
        self.costs = {}
        masks = masks or {}
        for l in L:
            t, name = l["type"], l["name"]
            ins = [A[L[i]["name"]] for i in l.get("inputs", [])]
            if t == "data":
                A[name] = _f32(data[l["dataIdx"]])
            elif t in ("conv", "local"):                                   # layer.cu:470-489, 557-570
                out = None
                for i, x in enumerate(ins):
                    x, ch = self._conv_input(l, i, x, N)
                    geo = (N, ch, l["imgSize"][i], l["imgSize"][i], l["filters"], l["filterSize"][i], l["padding"][i],
                           l["stride"][i], l["modulesX"], l["modulesX"], l["groups"][i])
                    out = G.filter_acts(x, self.weight(name, i), *geo, targets=out, scale_targets=0.0 if out is None else 1.0, conv=(t == "conv"))
                b = self.B[name]
                if t == "conv" and l["sharedBiases"]:
                    out = (out.reshape(l["filters"], -1) + b.reshape(-1, 1)).reshape(out.shape).astype(F32)
                else:
                    out = (out + b.reshape(-1, 1)).astype(F32)
                A[name] = out
            elif t == "fc":                                                # layer.cu:373-378
                out = sum(G.matmul(self.weight(name, i).T, x).astype(np.float64) for i, x in enumerate(ins))
                A[name] = (out + self.B[name].reshape(-1, 1)).astype(F32)
            elif t == "cccp":                                              # extralayers.cu:66-92
                ch, ni, no = l["channels"], l["in_nodes"], l["out_nodes"]
                x = ins[0]
                P = x.shape[0] // (ch * ni)
                X = x.reshape(ch, ni, P * N)
                W = self.weight(name, 0)                                   # (in, out*ch)
                Y = np.stack([G.matmul(W[:, c * no:(c + 1) * no].T, X[c]) for c in range(ch)])
                Y = Y + self.B[name].reshape(ch, no, 1)
                A[name] = _f32(Y.reshape(ch * no * P, N))
            elif t == "neuron":                                            # layer.cu:251-262
                nt, params = l["neuron"]["type"], l["neuron"]["params"]
                if nt == "dropout":
                    m = masks.get(name)
                    self.aux[name] = m
                    A[name] = G.neuron_fwd("dropout", params, ins[0], train=(train and not gc), mask=m)
                else:
                    A[name] = G.neuron_fwd(nt, params, ins[0])
            elif t == "pool":                                              # layer.cu:741-763
                A[name] = G.pool(ins[0], N, l["channels"], l["imgSize"], l["sizeX"], l["start"], l["stride"], l["outputsX"], l["pool"])
            elif t == "rnorm":                                             # layer.cu:882-884
                A[name], self.aux[name] = G.rnorm(ins[0], N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"])
            elif t == "cnorm":                                             # layer.cu:925-930
                mean = G.pool(ins[0], N, l["channels"], l["imgSize"], l["size"], -(l["size"] // 2), 1, l["imgSize"], "avg")
                md = (ins[0] - mean).astype(F32)
                A[name], den = G.rnorm(ins[0], N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"], mean_diffs=md)
                self.aux[name] = (den, md)
            elif t == "cmrnorm":                                           # layer.cu:907-909
                A[name], self.aux[name] = G.cmrnorm(ins[0], N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"], bool(l["blocked"]), l["k"])
            elif t == "softmax":                                           # layer.cu:612-622
                A[name] = G.softmax(ins[0])
            elif t == "eltsum":                                            # layer.cu:645-651
                A[name] = sum(F32(c) * x for c, x in zip(l["coeffs"], ins)).astype(F32)
            elif t == "eltmax":                                            # layer.cu:670-676
                A[name] = G.eltwise_max(ins)
            elif t == "nailbed":                                           # layer.cu:778-780
                A[name] = G.bed_of_nails(ins[0], N, l["channels"], l["imgSize"], l["start"], l["stride"])
            elif t == "blur":                                              # layer.cu:796-799: rows, then columns
                h = G.gaussian_blur(ins[0], l["filter"], N, l["channels"], l["imgSize"], True)
                A[name] = G.gaussian_blur(h, l["filter"], N, l["channels"], l["imgSize"], False)
            elif t == "resize":                                            # layer.cu:825-827
                A[name] = G.resize_bilinear(ins[0], N, l["channels"], l["imgSize"], l["tgtSize"], l["scale"])
            elif t == "rgb2yuv":                                           # layer.cu:842-844
                A[name] = G.rgb_to_yuv(ins[0], N, l["imgPixels"])
            elif t == "rgb2lab":                                           # layer.cu:860-862
                A[name] = G.rgb_to_lab(ins[0], N, l["imgPixels"], bool(l["center"]))
            elif t == "cost.logreg":                                       # layer.cu:993-1005
                self.costs[name] = G.logreg_cost(ins[0], ins[1])
            elif t == "cost.gsinlabel":                                    # PluginsSrc/src/extralayers.cu:10-32
                self.costs[name] = [G.group_sparsity_in_label_cost(ins[0], ins[1], l["channels"])[0]]
            elif t == "cost.sum2":                                         # layer.cu:1028-1032
                self.costs[name] = [0.5 * float((ins[0].astype(np.float64) ** 2).sum())]
            else:
                raise NotImplementedError("golden net: layer type " + t)
        return self.costs

    @staticmethod
    def _conv_input(l, i, x, N):
        """(input tensor, channel count) the dense grouped contraction of input i runs on: for a randSparse input
        the connected channels gathered group by group (layer.cu:471-474, 509-512, 526-530)."""
        if l["randSparse"][i]:
            return G.gather_channels(x, l["filterConns"][i], l["channels"][i], N), l["groups"][i] * l["filterChannels"][i]
        return x, l["channels"][i]

    def cost_value(self):
        """Cost::getValue (Kernel/src/layers/cost.cu:64-70)."""
        return sum(l["coeff"] * self.costs[l["name"]][0] for l in self.layers if l["type"].startswith("cost."))

    # -- backward -----------------------------------------------------------------------------
    def _grad_consumer(self, idx):
        l = self.layers[idx]
        return bool(l["gradConsumer"]) or any(self._grad_consumer(i) for i in l.get("inputs", []))

    def bprop(self, train=True, gc=False, decide=None):
        """Fills self.grads (activity gradients) and self.wgrads / self.bgrads (already scaled by
        eps / numCases as the reference does, layer.cu:491-523; scale 1 under PASS_GC).

        decide (tests only): dict layer name -> that layer's forward output as ANOTHER implementation computed it.  The
        two discontinuous decisions of back-propagation -- a relu's gradient mask (output > 0, neuron.cuh:166-178) and a
        max-pool's equality routing (conv_util.cu:1567) -- are then taken from those values instead of this net's own, so
        that a pre-activation within rounding of zero, or two window entries within rounding of each other, cannot make
        two correct implementations disagree by O(1) in single gradient entries.  The caller checks that the supplied
        values agree with this net's forward pass and counts the decisions that differ (self.flips)."""
        L, A, N = self.layers, self.acts, self.N
        decide = decide or {}
        self.flips = {}
        Ng = self.global_cases if getattr(self, 'global_cases', 0) else N * self.world   # data parallel: cases of the GLOBAL minibatch
        g = {}
        self.wgrads, self.bgrads = {}, {}

        def send(idx, val):
            if not self._grad_consumer(idx):
                return
            nm = L[idx]["name"]
            g[nm] = _f32(val) if nm not in g else (g[nm] + val).astype(F32)

        for li in range(len(L) - 1, -1, -1):
            l = L[li]
            t, name = l["type"], l["name"]
            inp = l.get("inputs", [])
            if t == "data":
                continue
            if t.startswith("cost."):
                if l["coeff"] == 0:
                    continue
                if t == "cost.logreg":                                     # layer.cu:1007-1018
                    sm = L[inp[1]]
                    consumers = sum(inp[1] in ll.get("inputs", []) for ll in L)
                    if consumers > 1 or sm["type"] != "softmax":
                        send(inp[1], G.logreg_grad(A[L[inp[0]]["name"]], A[sm["name"]], l["coeff"]))
                    else:
                        g.setdefault(sm["name"], None)    # the softmax computes the fused gradient itself
                elif t == "cost.gsinlabel":                                # PluginsSrc/src/extralayers.cu:34-38
                    send(inp[1], G.group_sparsity_in_label_grad(A[L[inp[0]]["name"]], A[L[inp[1]]["name"]], l["channels"], l["coeff"]))
                else:                                                      # layer.cu:1034-1036
                    send(inp[0], -F32(l["coeff"]) * A[L[inp[0]]["name"]])
                continue
            if name not in g:
                continue
            v = g[name]
            ins = [A[L[i]["name"]] for i in inp]
            if t in ("conv", "local"):
                conv = t == "conv"
                if l["epsB"] > 0:                                          # layer.cu:491-501, 572-576
                    s = 1.0 if gc else l["epsB"] / Ng
                    rows = l["filters"] if (conv and l["sharedBiases"]) else l["filters"] * l["modules"]
                    self.bgrads[name] = (s * v.reshape(rows, -1).astype(np.float64).sum(axis=1)).astype(F32).reshape(self.B[name].shape)
                for i, x in enumerate(ins):
                    x, ch = self._conv_input(l, i, x, N)
                    geo = (N, ch, l["imgSize"][i], l["imgSize"][i], l["filters"], l["filterSize"][i], l["padding"][i],
                           l["stride"][i], l["modulesX"], l["modulesX"], l["groups"][i])
                    if l["epsW"][i] > 0:                                   # layer.cu:503-523, 578-591
                        s = 1.0 if gc else l["epsW"][i] / Ng
                        dw = G.weight_acts(x, v, *geo, partial_sum=0, scale_output=s, conv=conv)
                        key = self._owner((name, i))
                        self.wgrads[key] = dw if key not in self.wgrads else (self.wgrads[key] + dw).astype(F32)
                    dx = G.img_acts(v, self.weight(name, i), *geo, conv=conv)               # layer.cu:525-540, 593-602
                    if l["randSparse"][i]:
                        dx = G.scatter_channels(dx, l["filterConns"][i], l["channels"][i], N)
                    send(inp[i], dx)
            elif t == "fc":                                                # layer.cu:380-409
                if l["epsB"] > 0:
                    s = 1.0 if gc else l["epsB"] / Ng
                    self.bgrads[name] = (s * v.astype(np.float64).sum(axis=1)).astype(F32).reshape(self.B[name].shape)
                for i, x in enumerate(ins):
                    if l["epsW"][i] > 0:
                        s = 1.0 if gc else l["epsW"][i] / Ng
                        dw = (s * (x.astype(np.float64) @ v.astype(np.float64).T)).astype(F32)     # (in, out)
                        key = self._owner((name, i))
                        self.wgrads[key] = dw if key not in self.wgrads else (self.wgrads[key] + dw).astype(F32)
                    send(inp[i], G.matmul(self.weight(name, i), v))
            elif t == "cccp":                                              # extralayers.cu:103-174
                ch, ni, no = l["channels"], l["in_nodes"], l["out_nodes"]
                P = ins[0].shape[0] // (ch * ni)
                V = v.reshape(ch, no, P * N)
                X = ins[0].reshape(ch, ni, P * N)
                W = self.weight(name, 0)
                if l["epsB"] > 0:
                    s = 1.0 if gc else l["epsB"] / Ng
                    self.bgrads[name] = (s * V.astype(np.float64).sum(axis=2)).astype(F32).reshape(self.B[name].shape)
                if l["epsW"][0] > 0:
                    s = 1.0 if gc else l["epsW"][0] / Ng
                    dw = np.concatenate([(s * (X[c].astype(np.float64) @ V[c].astype(np.float64).T)) for c in range(ch)], axis=1).astype(F32)
                    self.wgrads[self._owner((name, 0))] = dw
                dX = np.stack([G.matmul(W[:, c * no:(c + 1) * no], V[c]) for c in range(ch)])
                send(inp[0], dX.reshape(ch * ni * P, N))
            elif t == "neuron":
                nt, params = l["neuron"]["type"], l["neuron"]["params"]
                if nt == "dropout":
                    send(inp[0], G.neuron_bwd("dropout", params, v, A[name], ins[0], train=(train and not gc), mask=self.aux.get(name)))
                elif nt == "relu" and name in decide:
                    theirs = decide[name]
                    self.flips[name] = np.flatnonzero((theirs > 0) != (A[name] > 0))
                    send(inp[0], G.neuron_bwd(nt, params, v, theirs, ins[0]))
                else:
                    send(inp[0], G.neuron_bwd(nt, params, v, A[name], ins[0]))
            elif t == "pool":
                geo = (N, l["channels"], l["imgSize"], l["sizeX"], l["start"], l["stride"], l["outputsX"])
                if l["pool"] == "max" and name in decide and L[inp[0]]["name"] in decide:
                    x2, y2 = decide[L[inp[0]]["name"]], decide[name]
                    mine, theirs = G.max_undo(ins[0], np.ones_like(v), A[name], *geo), G.max_undo(x2, np.ones_like(v), y2, *geo)
                    self.flips[name] = np.flatnonzero(mine != theirs)          # input pixels routed to differently
                    send(inp[0], G.max_undo(x2, v, y2, *geo))
                elif l["pool"] == "max":
                    send(inp[0], G.max_undo(ins[0], v, A[name], *geo))
                else:
                    send(inp[0], G.avg_undo(v, *geo))
            elif t == "rnorm":
                send(inp[0], G.rnorm_undo(v, self.aux[name], ins[0], A[name].copy(), N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"]))
            elif t == "cnorm":
                den, md = self.aux[name]
                send(inp[0], G.rnorm_undo(v, den, md, A[name].copy(), N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"]))
            elif t == "cmrnorm":
                send(inp[0], G.cmrnorm_undo(v, self.aux[name], ins[0], A[name].copy(), N, l["channels"], l["imgSize"], l["size"], l["scale"], l["pow"], bool(l["blocked"])))
            elif t == "softmax":                                           # layer.cu:624-634
                nxt = [ll for ll in L if li in ll.get("inputs", [])]
                if len(nxt) == 1 and nxt[0]["type"] == "cost.logreg":
                    labels = A[L[nxt[0]["inputs"][0]]["name"]]
                    send(inp[0], G.logreg_softmax_grad(labels, A[name], nxt[0]["coeff"]))
                else:
                    send(inp[0], G.softmax_grad(A[name], v))
            elif t == "eltsum":
                for i in range(len(inp)):
                    send(inp[i], F32(l["coeffs"][i]) * v)
            elif t == "eltmax":                                            # layer.cu:678-680
                for i in range(len(inp)):
                    send(inp[i], G.eltwise_max_grad(v, ins[i], A[name]))
            elif t == "nailbed":                                           # layer.cu:782-784
                send(inp[0], G.bed_of_nails_undo(v, N, l["channels"], l["imgSize"], l["start"], l["stride"]))
            elif t == "blur":                                              # layer.cu:803-807
                h = G.gaussian_blur(v, l["filter"], N, l["channels"], l["imgSize"], True)
                send(inp[0], G.gaussian_blur(h, l["filter"], N, l["channels"], l["imgSize"], False))
            elif t in ("resize", "rgb2yuv", "rgb2lab"):                    # layer.cu:829-832, 846-849, 864-867
                assert not self._grad_consumer(inp[0]), "no backward pass through " + t
            else:
                raise NotImplementedError("golden net: layer type " + t)
        self.grads = g

    # -- update -------------------------------------------------------------------------------
    def update(self):
        """Weights::update for every owned tensor (Kernel/include/layers/weights.cuh:150-164)."""
        for l in self.layers:
            if "weights" not in l:
                continue
            name = l["name"]
            for i in range(len(l["weights"])):
                key = (name, i)
                if isinstance(self.W[key], np.ndarray) and l["epsW"][i] > 0 and key in self.wgrads:
                    G.sgd_update(self.W[key], self.Winc[key], self.wgrads[key].reshape(self.W[key].shape), l["epsW"][i], l["wc"][i], l["momW"][i])
            if l["epsB"] > 0 and name in self.bgrads:
                G.sgd_update(self.B[name], self.Binc[name], self.bgrads[name], l["epsB"], 0.0, l["momB"])

    def train_step(self, data, masks=None):
        costs = self.fprop(data, train=True, masks=masks)
        self.bprop(train=True)
        self.update()
        return costs
