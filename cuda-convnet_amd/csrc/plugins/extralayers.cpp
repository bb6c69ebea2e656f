// extralayers.cpp -> Plugins/extralayers.so : the NIN plugin.
//   layer  "cccp"    : cascaded cross-channel parametric pooling (a 1x1 convolution over feature
//                      maps), reference PluginsSrc/src/extralayers.cu:58-174
//   neuron "dropout" : reference PluginsSrc/include/extralayers.cuh:34-102
//   cost   "cost.gsinlabel" : group sparsity over the cases that share a label,
//                      reference PluginsSrc/src/extralayers.cu:5-46 + extrautils.cu
// Exports the two plugin entry points of the reference (extralayers.cu:200-215).
#include <stdlib.h>
#include <iostream>

#include "../runtime/convnet.h"

using namespace rt;

static inline mcStream_t ST() { return reinterpret_cast<mcStream_t>(computeStream()); }

// Input viewed as (channels*in_nodes, pixels*numCases): feature f = (ch*in_nodes + i)*pixels + p,
// which is how the reference's reshape of the (numCases, features) column-major matrix groups
// it (extralayers.cu:70-79).  Per channel slice: Y(out, P*N) = W(out, in) * X(in, P*N) + b.
class CCCPLayer : public WeightLayer {
public:
    CCCPLayer(ConvNet* convNet, PyObject* paramsDict) : WeightLayer(convNet, paramsDict, true) {
        _wStep = 0.1f;
        _bStep = 0.01f;
        _channels = pyDictGetInt(paramsDict, "channels");
        _in = pyDictGetInt(paramsDict, "in_nodes");
        _out = pyDictGetInt(paramsDict, "out_nodes");
        if (_channels <= 0 || _in <= 0 || _out <= 0) throw std::runtime_error("cccp: channels, in_nodes, out_nodes must be positive");
    }

protected:
    int pixels(int inputFeatures) const {
        if (inputFeatures % (_channels * _in) != 0) throw std::runtime_error(fmt("cccp layer '%s': %d input features are not a multiple of channels*in_nodes = %d", _name.c_str(), inputFeatures, _channels * _in));
        return inputFeatures / (_channels * _in);
    }

    bool readsReluBits(int rows, int, int) const override { return _channels == 1 && rows == _in; }   // its view: (in_nodes, pixels * cases)

    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override {
        const int N = numCases(), P = pixels(_inputs[inpIdx]->rows()), cols = P * N;
        NeuronLayer* relu = fusableRelu();
        if (relu) relu->setFusedIntoProducer(true);
        const size_t wsb = mc_gemm_workspace_bytes(_out, cols, _in);
        // the relu's gradient mask as bits for the dense product above (FCLayer::fpropActs)
        const bool bits = relu && passType == PASS_TRAIN && scaleTargets == 0 && wsb == 0 && _channels == 1 && relu->wantsBits(_out, cols, N);
        if (relu) relu->setBitsValid(bits);
        if (bits) {
            RT_MC(mc_gemm_relu_bits(_weights[inpIdx].getW(), _inputs[inpIdx]->data(), getActs().data(), _out, cols, _in, 0, 0, _in, cols, cols,
                                    _biases->getW(), relu->bitsBuffer(_out, cols), ST()));
            return;
        }
        for (int ch = 0; ch < _channels; ch++) {
            const float* W = _weights[inpIdx].getW() + (size_t)ch * _out * _in;
            const float* X = _inputs[inpIdx]->data() + (size_t)ch * _in * cols;
            float* Y = getActs().data() + (size_t)ch * _out * cols;
            RT_MC(mc_gemm(W, X, Y, _out, cols, _in, 0, 0, _in, cols, cols, scaleTargets, 1.f, scaleTargets == 0 ? _biases->getW() + ch * _out : nullptr,
                          relu ? MC_NEURON_RELU : MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
        }
    }

    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE) override {
        const int N = numCases(), P = pixels(_prev[inpIdx]->getNumOutputs()), cols = P * N;
        const size_t wsb = mc_gemm_workspace_bytes(_in, cols, _out);
        const unsigned* bits = (scaleTargets == 0.f && _channels == 1) ? claimReluGradBits(inpIdx) : nullptr;
        if (bits) {   // the input relu's mask arrives as bits
            RT_MC(mc_gemm_relu_grad_bits(_weights[inpIdx].getW(), v.data(), _prev[inpIdx]->getActsGrad().data(), _in, cols, _out, 1, 0, _in, cols, cols,
                                         1.f, bits, ST()));
            return;
        }
        const float* mask = scaleTargets == 0.f ? claimReluGradMask(inpIdx) : nullptr;   // see FCLayer::bpropActs
        for (int ch = 0; ch < _channels; ch++) {
            const float* W = _weights[inpIdx].getW() + (size_t)ch * _out * _in;
            const float* G = v.data() + (size_t)ch * _out * cols;
            float* dX = _prev[inpIdx]->getActsGrad().data() + (size_t)ch * _in * cols;
            if (mask)   // the input is a relu's output: its gradient mask goes into this product's store
                RT_MC(mc_gemm_relu_grad(W, G, dX, _in, cols, _out, 1, 0, _in, cols, cols, 0.f, 1.f, mask + (size_t)ch * _in * cols, cols,
                                        wsb ? workspace(wsb) : nullptr, wsb, ST()));
            else
                RT_MC(mc_gemm(W, G, dX, _in, cols, _out, 1, 0, _in, cols, cols, scaleTargets, 1.f, nullptr, MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
        }
    }

    void bpropBiases(DevMat& v, PASS_TYPE passType) override {
        // the sum runs over pixels and cases, the scale is still eps / numCases (extralayers.cu:130-138)
        const int N = numCases(), P = v.rows() / (_channels * _out);
        const float scale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(N);
        RT_MC(mc_sum_rows(v.data(), _biases->getGrad(), _channels * _out, P * N, 0.f, scale, ST()));
    }

    bool fusesBiasGrad() const override { return true; }

    void bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases = false) override {
        const int N = numCases(), P = pixels(_prev[inpIdx]->getNumOutputs()), cols = P * N;
        const float scale = passType == PASS_GC ? 1.f : _weights[inpIdx].getEps() / _convNet->gradScaleCases(N);
        const float bscale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(N);
        const float scaleTargets = _weights[inpIdx].getNumUpdates() > 0 ? 1.f : 0.f;
        const size_t wsb = withBiases ? mc_gemm_row_sums_workspace_bytes(_out, _in, cols) : mc_gemm_workspace_bytes(_out, _in, cols);
        for (int ch = 0; ch < _channels; ch++) {
            const float* G = v.data() + (size_t)ch * _out * cols;
            const float* X = _prev[inpIdx]->getActs().data() + (size_t)ch * _in * cols;
            float* dW = _weights[inpIdx].getGrad() + (size_t)ch * _out * _in;
            // dW(out, in) = G(out, P*N) * X(in, P*N)^T; with the biases, db(out) = the row sums of G from the same pass
            if (withBiases) {
                RT_MC(mc_gemm_row_sums(G, X, dW, _out, _in, cols, cols, cols, _in, scaleTargets, scale, _biases->getGrad() + (size_t)ch * _out, bscale,
                                       wsb ? workspace(wsb) : nullptr, wsb, ST()));
                continue;
            }
            RT_MC(mc_gemm(G, X, dW, _out, _in, cols, 0, 1, cols, cols, _in, scaleTargets, scale, nullptr, MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
        }
    }

    int _channels, _in, _out;
};

// train: y = u < p ? 0 : x with a fresh uniform mask, the same mask routes the gradient;
// test and gradient check: y = (1 - p) * x.  No 1/(1-p) rescale at train time.
class DropoutNeuron : public Neuron {
public:
    explicit DropoutNeuron(PyObject* neuronParamsDict) {
        _prob = pyDictGetFloat(neuronParamsDict, "prob");
        if (!(_prob > 0 && _prob < 1)) throw std::runtime_error("dropout probability must be in (0, 1)");
    }

    void activate(DevMat& inputs, DevMat& outputs, PASS_TYPE passType) override {
        _activated = true;
        _fusedNow = false;
        _inputs = &inputs;
        _outputs = &outputs;
        if (&inputs != &outputs) outputs.resize(inputs);
        if (passType == PASS_TRAIN) {
            // the uniforms are a counter-based stream: forward and backward regenerate them from (seed, offset)
            // instead of writing and re-reading a mask matrix; an injected mask (tests) takes the stored path
            _counter = reserveUniform(inputs.numel(), &_seed, &_offset);
            if (_counter) {
                RT_MC(mc_dropout_counter(inputs.data(), outputs.data(), _prob, 0, inputs.numel(), _seed, _offset, ST()));
            } else {
                _mask.resize(inputs);
                fillUniform(_mask);
                RT_MC(mc_dropout_apply(inputs.data(), _mask.data(), outputs.data(), _prob, 0, inputs.numel(), ST()));
            }
        } else {
            RT_MC(mc_scale(outputs.data(), inputs.data(), 1.f - _prob, inputs.numel(), ST()));
        }
    }

    // the pool layer below applies this neuron in its own store (Neuron::beginFusedDropout): reserve the stream positions
    // activate() would have taken, in the same place of the pass, so that fused and separate runs draw the same masks
    bool beginFusedDropout(size_t numel, PASS_TYPE passType, float* prob, unsigned long long* seed, unsigned long long* offset) override {
        _fusedNow = false;
        if (passType != PASS_TRAIN) return false;
        if (!reserveUniform(numel, &_seed, &_offset)) return false;   // an injected mask is waiting: the stored-mask path takes it
        _activated = true;
        _counter = true;
        _fusedNow = true;
        *prob = _prob; *seed = _seed; *offset = _offset;
        return true;
    }
    bool fusedDropoutActive() const override { return _fusedNow; }

    void computeInputGrad(DevMat& actsGrad, DevMat& target, bool add, PASS_TYPE passType) override {
        if (!add && &target != &actsGrad) target.resize(actsGrad);
        if (passType == PASS_TRAIN && _counter) {
            RT_MC(mc_dropout_counter(actsGrad.data(), target.data(), _prob, add ? 1 : 0, actsGrad.numel(), _seed, _offset, ST()));
        } else if (passType == PASS_TRAIN) {
            RT_MC(mc_dropout_apply(actsGrad.data(), _mask.data(), target.data(), _prob, add ? 1 : 0, actsGrad.numel(), ST()));
        } else if (add) {
            RT_MC(mc_axpby(target.data(), actsGrad.data(), 1.f, 1.f - _prob, actsGrad.numel(), ST()));
        } else {
            RT_MC(mc_scale(target.data(), actsGrad.data(), 1.f - _prob, actsGrad.numel(), ST()));
        }
    }

private:
    DevMat _mask;
    float _prob;
    bool _counter = false, _fusedNow = false;
    unsigned long long _seed = 0, _offset = 0;
};

// Group sparsity in the label domain: inputs (labels, acts); for every channel and every distinct label of the
// minibatch S = sqrt(count * sum of acts^2 over the label's cases and all pixels); cost = sum of S
// (extralayers.cu:10-32), gradient -coeff * count * acts / (S + 2.5e-7) into the second input (:34-38).
class GroupSparsityInLabelCostLayer : public CostLayer {
public:
    GroupSparsityInLabelCostLayer(ConvNet* convNet, PyObject* paramsDict) : CostLayer(convNet, paramsDict) {
        _channels = pyDictGetInt(paramsDict, "channels");
        _imgSize = pyDictGetInt(paramsDict, "imgSize");
    }
    int numCostValues() const override { return 1; }

protected:
    void fpropActs(int inpIdx, float, PASS_TYPE) override {
        if (inpIdx != 0) return;   // both inputs are used together, once
        DevMat& labels = *_inputs[0];
        DevMat& acts = *_inputs[1];
        const int N = labels.cols();
        if (acts.rows() % _channels) throw std::runtime_error("cost.gsinlabel '" + _name + "': input rows are not a multiple of channels");
        _pixels = acts.rows() / _channels;
        _sumsq.resize(_channels, N);
        _sqrts.resize(_channels, N);
        _terms.resize(_channels, N);
        _counts.resize(1, N);
        RT_MC(mc_group_sparsity_in_label_cost(labels.data(), acts.data(), _sumsq.data(), _sqrts.data(), _counts.data(), _terms.data(), N, _channels,
                                              _pixels, ST()));
        RT_MC(mc_sum_all(_terms.data(), _terms.numel(), costSlot(), ST()));
    }
    void bpropActs(DevMat&, int inpIdx, float scaleTargets, PASS_TYPE) override {
        if (inpIdx != 1) throw std::runtime_error("cost.gsinlabel: gradient requested for the label input");
        DevMat& acts = *_inputs[1];
        RT_MC(mc_group_sparsity_in_label_grad(acts.data(), _sqrts.data(), _counts.data(), _prev[1]->getActsGrad().data(), acts.cols(), _channels, _pixels,
                                              scaleTargets, -_coeff, ST()));
    }
    int _channels, _imgSize, _pixels = 0;
    DevMat _sumsq, _sqrts, _terms, _counts;
};

static Layer* CreateCCCPLayer(ConvNet* convNet, PyObject* paramsDict) { return new CCCPLayer(convNet, paramsDict); }
static Neuron* CreateDropoutNeuron(PyObject* neuronParamsDict) { return new DropoutNeuron(neuronParamsDict); }
static Layer* CreateGroupSparsityInLabelCostLayer(ConvNet* convNet, PyObject* paramsDict) { return new GroupSparsityInLabelCostLayer(convNet, paramsDict); }

extern "C" __attribute__((visibility("default"))) std::map<std::string, layerConFunc> layerConstructor() {
    std::map<std::string, layerConFunc> ret;
    ret["cccp"] = &CreateCCCPLayer;
    ret["cost.gsinlabel"] = &CreateGroupSparsityInLabelCostLayer;
    return ret;
}

extern "C" __attribute__((visibility("default"))) std::map<std::string, neuronConFunc> neuronConstructor() {
    std::map<std::string, neuronConFunc> ret;
    ret["dropout"] = &CreateDropoutNeuron;
    return ret;
}
