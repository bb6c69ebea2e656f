// layer.cpp -- layer graph semantics (reference: Kernel/src/layers/layer.cu, cited per block).
#include "layer.h"

#include <string.h>

#include "convnet.h"

namespace rt {

static inline mcStream_t ST() { return reinterpret_cast<mcStream_t>(computeStream()); }

// ==========================================================================================
// Layer                                                                    (layer.cu:45-244)
// ==========================================================================================
Layer::Layer(ConvNet* convNet, PyObject* paramsDict) : _convNet(convNet) {
    _name = pyDictGetString(paramsDict, "name");
    _type = pyDictGetString(paramsDict, "type");
    _gradConsumer = pyDictGetInt(paramsDict, "gradConsumer") != 0;
    _actsTarget = pyDictGetInt(paramsDict, "actsTarget");
    _actsGradTarget = pyDictGetInt(paramsDict, "actsGradTarget");
    _conserveMem = pyDictGetInt(paramsDict, "conserveMem") != 0;
    _numOutputs = pyDictHas(paramsDict, "outputs") ? pyDictGetInt(paramsDict, "outputs") : 0;
    _outputs = _actsTarget < 0 ? &_ownOutputs : nullptr;
    _actsGrad = _actsGradTarget < 0 ? &_ownActsGrad : nullptr;
}

void Layer::fpropNext(PASS_TYPE passType) {
    for (Layer* l : _next) l->fprop(passType);
}

void Layer::truncBwdActs() {
    if (_conserveMem && _actsGradTarget < 0) getActsGrad().free();
    if (_conserveMem && _actsTarget < 0) getActs().free();
}

void Layer::fprop(PASS_TYPE passType) {
    _rcvdFInputs += 1;
    if (_rcvdFInputs == (int)_prev.size()) {
        DevMatV v;
        for (Layer* p : _prev) v.push_back(&p->getActs());
        fprop(v, passType);
    }
}

void Layer::fprop(DevMat& v, PASS_TYPE passType) {
    DevMatV vl;
    vl.push_back(&v);
    fprop(vl, passType);
}

void Layer::fprop(DevMatV& v, PASS_TYPE passType) {
    if (v.size() != _prev.size()) throw std::runtime_error(fmt("layer '%s': %zu inputs given, %zu expected", _name.c_str(), v.size(), _prev.size()));
    _inputs = v;
    _outputs = _actsTarget < 0 ? &_ownOutputs : _inputs[_actsTarget];
    _rcvdFInputs = (int)_prev.size();
    _numCases = v.empty() ? 0 : v[0]->cols();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i]->cols() != _numCases || (_prev[i]->getNumOutputs() > 0 && v[i]->rows() != _prev[i]->getNumOutputs()))
            throw std::runtime_error(fmt("layer '%s': input %zu is %dx%d, expected %dx%d", _name.c_str(), i, v[i]->rows(), v[i]->cols(),
                                         _prev[i]->getNumOutputs(), _numCases));
    if (_actsTarget < 0 && _numOutputs > 0) _ownOutputs.resize(_numOutputs, _numCases);

    profileSetLayer(_name.c_str());
    profileSetPhase("fwd");
    // the input whose matrix this layer writes into goes first, the others accumulate
    if (_actsTarget >= 0) fpropActs(_actsTarget, 0, passType);
    for (int i = 0; i < (int)_prev.size(); i++)
        if (i != _actsTarget) fpropActs(i, (_actsTarget >= 0 || i > 0) ? 1.f : 0.f, passType);
    fpropNext(passType);
}

void Layer::bprop(PASS_TYPE passType) {
    if (_rcvdBInputs == _numGradProducersNext) {
        _rcvdBInputs++;   // guards against a second run
        bprop(getActsGrad(), passType);
    }
}

void Layer::bprop(DevMat& v, PASS_TYPE passType) {
    profileSetLayer(_name.c_str());
    profileSetPhase("wgrad");
    bpropCommon(v, passType);
    profileSetPhase("bwd");
    if (isGradProducer() && bpropOverwritesActs() && _convNet->forksWeightGrads()) _convNet->joinWeightGradBranch();
    if (isGradProducer()) {
        auto send = [&](int i) {
            Layer* p = _prev[i];
            DevMat& tgt = p->getActsGrad();
            const bool accumulate = p->getRcvdBInputs() > 0;
            if (!accumulate || tgt.rows() != p->getNumOutputs() || tgt.cols() != _numCases) {
                if (accumulate) throw std::runtime_error(fmt("layer '%s': gradient of '%s' changed shape while accumulating", _name.c_str(), p->getName().c_str()));
                tgt.resize(p->getNumOutputs(), _numCases);
            }
            profileSetLayer(_name.c_str());
            bpropActs(v, i, accumulate ? 1.f : 0.f, passType);
            p->incRcvdBInputs();
        };
        // first the inputs whose gradient matrix is certainly not this layer's own ...
        for (int i = 0; i < (int)_prev.size(); i++)
            if (_prev[i]->isGradConsumer() && _actsGradTarget != i) send(i);
        // ... then the one it shares (the write may be in place)
        if (_actsGradTarget >= 0 && _prev[_actsGradTarget]->isGradConsumer()) send(_actsGradTarget);
    }
    truncBwdActs();
    if (isGradProducer())
        for (Layer* p : _prev)
            if (p->isGradConsumer()) p->bprop(passType);
}

void Layer::reset() {
    _rcvdFInputs = 0;
    _rcvdBInputs = 0;
}

void Layer::addNext(Layer* l) {
    _next.push_back(l);
    _numGradProducersNext += l->isGradProducer();
}

void Layer::postInit() {
    _actsGrad = _actsGradTarget < 0 ? &_ownActsGrad : &_prev.at(_actsGradTarget)->getActsGrad();
}

bool Layer::isGradConsumer() {
    if (!_foundGradConsumers) {
        for (Layer* p : _prev) _gradConsumer |= p->isGradConsumer();
        _foundGradConsumers = true;
    }
    return _gradConsumer;
}

DevMat& Layer::getActs() {
    if (!_outputs) throw std::runtime_error(fmt("layer '%s' has no output yet", _name.c_str()));
    return *_outputs;
}

DevMat& Layer::getActsGrad() {
    if (!_actsGrad) throw std::runtime_error(fmt("layer '%s' has no gradient matrix", _name.c_str()));
    return *_actsGrad;
}

// ==========================================================================================
// Neurons                                               (neuron.cuh:65-462, neuron.cu:33-88)
// ==========================================================================================
void Neuron::activate(DevMat& inputs, DevMat& outputs, PASS_TYPE) {
    _activated = true;
    _inputs = &inputs;
    _outputs = &outputs;
    if (&inputs != &outputs) outputs.resize(inputs);
    if (_mcType == MC_N_IDENT) {
        if (&inputs != &outputs) outputs.copyFrom(inputs);
        return;
    }
    RT_MC(mc_neuron_activate(_mcType, _a, _b, inputs.data(), outputs.data(), inputs.numel(), ST()));
}

void Neuron::computeInputGrad(DevMat& actsGrad, DevMat& target, bool add, PASS_TYPE) {
    if (!_activated) throw std::runtime_error("neuron gradient requested before activation");
    if (!add && &target != &actsGrad) target.resize(actsGrad);
    if (_mcType == MC_N_IDENT) {
        if (&target == &actsGrad) return;
        if (!add) target.copyFrom(actsGrad);
        else RT_MC(mc_axpby(target.data(), actsGrad.data(), 1.f, 1.f, target.numel(), ST()));
        return;
    }
    const DevMat* ref = _usesInputs ? _inputs : _outputs;
    RT_MC(mc_neuron_input_grad(_mcType, _a, _b, actsGrad.data(), ref->data(), target.data(), add ? 1 : 0, actsGrad.numel(), ST()));
}

class BuiltinNeuron : public Neuron {
public:
    BuiltinNeuron(int type, float a, float b, bool usesInputs) {
        _mcType = type;
        _a = a;
        _b = b;
        _usesInputs = usesInputs;
    }
    bool isRelu() const override { return _mcType == MC_N_RELU; }
};

Neuron& Neuron::makeNeuron(PyObject* neuronDict) {
    if (!neuronDict) throw std::runtime_error("neuron layer without a 'neuron' entry");
    const std::string type = pyDictGetString(neuronDict, "type");
    PyObject* params = PyDict_GetItemString(neuronDict, "params");
    auto P = [&](const char* k) { return pyDictGetFloat(params, k); };
    if (type == "relu") return *new BuiltinNeuron(MC_N_RELU, 0, 0, false);
    if (type == "softrelu") return *new BuiltinNeuron(MC_N_SOFTRELU, 0, 0, false);
    if (type == "brelu") return *new BuiltinNeuron(MC_N_BRELU, P("a"), 0, false);
    if (type == "abs") return *new BuiltinNeuron(MC_N_ABS, 0, 0, true);
    if (type == "logistic") return *new BuiltinNeuron(MC_N_LOGISTIC, 0, 0, false);
    if (type == "tanh") return *new BuiltinNeuron(MC_N_TANH, P("a"), P("b"), false);
    if (type == "square") return *new BuiltinNeuron(MC_N_SQUARE, 0, 0, true);
    if (type == "sqrt") return *new BuiltinNeuron(MC_N_SQRT, 0, 0, false);
    if (type == "linear") return *new BuiltinNeuron(MC_N_LINEAR, P("a"), P("b"), false);
    if (type == "ident") return *new Neuron();
    auto it = pluginNeurons().find(type);
    if (it != pluginNeurons().end()) return *it->second(params);
    throw std::runtime_error("Unknown neuron type: " + type);
}

NeuronLayer::NeuronLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _neuron = &Neuron::makeNeuron(PyDict_GetItemString(paramsDict, "neuron"));
}

void NeuronLayer::fpropActs(int, float, PASS_TYPE passType) {
    if (_fused) return;   // the producing contraction already applied relu in its epilogue, in place
    if (_dropFwd || _reluFwd) {   // the pool layer below wrote this layer's output in its own store (fuseDropoutIntoPool / fuseReluIntoPool)
        _dropFwd = _reluFwd = false;
        return;
    }
    _neuron->activate(*_inputs[0], getActs(), passType);
}

bool NeuronLayer::fuseDropoutIntoPool(size_t numel, PASS_TYPE passType, float* prob, unsigned long long* seed, unsigned long long* offset) {
    static const bool off = getenv("MC_NO_FUSE_DROPOUT") != nullptr;
    _dropFwd = _dropBwd = false;
    // its own output matrix (the pool's output stays the un-dropped tensor the max-undo routes by), one input
    if (off || passType != PASS_TRAIN || _actsTarget >= 0 || _prev.size() != 1) return false;
    if (!_neuron->beginFusedDropout(numel, passType, prob, seed, offset)) return false;
    _dropFwd = true;
    // backward: only when this layer hands the gradient down in place (the pool then reads the very matrix the layer above
    // wrote) and has a single gradient producer (nothing accumulates into it)
    _dropBwd = _actsGradTarget == 0 && _numGradProducersNext == 1;
    return true;
}

bool NeuronLayer::fuseReluIntoPool(PASS_TYPE passType) {
    static const bool off = getenv("MC_NO_FUSE_POOL_RELU") != nullptr;
    _reluBwd = false;
    if (off || !_neuron->isRelu() || _prev.size() != 1 || _actsTarget > 0) return false;   // in place (_actsTarget == 0) or a matrix of its own
    _reluFwd = true;
    _bitsValid = false;
    _reluBwd = passType != PASS_TEST && _actsGradTarget == 0 && _numGradProducersNext == 1;
    return true;
}

bool NeuronLayer::canFuseGradMask() {
    static const bool off = getenv("MC_NO_FUSE_RELU_GRAD") != nullptr;
    // in place on the gradient side (the matrix the consumer writes IS the one this layer hands down), a
    // single gradient producer (nothing to accumulate), and somebody below who wants the gradient
    return !off && _neuron->isRelu() && _actsGradTarget == 0 && _numGradProducersNext == 1 && _prev[0]->isGradConsumer();
}

const float* Layer::claimReluGradMask(int inpIdx) {
    NeuronLayer* nl = dynamic_cast<NeuronLayer*>(_prev[inpIdx]);
    if (!nl || !nl->canFuseGradMask()) return nullptr;
    nl->setGradMasked();
    return nl->getActs().data();
}

bool NeuronLayer::wantsBits(int rows, int ld, int numCases) {
    static const bool off = getenv("MC_NO_RELU_BITS") != nullptr;
    return !off && _next.size() == 1 && _next[0]->readsReluBits(rows, ld, numCases) && canFuseGradMask();
}

unsigned* NeuronLayer::bitsBuffer(int rows, int ld) {
    _bits.resize((rows + 31) / 32, ld);
    return reinterpret_cast<unsigned*>(_bits.data());
}

const unsigned* Layer::claimReluGradBits(int inpIdx) {
    NeuronLayer* nl = dynamic_cast<NeuronLayer*>(_prev[inpIdx]);
    if (!nl || !nl->canFuseGradMask() || !nl->validBits()) return nullptr;
    nl->setGradMasked();
    return nl->validBits();
}

void NeuronLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE passType) {
    if (_dropBwd && _neuron->fusedDropoutActive() && passType == PASS_TRAIN && scaleTargets == 0.f) return;   // the pool's max-undo applies the keep bytes
    _dropBwd = false;
    if (_reluBwd && scaleTargets == 0.f) {   // the pool's max-undo gates the gradient by its relu'd output
        _gradMasked = false;
        return;
    }
    _reluBwd = false;
    if (_gradMasked) {   // the layer above stored grad * (act > 0) straight into the shared gradient matrix
        _gradMasked = false;
        return;
    }
    if (_fused) {   // the fused path never ran activate(); the output IS the input matrix
        DevMat& y = getActs();
        DevMat& tgt = _prev[0]->getActsGrad();
        RT_MC(mc_neuron_input_grad(MC_N_RELU, 0, 0, v.data(), y.data(), tgt.data(), scaleTargets > 0 ? 1 : 0, v.numel(), ST()));
        return;
    }
    _neuron->computeInputGrad(v, _prev[0]->getActsGrad(), scaleTargets > 0, passType);
}

// ==========================================================================================
// Weights                                                              (weights.cuh:41-219)
// ==========================================================================================
Weights::Weights(HostMat& hW, HostMat& hInc, float eps, float wc, float mom, bool deviceFortran)
    : _hW(&hW), _hInc(&hInc), _eps(eps), _wc(wc), _mom(mom), _devFortran(deviceFortran) {
    if (hW.rows() != hInc.rows() || hW.cols() != hInc.cols()) throw std::runtime_error("weights and weightsInc differ in shape");
    _contributors = eps > 0 ? 1 : 0;
}

Weights::Weights(Weights& src, float eps)
    : _hW(nullptr), _hInc(nullptr), _src(&src), _eps(eps), _wc(0), _mom(src.getMom()) {
    _hW = src._hW;
    _hInc = src._hInc;
    if (eps > 0) const_cast<Weights&>(src.owner())._contributors++;
}

Weights::~Weights() {
    if (!_src) {
        delete _hW;
        delete _hInc;
    }
}

void Weights::releaseHost() {
    if (!_src) {
        _hW->release();
        _hInc->release();
    }
}

void Weights::bind(float* w, float* inc, float* grad, size_t offset) {
    _w = w;
    _inc = inc;
    _grad = grad;
    _offset = offset;
}

// host ndarray bytes -> device layout (row-major, or column-major when _devFortran)
static void to_device_order(const HostMat& h, bool devFortran, std::vector<float>& buf) {
    buf.resize(h.numel());
    if (h.fortran() == devFortran) {
        memcpy(buf.data(), h.data(), h.numel() * sizeof(float));
        return;
    }
    std::vector<float> rm(h.numel());
    h.toRowMajor(rm.data());
    if (!devFortran) {
        buf.swap(rm);
        return;
    }
    for (int r = 0; r < h.rows(); ++r)
        for (int c = 0; c < h.cols(); ++c) buf[(size_t)c * h.rows() + r] = rm[(size_t)r * h.cols() + c];
}

static void from_device_order(HostMat& h, bool devFortran, const std::vector<float>& buf) {
    if (h.fortran() == devFortran) {
        memcpy(h.data(), buf.data(), h.numel() * sizeof(float));
        return;
    }
    if (!devFortran) {
        h.fromRowMajor(buf.data());
        return;
    }
    std::vector<float> rm(h.numel());
    for (int r = 0; r < h.rows(); ++r)
        for (int c = 0; c < h.cols(); ++c) rm[(size_t)r * h.cols() + c] = buf[(size_t)c * h.rows() + r];
    h.fromRowMajor(rm.data());
}

void Weights::copyToGPU() {
    if (_src) return;
    if (!_w) throw std::runtime_error("Weights::copyToGPU before arena binding");
    std::vector<float> buf;
    to_device_order(*_hW, _devFortran, buf);
    RT_HIP(hipMemcpy(_w, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    to_device_order(*_hInc, _devFortran, buf);
    RT_HIP(hipMemcpy(_inc, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
}

void Weights::copyToCPU() {
    if (_src) return;
    std::vector<float> buf(numel());
    RT_HIP(hipStreamSynchronize(computeStream()));
    RT_HIP(hipMemcpy(buf.data(), _w, buf.size() * sizeof(float), hipMemcpyDeviceToHost));
    from_device_order(*_hW, _devFortran, buf);
    RT_HIP(hipMemcpy(buf.data(), _inc, buf.size() * sizeof(float), hipMemcpyDeviceToHost));
    from_device_order(*_hInc, _devFortran, buf);
}

void Weights::update() {
    // only the true owner updates, and only tensors that learn (weights.cuh:153)
    if (_src == nullptr && _eps > 0) {
        RT_MC(mc_weights_update(_w, _inc, _grad, numel(), _mom, _eps, _wc, ST()));
        _numUpdates = 0;
    }
}

// ==========================================================================================
// WeightLayer                                                            (layer.cu:269-361)
// ==========================================================================================
WeightLayer::WeightLayer(ConvNet* convNet, PyObject* paramsDict, bool deviceFortran) : Layer(convNet, paramsDict) {
    HostMatV* hWeights = pyDictGetMatrixV(paramsDict, "weights");
    HostMatV* hWeightsInc = pyDictGetMatrixV(paramsDict, "weightsInc");
    HostMat* hBiases = pyDictGetMatrix(paramsDict, "biases");
    HostMat* hBiasesInc = pyDictGetMatrix(paramsDict, "biasesInc");
    floatv* momW = pyDictGetFloatV(paramsDict, "momW");
    floatv* epsW = pyDictGetFloatV(paramsDict, "epsW");
    floatv* wc = pyDictGetFloatV(paramsDict, "wc");
    const float momB = pyDictGetFloat(paramsDict, "momB"), epsB = pyDictGetFloat(paramsDict, "epsB");
    intv* srcLayers = pyDictGetIntV(paramsDict, "weightSourceLayerIndices");
    intv* srcMats = pyDictGetIntV(paramsDict, "weightSourceMatrixIndices");
    if (!momW || !epsW || !wc || !srcLayers || !srcMats) throw std::runtime_error(fmt("layer '%s': missing weight hyper-parameters", _name.c_str()));
    const size_t n = hWeights->size();
    if (hWeightsInc->size() != n || momW->size() != n || epsW->size() != n || wc->size() != n || srcLayers->size() != n || srcMats->size() != n)
        throw std::runtime_error(fmt("layer '%s': per-input weight lists differ in length", _name.c_str()));

    for (size_t i = 0; i < n; i++) {
        const int srcLayerIdx = srcLayers->at(i), matrixIdx = srcMats->at(i);
        Weights* w;
        if (srcLayerIdx == convNet->getNumLayers()) {          // another matrix of this very layer
            w = new Weights(_weights[matrixIdx], epsW->at(i));
        } else if (srcLayerIdx >= 0) {                           // a matrix of an earlier layer
            WeightLayer* src = dynamic_cast<WeightLayer*>(&convNet->getLayer(srcLayerIdx));
            if (!src) throw std::runtime_error(fmt("layer '%s': weight source layer %d has no weights", _name.c_str(), srcLayerIdx));
            w = new Weights(src->getWeights(matrixIdx), epsW->at(i));
        } else {
            w = new Weights(*hWeights->at(i), *hWeightsInc->at(i), epsW->at(i), wc->at(i), momW->at(i), deviceFortran);
            (*hWeights)[i] = nullptr;
            (*hWeightsInc)[i] = nullptr;
        }
        _weights.addWeights(*w);
        if (w->isOwner()) convNet->registerWeights(w);
    }
    // views of shared matrices were only needed for the shape checks Python already did
    for (HostMatV* v : {hWeights, hWeightsInc})
        for (HostMat* m : *v)
            if (m) { m->release(); delete m; }
    _biases = new Weights(*hBiases, *hBiasesInc, epsB, 0, momB, deviceFortran);
    convNet->registerWeights(_biases);
    delete hWeights; delete hWeightsInc; delete momW; delete epsW; delete wc; delete srcLayers; delete srcMats;
}

void WeightLayer::bpropCommon(DevMat& v, PASS_TYPE passType) {
    // The weight-gradient product of input 0 reads all of v; layers that can have it deliver the bias gradient (a sum
    // over v) from that pass instead of reading v once more (MC_NO_FUSED_BIAS=1: the separate reduction, for A/B runs).
    static const bool noFuse = getenv("MC_NO_FUSED_BIAS") != nullptr;
    const bool biases = _biases->getEps() > 0;
    const bool fused = biases && !noFuse && fusesBiasGrad() && _weights.getSize() > 0 && _weights[0].getEps() > 0;
    // When the step forks (convnet.cpp, "One training step") these products go to the weight-gradient branch: they read v and
    // the inputs, which nothing below this layer writes (not so if this layer's gradient were written in place).  A layer
    // with nothing below it to overlap with (the first layer of a net: its input takes no gradient) stays on the main stream
    // instead and runs beside whatever the branch still has queued -- unless it shares a weight tensor, whose other
    // contributions are accumulated on the branch.
    const bool forks = passType == PASS_TRAIN && _convNet->forksWeightGrads();
    bool below = false, shares = false;
    for (Layer* p : _prev) below |= p->isGradConsumer();
    for (int i = 0; i < _weights.getSize(); i++) shares |= _weights[i].contributors() > 1;
    const bool onBranch = forks && _actsGradTarget < 0 && (below || shares);
    const bool besideBranch = forks && _actsGradTarget < 0 && !onBranch;
    struct Branch {
        ConvNet* net;
        explicit Branch(ConvNet* n) : net(n) { if (net) net->beginWeightGradBranch(); }
        ~Branch() { if (net) net->endWeightGradBranch(); }
    } branch(onBranch ? _convNet : nullptr);
    // (a layer that cannot fork queues on the main stream what earlier layers queued on the branch: shared tensors assume one
    // producer stream at a time, so the branch is joined first)
    if (forks && !onBranch && !besideBranch) _convNet->joinWeightGradBranch();
    if (biases && !fused) {
        bpropBiases(v, passType);
        _convNet->gradientReady(*_biases);
    }
    for (int i = 0; i < _weights.getSize(); i++) {
        if (_weights[i].getEps() > 0) {
            bpropWeights(v, i, passType, fused && i == 0);
            _weights[i].incNumUpdates();
            if (fused && i == 0) _convNet->gradientReady(*_biases);
            _convNet->gradientReady(_weights[i]);
        }
    }
}

void WeightLayer::updateWeights() {
    profileSetLayer(_name.c_str());
    _weights.update();
    _biases->update();
}

void WeightLayer::copyToCPU() {
    _weights.copyToCPU();
    _biases->copyToCPU();
}

void WeightLayer::copyToGPU() {
    _weights.copyToGPU();
    _biases->copyToGPU();
}

void WeightLayer::releaseHost() {
    for (int i = 0; i < _weights.getSize(); i++) _weights[i].releaseHost();
    _biases->releaseHost();
}

void WeightLayer::checkGradients() {
    for (int i = 0; i < _weights.getSize(); i++) _convNet->checkGradient(_name + " weights[" + std::to_string(i) + "]", _wStep, _weights[i]);
    _convNet->checkGradient(_name + " biases", _bStep, *_biases);
}

NeuronLayer* WeightLayer::fusableRelu() {
    if (_prev.size() != 1 || _next.size() != 1) return nullptr;
    NeuronLayer* nl = dynamic_cast<NeuronLayer*>(_next[0]);
    if (!nl || !nl->inPlace() || !nl->getNeuron().isRelu()) return nullptr;
    return nl;
}

float* WeightLayer::workspace(size_t bytes) { return _convNet->workspace(bytes); }

// ==========================================================================================
// FCLayer                                                                (layer.cu:368-409)
// Device weights are (outputs, inputs) row-major = the Fortran-ordered (inputs, outputs) ndarray.
// acts(out, N) = W(out, in) * x(in, N) + b
// ==========================================================================================
FCLayer::FCLayer(ConvNet* convNet, PyObject* paramsDict) : WeightLayer(convNet, paramsDict, true) {
    _wStep = 0.1f;
    _bStep = 0.01f;
}

void FCLayer::fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) {
    const int N = numCases(), in = _inputs[inpIdx]->rows(), out = _numOutputs;
    NeuronLayer* relu = fusableRelu();
    if (relu) relu->setFusedIntoProducer(true);
    const size_t wsb = mc_gemm_workspace_bytes(out, N, in);
    // a relu whose gradient mask the layer above can read as bits: write them from this product's epilogue (unsplit products only)
    const bool bits = relu && passType == PASS_TRAIN && scaleTargets == 0 && wsb == 0 && relu->wantsBits(out, N, N);
    if (relu) relu->setBitsValid(bits);
    if (bits)
        RT_MC(mc_gemm_relu_bits(_weights[inpIdx].getW(), _inputs[inpIdx]->data(), getActs().data(), out, N, in, 0, 0, in, N, N, _biases->getW(),
                                relu->bitsBuffer(out, N), ST()));
    else
        RT_MC(mc_gemm(_weights[inpIdx].getW(), _inputs[inpIdx]->data(), getActs().data(), out, N, in, 0, 0, in, N, N, scaleTargets, 1.f,
                      scaleTargets == 0 ? _biases->getW() : nullptr, relu ? MC_NEURON_RELU : MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
}

void FCLayer::bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE) {
    const int N = numCases(), in = _prev[inpIdx]->getNumOutputs(), out = _numOutputs;
    const size_t wsb = mc_gemm_workspace_bytes(in, N, out);
    // the relu gradient mask of the layer below goes into this product's store (one pass over the gradient instead of
    // three); measured +0.9 % on the NIN step once the epilogue's mask reads became branch-free buffer loads
    const unsigned* bits = scaleTargets == 0.f ? claimReluGradBits(inpIdx) : nullptr;
    const float* mask = (scaleTargets == 0.f && !bits) ? claimReluGradMask(inpIdx) : nullptr;
    if (bits)
        RT_MC(mc_gemm_relu_grad_bits(_weights[inpIdx].getW(), v.data(), _prev[inpIdx]->getActsGrad().data(), in, N, out, 1, 0, in, N, N, 1.f, bits, ST()));
    else if (mask)
        RT_MC(mc_gemm_relu_grad(_weights[inpIdx].getW(), v.data(), _prev[inpIdx]->getActsGrad().data(), in, N, out, 1, 0, in, N, N, 0.f, 1.f, mask, N,
                                wsb ? workspace(wsb) : nullptr, wsb, ST()));
    else
        RT_MC(mc_gemm(_weights[inpIdx].getW(), v.data(), _prev[inpIdx]->getActsGrad().data(), in, N, out, 1, 0, in, N, N, scaleTargets, 1.f, nullptr,
                      MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
}

void FCLayer::bpropBiases(DevMat& v, PASS_TYPE passType) {
    const float scale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(numCases());
    RT_MC(mc_sum_rows(v.data(), _biases->getGrad(), _numOutputs, numCases(), 0.f, scale, ST()));
}

void FCLayer::bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases) {
    const int N = numCases(), in = _prev[inpIdx]->getNumOutputs(), out = _numOutputs;
    const float scale = passType == PASS_GC ? 1.f : _weights[inpIdx].getEps() / _convNet->gradScaleCases(N);
    const float scaleTargets = _weights[inpIdx].getNumUpdates() > 0 ? 1.f : 0.f;
    // dW(out, in) = g(out, N) * x(in, N)^T
    if (withBiases) {   // ... and db(out) = the row sums of g
        const float bscale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(N);
        const size_t wsb = mc_gemm_row_sums_workspace_bytes(out, in, N);
        RT_MC(mc_gemm_row_sums(v.data(), _prev[inpIdx]->getActs().data(), _weights[inpIdx].getGrad(), out, in, N, N, N, in, scaleTargets, scale,
                               _biases->getGrad(), bscale, wsb ? workspace(wsb) : nullptr, wsb, ST()));
        return;
    }
    const size_t wsb = mc_gemm_workspace_bytes(out, in, N);
    RT_MC(mc_gemm(v.data(), _prev[inpIdx]->getActs().data(), _weights[inpIdx].getGrad(), out, in, N, 0, 1, N, N, in, scaleTargets, scale, nullptr,
                  MC_NEURON_NONE, wsb ? workspace(wsb) : nullptr, wsb, ST()));
}

// ==========================================================================================
// LocalLayer / ConvLayer / LocalUnsharedLayer                             (layer.cu:416-602)
// ==========================================================================================
LocalLayer::LocalLayer(ConvNet* convNet, PyObject* paramsDict) : WeightLayer(convNet, paramsDict, false) {
    _padding = pyDictGetIntV(paramsDict, "padding");
    _stride = pyDictGetIntV(paramsDict, "stride");
    _filterSize = pyDictGetIntV(paramsDict, "filterSize");
    _channels = pyDictGetIntV(paramsDict, "channels");
    _imgSize = pyDictGetIntV(paramsDict, "imgSize");
    _groups = pyDictGetIntV(paramsDict, "groups");
    _filterChannels = pyDictGetIntV(paramsDict, "filterChannels");
    _randSparse = pyDictGetIntV(paramsDict, "randSparse");
    _overSample = pyDictGetIntV(paramsDict, "overSample");
    _filterPixels = pyDictGetIntV(paramsDict, "filterPixels");
    _imgPixels = pyDictGetIntV(paramsDict, "imgPixels");
    _numFilters = pyDictGetInt(paramsDict, "filters");
    _modulesX = pyDictGetInt(paramsDict, "modulesX");
    _modules = pyDictGetInt(paramsDict, "modules");
    const size_t ni = _randSparse->size();
    _conns.resize(ni);
    _inverse.resize(ni);
    _dConns.assign(ni, nullptr);
    _dInverse.assign(ni, nullptr);
    _gathered.assign(ni, nullptr);
    _gatheredGrad.assign(ni, nullptr);
    PyObject* pyConns = PyDict_GetItemString(paramsDict, "filterConns");
    for (size_t i = 0; i < ni; i++) {
        if (!_randSparse->at(i)) continue;
        PyObject* li = (pyConns && PyList_Check(pyConns) && (size_t)PyList_GET_SIZE(pyConns) > i) ? PyList_GET_ITEM(pyConns, i) : nullptr;
        const int C = _channels->at(i), K = _groups->at(i) * _filterChannels->at(i), O = _overSample->at(i);
        if (!li || !PyList_Check(li) || PyList_GET_SIZE(li) != K || O < 1 || O * C != K)
            throw std::runtime_error(fmt("layer '%s'[%zu]: randSparse needs filterConns of length groups*filterChannels = %d = overSample*channels", _name.c_str(), i, K));
        std::vector<int>& conns = _conns[i];
        std::vector<int>& inv = _inverse[i];
        conns.resize(K);
        inv.assign((size_t)C * O, -1);
        std::vector<int> seen(C, 0);
        for (int k = 0; k < K; k++) {
            const long c = PyLong_AsLong(PyList_GET_ITEM(li, k));
            if (c < 0 || c >= C) throw std::runtime_error(fmt("layer '%s'[%zu]: filterConns[%d] = %ld is not a channel index", _name.c_str(), i, k, c));
            conns[k] = (int)c;
            // every block of `channels` consecutive entries is a permutation (layer.py:860): each channel is copied
            // exactly overSample times, in ascending k
            if (seen[c] >= O) throw std::runtime_error(fmt("layer '%s'[%zu]: channel %ld is connected more than overSample = %d times", _name.c_str(), i, c, O));
            inv[(size_t)c * O + seen[c]++] = k;
        }
        _gathered[i] = new DevMat();
        _gatheredGrad[i] = new DevMat();
    }
}

LocalLayer::~LocalLayer() {
    for (intv* v : {_padding, _stride, _filterSize, _channels, _imgSize, _groups, _filterChannels, _randSparse, _overSample, _filterPixels, _imgPixels}) delete v;
    for (int* p : _dConns) if (p) (void)hipFree(p);
    for (int* p : _dInverse) if (p) (void)hipFree(p);
    for (DevMat* m : _gathered) delete m;
    for (DevMat* m : _gatheredGrad) delete m;
}

void LocalLayer::copyToGPU() {
    WeightLayer::copyToGPU();
    for (size_t i = 0; i < _conns.size(); i++) {
        if (_conns[i].empty() || _dConns[i]) continue;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_dConns[i]), _conns[i].size() * sizeof(int)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_dInverse[i]), _inverse[i].size() * sizeof(int)));
        RT_HIP(hipMemcpy(_dConns[i], _conns[i].data(), _conns[i].size() * sizeof(int), hipMemcpyHostToDevice));
        RT_HIP(hipMemcpy(_dInverse[i], _inverse[i].data(), _inverse[i].size() * sizeof(int), hipMemcpyHostToDevice));
    }
}

const float* LocalLayer::sparseInput(int i, const DevMat& input) {
    DevMat& g = *_gathered[i];
    g.resize(sparseChannels(i) * _imgPixels->at(i), input.cols());
    RT_MC(mc_conv_gather_channels(input.data(), _dConns[i], g.data(), input.cols(), _channels->at(i), sparseChannels(i), _imgPixels->at(i), ST()));
    return g.data();
}

void LocalLayer::sparseScatter(int i, const DevMat& gathered, DevMat& target, float scaleTargets) {
    RT_MC(mc_conv_scatter_channels(gathered.data(), _dInverse[i], target.data(), gathered.cols(), _channels->at(i), _overSample->at(i), _imgPixels->at(i),
                                   scaleTargets, 1.f, ST()));
}

// dense inputs: the layer's channels; sparse inputs: the gathered tensor's groups*filterChannels channels
#define CONV_GEOM(i) numCases(), (sparse(i) ? sparseChannels(i) : _channels->at(i)), _imgSize->at(i), _imgSize->at(i), _numFilters, _filterSize->at(i), _padding->at(i), _stride->at(i), _modulesX, _modulesX, _groups->at(i)

ConvLayer::ConvLayer(ConvNet* convNet, PyObject* paramsDict) : LocalLayer(convNet, paramsDict) {
    _partialSum = pyDictGetInt(paramsDict, "partialSum");
    _sharedBiases = pyDictGetInt(paramsDict, "sharedBiases") != 0;
}

size_t ConvLayer::splitWorkspaceBytes(int numCases) const {
    size_t need = 0;
    for (size_t i = 0; i < _channels->size(); i++) {
        const int ch = sparse(i) ? sparseChannels(i) : _channels->at(i);
        need = std::max(need, mc_conv_split_workspace_bytes(numCases, ch, _imgSize->at(i), _imgSize->at(i), _numFilters, _filterSize->at(i),
                                                            _modulesX, _modulesX, _groups->at(i)));
    }
    return need;
}

void ConvLayer::fpropActs(int i, float scaleTargets, PASS_TYPE passType) {
    // bias (+ an in-place relu consumer) ride in the contraction's epilogue when the biases are
    // shared per filter: the same values as convFilterActs + reshape/addVector (layer.cu:476-487)
    const bool first = scaleTargets == 0;
    NeuronLayer* relu = _sharedBiases ? fusableRelu() : nullptr;
    if (relu) relu->setFusedIntoProducer(true);
    const float* in = sparse(i) ? sparseInput(i, *_inputs[i]) : _inputs[i]->data();
    // (see FCLayer::fpropActs) the relu's gradient mask as bits, written by this product's epilogue
    const bool bits = relu && passType == PASS_TRAIN && first && !sparse(i) && _groups->at(i) == 1 &&
                      relu->wantsBits(_numFilters, _modules * numCases(), numCases());
    if (relu) relu->setBitsValid(bits);
    if (bits)
        RT_MC(mc_conv_filter_acts_relu_bits(in, _weights[i].getW(), getActs().data(), _biases->getW(), relu->bitsBuffer(_numFilters, _modules * numCases()),
                                            numCases(), _channels->at(i), _imgSize->at(i), _imgSize->at(i), _numFilters, _filterSize->at(i),
                                            _padding->at(i), _stride->at(i), _modulesX, _modulesX, ST()));
    else
    RT_MC(mc_conv_filter_acts_fused(in, _weights[i].getW(), getActs().data(), (first && _sharedBiases) ? _biases->getW() : nullptr,
                                    relu ? MC_NEURON_RELU : MC_NEURON_NONE, CONV_GEOM(i), scaleTargets, 1.f, ST()));
    if (first && !_sharedBiases) RT_MC(mc_add_row_vector(getActs().data(), _biases->getW(), _numFilters * _modules, numCases(), 1.f, ST()));
}

void ConvLayer::bpropBiases(DevMat& v, PASS_TYPE passType) {
    const float scale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(numCases());
    if (_sharedBiases) RT_MC(mc_sum_rows(v.data(), _biases->getGrad(), _numFilters, _modules * numCases(), 0.f, scale, ST()));
    else RT_MC(mc_sum_rows(v.data(), _biases->getGrad(), _numFilters * _modules, numCases(), 0.f, scale, ST()));
}

void ConvLayer::bpropWeights(DevMat& v, int i, PASS_TYPE passType, bool withBiases) {
    // The reference sums `partialSum` modules per chunk into a temporary and then reduces the
    // chunks (layer.cu:503-523).  Same structure; the chunking is chosen for this device rather
    // than read from the layer file (the sum is the same, only its association differs), and the
    // chunks need not be whole modules.
    const float scale = passType == PASS_GC ? 1.f : _weights[i].getEps() / _convNet->gradScaleCases(numCases());
    const float scaleTargets = _weights[i].getNumUpdates() > 0 ? 1.f : 0.f;
    const size_t wsb = mc_conv_weight_grad_workspace_bytes(numCases(), sparse(i) ? sparseChannels(i) : _channels->at(i), _numFilters, _filterSize->at(i),
                                                           _modules, _groups->at(i));
    // a sparse input was gathered by fpropActs; the gathered tensor is the layer's own and still holds this minibatch
    const float* in = sparse(i) ? _gathered[i]->data() : _prev[i]->getActs().data();
    if (withBiases) {   // per-filter biases: their gradient is the sum of v over modules and cases, which this product reads anyway
        const float bscale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(numCases());
        const size_t wsbb = mc_conv_weight_grad_bias_workspace_bytes(numCases(), sparse(i) ? sparseChannels(i) : _channels->at(i), _numFilters,
                                                                     _filterSize->at(i), _modules, _groups->at(i));
        RT_MC(mc_conv_weight_grad_bias(in, v.data(), _weights[i].getGrad(), _biases->getGrad(), CONV_GEOM(i), scaleTargets, scale, bscale,
                                       wsbb ? workspace(wsbb) : nullptr, wsbb, ST()));
        return;
    }
    RT_MC(mc_conv_weight_grad(in, v.data(), _weights[i].getGrad(), CONV_GEOM(i), scaleTargets, scale,
                              wsb ? workspace(wsb) : nullptr, wsb, ST()));
}

void ConvLayer::bpropActs(DevMat& v, int i, float scaleTargets, PASS_TYPE) {
    if (sparse(i)) {   // layer.cu:526-535: gradient of every connected copy, then the sum over the copies of a channel
        DevMat& tmp = *_gatheredGrad[i];
        tmp.resize(sparseChannels(i) * _imgPixels->at(i), numCases());
        RT_MC(mc_conv_img_acts(v.data(), _weights[i].getW(), tmp.data(), CONV_GEOM(i), 0.f, 1.f, ST()));
        sparseScatter(i, tmp, _prev[i]->getActsGrad(), scaleTargets);
        return;
    }
    RT_MC(mc_conv_img_acts(v.data(), _weights[i].getW(), _prev[i]->getActsGrad().data(), CONV_GEOM(i), scaleTargets, 1.f, ST()));
}

LocalUnsharedLayer::LocalUnsharedLayer(ConvNet* convNet, PyObject* paramsDict) : LocalLayer(convNet, paramsDict) {}

void LocalUnsharedLayer::fpropActs(int i, float scaleTargets, PASS_TYPE) {
    const float* in = sparse(i) ? sparseInput(i, *_inputs[i]) : _inputs[i]->data();
    RT_MC(mc_local_filter_acts(in, _weights[i].getW(), getActs().data(), CONV_GEOM(i), scaleTargets, 1.f, ST()));
    if (scaleTargets == 0) RT_MC(mc_add_row_vector(getActs().data(), _biases->getW(), _numFilters * _modules, numCases(), 1.f, ST()));
}

void LocalUnsharedLayer::bpropBiases(DevMat& v, PASS_TYPE passType) {
    const float scale = passType == PASS_GC ? 1.f : _biases->getEps() / _convNet->gradScaleCases(numCases());
    RT_MC(mc_sum_rows(v.data(), _biases->getGrad(), _numFilters * _modules, numCases(), 0.f, scale, ST()));
}

void LocalUnsharedLayer::bpropWeights(DevMat& v, int i, PASS_TYPE passType, bool) {
    const float scale = passType == PASS_GC ? 1.f : _weights[i].getEps() / _convNet->gradScaleCases(numCases());
    const float scaleTargets = _weights[i].getNumUpdates() > 0 ? 1.f : 0.f;
    const float* in = sparse(i) ? _gathered[i]->data() : _prev[i]->getActs().data();
    RT_MC(mc_local_weight_acts(in, v.data(), _weights[i].getGrad(), CONV_GEOM(i), scaleTargets, scale, ST()));
}

void LocalUnsharedLayer::bpropActs(DevMat& v, int i, float scaleTargets, PASS_TYPE) {
    if (sparse(i)) {
        DevMat& tmp = *_gatheredGrad[i];
        tmp.resize(sparseChannels(i) * _imgPixels->at(i), numCases());
        RT_MC(mc_local_img_acts(v.data(), _weights[i].getW(), tmp.data(), CONV_GEOM(i), 0.f, 1.f, ST()));
        sparseScatter(i, tmp, _prev[i]->getActsGrad(), scaleTargets);
        return;
    }
    RT_MC(mc_local_img_acts(v.data(), _weights[i].getW(), _prev[i]->getActsGrad().data(), CONV_GEOM(i), scaleTargets, 1.f, ST()));
}

// ==========================================================================================
// Softmax / eltsum / data                                                 (layer.cu:609-705)
// ==========================================================================================
void SoftmaxLayer::fpropActs(int, float, PASS_TYPE passType) {
    // A small classifier tail (softmax -> logreg cost, labels straight from the data) is three dependent launches of a few
    // microseconds each: one launch computes the probabilities, the cost's outputs and sums and, when training, the gradient
    // coeff * (1[label] - p) into the layer below (same values as the separate kernels, bit for bit).  MC_NO_FUSED_TAIL=1: off.
    static const bool off = getenv("MC_NO_FUSED_TAIL") != nullptr;
    _gradDone = false;
    LogregCostLayer* cost = _next.size() == 1 ? dynamic_cast<LogregCostLayer*>(_next[0]) : nullptr;
    if (cost && !off && (passType == PASS_TRAIN || passType == PASS_TEST) && (long)numCases() * _numOutputs <= 32768 && cost->getPrev().size() == 2 &&
        cost->getPrev()[1] == this) {
        DataLayer* lab = dynamic_cast<DataLayer*>(cost->getPrev()[0]);
        DevMat* labels = lab ? _convNet->currentData(lab->dataIdx()) : nullptr;
        if (labels && labels->rows() == 1 && labels->cols() == numCases()) {
            float* grad = nullptr;
            Layer* below = _prev[0];
            if (passType == PASS_TRAIN && cost->getCoeff() != 0 && below->isGradConsumer() && below->getNext().size() == 1 && !_conserveMem &&
                !below->conservesMem()) {
                DevMat& tgt = below->getActsGrad();   // written now, read when back-propagation reaches `below`; nothing else touches it in between
                tgt.resize(below->getNumOutputs(), numCases());
                grad = tgt.data();
                _gradDone = true;
            }
            cost->fusedTail(labels->data(), _inputs[0]->data(), getActs().data(), grad, numCases(), _numOutputs);
            return;
        }
    }
    RT_MC(mc_softmax(_inputs[0]->data(), getActs().data(), numCases(), _numOutputs, ST()));
}

void SoftmaxLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    if (_gradDone) {   // the fused tail wrote this gradient during fprop (scaleTargets is 0: this layer is `below`'s only consumer)
        _gradDone = false;
        return;
    }
    // directly under a logreg cost the fused gradient coeff*(1[label] - p) is used (layer.cu:626-631)
    const bool doLogregGrad = _next.size() == 1 && _next[0]->getType() == "cost.logreg";
    DevMat& tgt = _prev[0]->getActsGrad();
    if (doLogregGrad) {
        DevMat& labels = _next[0]->getPrev()[0]->getActs();
        const float coeff = dynamic_cast<CostLayer*>(_next[0])->getCoeff();
        RT_MC(mc_compute_logreg_softmax_grad(labels.data(), getActs().data(), tgt.data(), scaleTargets == 1, coeff, numCases(), _numOutputs, ST()));
    } else {
        RT_MC(mc_compute_softmax_grad(getActs().data(), v.data(), tgt.data(), scaleTargets == 1, numCases(), _numOutputs, ST()));
    }
}

EltwiseSumLayer::EltwiseSumLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _coeffs = pyDictGetFloatV(paramsDict, "coeffs");
    if (!_coeffs) throw std::runtime_error("eltsum layer without coeffs");
}

void EltwiseSumLayer::fpropActs(int i, float scaleTargets, PASS_TYPE) {
    if (scaleTargets == 0) RT_MC(mc_scale(getActs().data(), _inputs[i]->data(), _coeffs->at(i), getActs().numel(), ST()));
    else RT_MC(mc_axpby(getActs().data(), _inputs[i]->data(), 1.f, _coeffs->at(i), getActs().numel(), ST()));
}

void EltwiseSumLayer::bpropActs(DevMat& v, int i, float scaleTargets, PASS_TYPE) {
    DevMat& tgt = _prev[i]->getActsGrad();
    if (scaleTargets == 0) RT_MC(mc_scale(tgt.data(), v.data(), _coeffs->at(i), v.numel(), ST()));
    else RT_MC(mc_axpby(tgt.data(), v.data(), scaleTargets, _coeffs->at(i), v.numel(), ST()));
}

// ==========================================================================================
// Element-wise max                                                        (layer.cu:667-680)
// ==========================================================================================
void EltwiseMaxLayer::fpropActs(int i, float, PASS_TYPE) {
    if (i == 1) RT_MC(mc_eltwise_max(_inputs[1]->data(), _inputs[0]->data(), getActs().data(), getActs().numel(), ST()));
    else if (i > 1) RT_MC(mc_eltwise_max(getActs().data(), _inputs[i]->data(), getActs().data(), getActs().numel(), ST()));
}

void EltwiseMaxLayer::bpropActs(DevMat& v, int i, float scaleTargets, PASS_TYPE) {
    RT_MC(mc_compute_eltwise_max_grad(v.data(), _inputs[i]->data(), getActs().data(), _prev[i]->getActsGrad().data(), scaleTargets != 0, v.numel(), ST()));
}

// ==========================================================================================
// Image layers: nailbed, blur, resize, rgb2yuv, rgb2lab                   (layer.cu:770-867)
// ==========================================================================================
NailbedLayer::NailbedLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _channels = pyDictGetInt(paramsDict, "channels");
    _start = pyDictGetInt(paramsDict, "start");
    _stride = pyDictGetInt(paramsDict, "stride");
    _outputsX = pyDictGetInt(paramsDict, "outputsX");
    _imgSize = pyDictGetInt(paramsDict, "imgSize");
}

void NailbedLayer::fpropActs(int, float, PASS_TYPE) {
    RT_MC(mc_conv_bed_of_nails(_inputs[0]->data(), getActs().data(), numCases(), _channels, _imgSize, _start, _stride, 0.f, 1.f, ST()));
}

void NailbedLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    RT_MC(mc_conv_bed_of_nails_undo(v.data(), _prev[0]->getActsGrad().data(), numCases(), _channels, _imgSize, _start, _stride, scaleTargets, 1.f, ST()));
}

GaussianBlurLayer::GaussianBlurLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _channels = pyDictGetInt(paramsDict, "channels");
    _imgSize = pyDictGetInt(paramsDict, "imgSize");
    HostMat* h = pyDictGetMatrix(paramsDict, "filter");   // (1, filterSize)
    if (h->rows() != 1 || h->cols() % 2 != 1) throw std::runtime_error(fmt("blur layer '%s': filter must be (1, odd), got %dx%d", _name.c_str(), h->rows(), h->cols()));
    _taps.resize(h->numel());
    h->toRowMajor(_taps.data());
    _radius = h->cols() / 2;
    h->release();
    delete h;
}

void GaussianBlurLayer::copyToGPU() { _filter.copyFromHost(_taps.data(), 1, (int)_taps.size()); }

void GaussianBlurLayer::fpropActs(int, float, PASS_TYPE) {
    // rows into the output, then columns in place (layer.cu:796-799)
    RT_MC(mc_conv_gaussian_blur(_inputs[0]->data(), _filter.data(), getActs().data(), numCases(), _channels, _imgSize, _radius, 1, 0.f, 1.f, ST()));
    RT_MC(mc_conv_gaussian_blur(getActs().data(), _filter.data(), getActs().data(), numCases(), _channels, _imgSize, _radius, 0, 0.f, 1.f, ST()));
}

// The filter is symmetric, so the blur is its own transpose (layer.cu:801-807).
void GaussianBlurLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    DevMat& tgt = _prev[0]->getActsGrad();
    DevMat& tgt1 = scaleTargets != 0 ? _actGradsTmp : tgt;
    if (&tgt1 == &_actGradsTmp) _actGradsTmp.resize(v);
    RT_MC(mc_conv_gaussian_blur(v.data(), _filter.data(), tgt1.data(), numCases(), _channels, _imgSize, _radius, 1, 0.f, 1.f, ST()));
    RT_MC(mc_conv_gaussian_blur(tgt1.data(), _filter.data(), tgt.data(), numCases(), _channels, _imgSize, _radius, 0, scaleTargets, 1.f, ST()));
}

ResizeLayer::ResizeLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _channels = pyDictGetInt(paramsDict, "channels");
    _imgSize = pyDictGetInt(paramsDict, "imgSize");
    _tgtSize = pyDictGetInt(paramsDict, "tgtSize");
    _scale = pyDictGetFloat(paramsDict, "scale");
}

void ResizeLayer::fpropActs(int, float, PASS_TYPE) {
    RT_MC(mc_conv_resize_bilinear(_inputs[0]->data(), getActs().data(), numCases(), _channels, _imgSize, _tgtSize, _scale, ST()));
}

void ResizeLayer::bpropActs(DevMat&, int, float, PASS_TYPE) {   // layer.cu:829-832: "Can't do this"
    throw std::runtime_error(fmt("resize layer '%s' has no backward pass", _name.c_str()));
}

RGBToYUVLayer::RGBToYUVLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _imgPixels = _numOutputs / 3;
}

void RGBToYUVLayer::fpropActs(int, float, PASS_TYPE) {
    RT_MC(mc_conv_rgb_to_yuv(_inputs[0]->data(), getActs().data(), numCases(), _imgPixels, ST()));
}

void RGBToYUVLayer::bpropActs(DevMat&, int, float, PASS_TYPE) {   // layer.cu:846-849, 864-867
    throw std::runtime_error(fmt("colour-transform layer '%s' has no backward pass", _name.c_str()));
}

RGBToLABLayer::RGBToLABLayer(ConvNet* convNet, PyObject* paramsDict) : RGBToYUVLayer(convNet, paramsDict) {
    _center = pyDictGetInt(paramsDict, "center") != 0;
}

void RGBToLABLayer::fpropActs(int, float, PASS_TYPE) {
    RT_MC(mc_conv_rgb_to_lab(_inputs[0]->data(), getActs().data(), numCases(), _imgPixels, _center, ST()));
}

DataLayer::DataLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _dataIdx = pyDictGetInt(paramsDict, "dataIdx");
}

void DataLayer::fprop(PASS_TYPE) { throw std::runtime_error("No data given!"); }

void DataLayer::fprop(DevMatV& data, PASS_TYPE passType) {
    if (_dataIdx < 0 || _dataIdx >= (int)data.size()) throw std::runtime_error(fmt("data layer '%s': dataIdx %d but %zu matrices given", _name.c_str(), _dataIdx, data.size()));
    _outputs = data[_dataIdx];
    if (_numOutputs > 0 && _outputs->rows() != _numOutputs)
        throw std::runtime_error(fmt("data layer '%s': matrix has %d rows, layer expects %d", _name.c_str(), _outputs->rows(), _numOutputs));
    _numCases = _outputs->cols();
    fpropNext(passType);
}

// ==========================================================================================
// Pooling                                                                  (layer.cu:712-763)
// ==========================================================================================
PoolLayer::PoolLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _channels = pyDictGetInt(paramsDict, "channels");
    _sizeX = pyDictGetInt(paramsDict, "sizeX");
    _start = pyDictGetInt(paramsDict, "start");
    _stride = pyDictGetInt(paramsDict, "stride");
    _outputsX = pyDictGetInt(paramsDict, "outputsX");
    _imgSize = pyDictGetInt(paramsDict, "imgSize");
    _pool = pyDictGetString(paramsDict, "pool");
}

PoolLayer& PoolLayer::makePoolLayer(ConvNet* convNet, PyObject* paramsDict) {
    const std::string pool = pyDictGetString(paramsDict, "pool");
    if (pool == "max") return *new MaxPoolLayer(convNet, paramsDict);
    if (pool == "avg") return *new AvgPoolLayer(convNet, paramsDict);
    throw std::runtime_error("Unknown pooling layer type " + pool);
}

void AvgPoolLayer::fpropActs(int, float, PASS_TYPE) {
    RT_MC(mc_conv_local_pool(_inputs[0]->data(), getActs().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, MC_POOL_AVG, ST()));
}

void AvgPoolLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    const float* mask = scaleTargets == 0.f ? claimReluGradMask(0) : nullptr;   // == _prev[0]->getActs()
    if (mask)
        RT_MC(mc_conv_local_avg_undo_relu_grad(v.data(), mask, _prev[0]->getActsGrad().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride,
                                               _outputsX, 1.f, ST()));
    else
        RT_MC(mc_conv_local_avg_undo(v.data(), _prev[0]->getActsGrad().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, scaleTargets, 1.f, ST()));
}

void MaxPoolLayer::fpropActs(int, float, PASS_TYPE passType) {
    // NIN's pool -> dropout pairs: the dropout neuron's forward step in this layer's store, one pass over the pooled tensor
    // instead of two (and the keep bytes for the backward pass)
    _dropNext = nullptr;
    NeuronLayer* nl = _next.size() == 1 ? dynamic_cast<NeuronLayer*>(_next[0]) : nullptr;
    float prob = 0;
    unsigned long long seed = 0, offset = 0;
    if (nl && nl->fuseDropoutIntoPool(getActs().numel(), passType, &prob, &seed, &offset)) {
        nl->getActs().resize(getActs());
        const size_t words = (mc_dropout_keep_bytes(getActs().numel()) + 3) / 4;
        _keep.resize(1, (int)words);
        RT_MC(mc_conv_local_pool_dropout(_inputs[0]->data(), getActs().data(), nl->getActs().data(), reinterpret_cast<unsigned char*>(_keep.data()),
                                         numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, MC_POOL_MAX, prob, seed, offset, ST()));
        _dropNext = nl;
        return;
    }
    // layers-80sec / 18pct: `neuron=relu` on the pool layer, an in-place relu above: applied in this layer's store
    _reluNext = nullptr;
    if (nl && nl->fuseReluIntoPool(passType)) {
        if (!nl->inPlace()) nl->getActs().resize(getActs());
        RT_MC(mc_conv_local_pool_relu(_inputs[0]->data(), getActs().data(), nl->getActs().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, MC_POOL_MAX, ST()));
        _reluNext = nl;
        return;
    }
    RT_MC(mc_conv_local_pool(_inputs[0]->data(), getActs().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, MC_POOL_MAX, ST()));
}

void MaxPoolLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE passType) {
    const float* mask = scaleTargets == 0.f ? claimReluGradMask(0) : nullptr;   // == _prev[0]->getActs()
    if (_reluNext && _reluNext->poolGatesRelu()) {   // v has not passed the relu's backward step yet: gated here
        RT_MC(mc_conv_local_max_undo_relu_out(mask ? mask : _prev[0]->getActs().data(), v.data(), getActs().data(), _prev[0]->getActsGrad().data(),
                                              numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX, mask ? 0.f : scaleTargets, 1.f,
                                              mask ? 1 : 0, ST()));
        return;
    }
    if (_dropNext && _dropNext->poolUndoesDropout() && passType == PASS_TRAIN) {   // v has not passed the dropout's backward step yet: masked here
        RT_MC(mc_conv_local_max_undo_dropout(mask ? mask : _prev[0]->getActs().data(), v.data(), getActs().data(), reinterpret_cast<const unsigned char*>(_keep.data()),
                                             _prev[0]->getActsGrad().data(), numCases(), _channels, _imgSize, _sizeX, _start, _stride, _outputsX,
                                             mask ? 0.f : scaleTargets, 1.f, mask ? 1 : 0, ST()));
        return;
    }
    if (mask)
        RT_MC(mc_conv_local_max_undo_relu_grad(mask, v.data(), getActs().data(), _prev[0]->getActsGrad().data(), numCases(), _channels, _imgSize,
                                               _sizeX, _start, _stride, _outputsX, 0.f, 1.f, ST()));
    else
        RT_MC(mc_conv_local_max_undo(_prev[0]->getActs().data(), v.data(), getActs().data(), _prev[0]->getActsGrad().data(), numCases(), _channels,
                                     _imgSize, _sizeX, _start, _stride, _outputsX, scaleTargets, 1.f, ST()));
}

// ==========================================================================================
// Normalisation                                                            (layer.cu:874-941)
// ==========================================================================================
ResponseNormLayer::ResponseNormLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _channels = pyDictGetInt(paramsDict, "channels");
    _size = pyDictGetInt(paramsDict, "size");
    _scale = pyDictGetFloat(paramsDict, "scale");
    _pow = pyDictGetFloat(paramsDict, "pow");
    _imgSize = pyDictGetInt(paramsDict, "imgSize");
}

void ResponseNormLayer::fpropActs(int, float, PASS_TYPE) {
    _denoms.resize(getActs());
    RT_MC(mc_conv_response_norm(_inputs[0]->data(), _denoms.data(), getActs().data(), numCases(), _channels, _imgSize, _size, _scale, _pow, ST()));
}

void ResponseNormLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    RT_MC(mc_conv_response_norm_undo(v.data(), _denoms.data(), _prev[0]->getActs().data(), getActs().data(), _prev[0]->getActsGrad().data(), numCases(),
                                     _channels, _imgSize, _size, _scale, _pow, scaleTargets, 1.f, ST()));
}

void ResponseNormLayer::truncBwdActs() {
    Layer::truncBwdActs();
    if (_conserveMem) _denoms.free();
}

CrossMapResponseNormLayer::CrossMapResponseNormLayer(ConvNet* convNet, PyObject* paramsDict) : ResponseNormLayer(convNet, paramsDict) {
    _blocked = pyDictGetInt(paramsDict, "blocked") != 0;
    _k = pyDictGetFloat(paramsDict, "k");
}

void CrossMapResponseNormLayer::fpropActs(int, float, PASS_TYPE) {
    _denoms.resize(getActs());
    RT_MC(mc_conv_response_norm_cross_map(_inputs[0]->data(), _denoms.data(), getActs().data(), numCases(), _channels, _imgSize, _size, _scale, _pow,
                                          _blocked, _k, ST()));
}

void CrossMapResponseNormLayer::bpropActs(DevMat& v, int, float scaleTargets, PASS_TYPE) {
    RT_MC(mc_conv_response_norm_cross_map_undo(v.data(), _denoms.data(), _prev[0]->getActs().data(), getActs().data(), _prev[0]->getActsGrad().data(),
                                               numCases(), _channels, _imgSize, _size, _scale, _pow, _blocked, scaleTargets, 1.f, ST()));
}

void ContrastNormLayer::fpropActs(int, float, PASS_TYPE) {
    DevMat& images = *_inputs[0];
    _meanDiffs.resize(images);
    _denoms.resize(getActs());
    // meanDiffs = images - local mean (layer.cu:925-930)
    RT_MC(mc_conv_local_pool(images.data(), _meanDiffs.data(), numCases(), _channels, _imgSize, _size, -_size / 2, 1, _imgSize, MC_POOL_AVG, ST()));
    RT_MC(mc_axpby(_meanDiffs.data(), images.data(), -1.f, 1.f, images.numel(), ST()));
    RT_MC(mc_conv_contrast_norm(images.data(), _meanDiffs.data(), _denoms.data(), getActs().data(), numCases(), _channels, _imgSize, _size, _scale, _pow, ST()));
}

void ContrastNormLayer::bpropActs(DevMat& v, int i, float scaleTargets, PASS_TYPE) {
    RT_MC(mc_conv_response_norm_undo(v.data(), _denoms.data(), _meanDiffs.data(), getActs().data(), _prev[i]->getActsGrad().data(), numCases(), _channels,
                                     _imgSize, _size, _scale, _pow, scaleTargets, 1.f, ST()));
}

void ContrastNormLayer::truncBwdActs() {
    ResponseNormLayer::truncBwdActs();
    if (_conserveMem) _meanDiffs.free();
}

// ==========================================================================================
// Costs                                                                   (layer.cu:948-1036)
// ==========================================================================================
CostLayer::CostLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {
    _coeff = pyDictGetFloat(paramsDict, "coeff");
}

CostLayer::~CostLayer() {
    if (_slots) (void)hipFree(_slots);
}

double* CostLayer::costSlot() {
    if (!_slots) {
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_slots), sizeof(double) * (MAX_SLOTS + 2) * numCostValues()));
        // on the compute stream: a null-stream memset is not ordered against a non-blocking stream
        RT_HIP(hipMemsetAsync(_slots, 0, sizeof(double) * (MAX_SLOTS + 2) * numCostValues(), computeStream()));
    }
    const int s = _convNet->costSlotIndex();
    if (s >= MAX_SLOTS && !_convNet->capturing()) throw std::runtime_error(fmt("more than %d minibatches in one batch", MAX_SLOTS));
    return _slots + (size_t)s * numCostValues();
}

void CostLayer::bprop(PASS_TYPE passType) {
    if (_coeff != 0) Layer::bprop(passType);
}

CostLayer& CostLayer::makeCostLayer(ConvNet* convNet, const std::string& type, PyObject* paramsDict) {
    if (type == "cost.logreg") return *new LogregCostLayer(convNet, paramsDict);
    if (type == "cost.sum2") return *new SumOfSquaresCostLayer(convNet, paramsDict);
    auto it = pluginLayers().find(type);
    if (it != pluginLayers().end()) {
        CostLayer* c = dynamic_cast<CostLayer*>(it->second(convNet, paramsDict));
        if (!c) throw std::runtime_error("plugin layer '" + type + "' is not a cost layer");
        return *c;
    }
    throw std::runtime_error("Unknown cost layer type " + type);
}

void LogregCostLayer::fusedTail(const float* labels, const float* in, float* probs, float* grad, int N, int numOut) {
    _ownOutputs.resize(1, N);
    _correctProbs.resize(1, N);
    RT_MC(mc_softmax_logreg(labels, in, probs, _ownOutputs.data(), _correctProbs.data(), costSlot(), grad, 0, _coeff, N, numOut, ST()));
    _fusedThisPass = true;
}

void LogregCostLayer::fpropActs(int inpIdx, float, PASS_TYPE) {
    if (inpIdx != 0) return;   // both inputs are used together
    if (_fusedThisPass) {      // the softmax layer below has already produced everything (fusedTail)
        _fusedThisPass = false;
        return;
    }
    DevMat& labels = *_inputs[0];
    DevMat& probs = *_inputs[1];
    const int N = labels.cols();
    _ownOutputs.resize(1, N);       // log p(label) per case
    _correctProbs.resize(1, N);
    // per-case values and both of their sums (this minibatch's cost slot) in one launch
    RT_MC(mc_compute_logreg_cost_sums(labels.data(), probs.data(), _ownOutputs.data(), _correctProbs.data(), costSlot(), N, probs.rows(), ST()));
}

void LogregCostLayer::finishCost(double* vals, int numCases) const {
    vals[0] = -vals[0];                // -sum log p(label)
    vals[1] = numCases - vals[1];      // number of errors (ties count fractionally)
}

void LogregCostLayer::bpropActs(DevMat&, int inpIdx, float scaleTargets, PASS_TYPE) {
    if (inpIdx != 1) throw std::runtime_error("logreg cost: gradient requested for the label input");
    // a softmax directly below computes the whole gradient itself (layer.cu:1012-1017)
    const bool doWork = _prev[1]->getNext().size() > 1 || _prev[1]->getType() != "softmax";
    if (!doWork) return;
    DevMat& probs = _prev[1]->getActs();
    RT_MC(mc_compute_logreg_grad(_prev[0]->getActs().data(), probs.data(), _prev[1]->getActsGrad().data(), scaleTargets == 1, _coeff, probs.cols(),
                                 probs.rows(), ST()));
}

void SumOfSquaresCostLayer::fpropActs(int, float, PASS_TYPE) {
    DevMat& x = *_inputs[0];
    _ownOutputs.resize(x);
    RT_MC(mc_neuron_activate(MC_N_SQUARE, 0, 0, x.data(), _ownOutputs.data(), x.numel(), ST()));
    RT_MC(mc_sum_all(_ownOutputs.data(), x.numel(), costSlot(), ST()));
}

void SumOfSquaresCostLayer::bpropActs(DevMat&, int i, float scaleTargets, PASS_TYPE) {
    DevMat& tgt = _prev[i]->getActsGrad();
    if (scaleTargets == 0) RT_MC(mc_scale(tgt.data(), _inputs[0]->data(), -_coeff, tgt.numel(), ST()));
    else RT_MC(mc_axpby(tgt.data(), _inputs[0]->data(), scaleTargets, -_coeff, tgt.numel(), ST()));
}

}  // namespace rt
