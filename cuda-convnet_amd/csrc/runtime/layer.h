// layer.h -- the layer graph of the host runtime.  Semantics follow the reference's
// Kernel/include/layers/layer.cuh:49-383 + Kernel/src/layers/layer.cu (cited per method);
// this is also the header a Plugins/*.so author compiles against (Layer, WeightLayer, CostLayer,
// Neuron are the base classes a plugin subclasses; see csrc/plugins/extralayers.cpp).
//
// Differences from the reference that are deliberate:
//   * activations are DevMat (features, numCases) row-major, no soft transposes;
//   * every weight tensor keeps an explicit gradient buffer (the reference's useGrad=true
//     formulation, weights.cuh:150-164, used for ALL layer types) because data parallelism
//     all-reduces raw gradients; FC / local / cccp in the reference fold the momentum into the
//     GEMM's beta instead (layer.cu:402-406), which is the same arithmetic on one GPU;
//   * weights, increments and gradients of the whole net live in three contiguous arenas
//     (ConvNet::copyToGPU) so gradient all-reduce buckets are address ranges;
//   * cost values stay on the device until the end of the batch (no per-minibatch sync).
#pragma once
#include <Python.h>

#include <map>
#include <string>
#include <vector>

#include "devmat.h"
#include "pyutil.h"

namespace rt {

class ConvNet;
class Layer;

enum PASS_TYPE { PASS_TRAIN, PASS_TEST, PASS_GC };

// ------------------------------------------------------------------------------------------
// Neuron (Kernel/include/layers/neuron.cuh:65-462): activation functor pair applied by
// NeuronLayer.  `inputs` and `outputs` may be the same matrix.
// ------------------------------------------------------------------------------------------
class Neuron {
public:
    virtual ~Neuron() {}
    virtual void activate(DevMat& inputs, DevMat& outputs, PASS_TYPE passType);
    // target = (add ? target : 0) + f'(.) * actsGrad ; target may alias actsGrad
    virtual void computeInputGrad(DevMat& actsGrad, DevMat& target, bool add, PASS_TYPE passType);
    // The plain relu can be folded into the producing contraction's epilogue.
    virtual bool isRelu() const { return false; }
    // A dropout-style neuron (zero where a counter-based uniform < prob, PluginsSrc/include/extralayers.cuh:39-77) that a pool layer
    // can apply in its own store (mc_conv_local_pool_dropout) and undo in its own backward pass.  beginFusedDropout is called
    // INSTEAD of activate() in a training pass and reserves the uniforms exactly as activate() would; false = not now (injected
    // masks, test passes): call activate().  fusedDropoutActive: this pass's forward step went that way.
    virtual bool beginFusedDropout(size_t numel, PASS_TYPE passType, float* prob, unsigned long long* seed, unsigned long long* offset) { return false; }
    virtual bool fusedDropoutActive() const { return false; }
    static Neuron& makeNeuron(PyObject* neuronDict);

protected:
    DevMat *_inputs = nullptr, *_outputs = nullptr;
    bool _activated = false;
    // built-in element-wise neurons are one (type, a, b) triple for the kernel library
    int _mcType = 0;  // MC_N_IDENT
    float _a = 0, _b = 0;
    bool _usesInputs = false;  // gradient needs the neuron's input rather than its output
    friend class BuiltinNeuron;
};

// ------------------------------------------------------------------------------------------
// Weights (Kernel/include/layers/weights.cuh:41-219)
// ------------------------------------------------------------------------------------------
class Weights {
public:
    // `deviceFortran`: the device copy holds the matrix column-major, i.e. as a row-major
    // (cols, rows) matrix -- what FC / cccp want, and what their NumPy arrays already are.
    Weights(HostMat& hW, HostMat& hInc, float eps, float wc, float mom, bool deviceFortran);
    Weights(Weights& src, float eps);  // shared with another layer's matrix
    ~Weights();

    float* getW() const { return owner()._w; }
    float* getInc() const { return owner()._inc; }
    float* getGrad() const { return owner()._grad; }
    int getNumRows() const { return _hW->rows(); }   // logical (Python-visible) shape
    int getNumCols() const { return _hW->cols(); }
    size_t numel() const { return _hW->numel(); }
    bool deviceFortran() const { return owner()._devFortran; }
    bool isOwner() const { return _src == nullptr; }
    float getEps() const { return _eps; }
    float getMom() const { return _mom; }
    float getWC() const { return _wc; }
    int getNumUpdates() const { return owner()._numUpdates; }
    int incNumUpdates() { return const_cast<Weights&>(owner())._numUpdates++; }
    size_t arenaOffset() const { return owner()._offset; }
    // data parallel bookkeeping (buckets.h): position of the owning tensor in the arena order, and how many layers' gradient
    // products add into it per step (the owner's and those of the layers sharing it, each only if it learns)
    int arenaIndex() const { return owner()._arenaIndex; }
    void setArenaIndex(int i) { _arenaIndex = i; }
    int contributors() const { return owner()._contributors; }

    void bind(float* w, float* inc, float* grad, size_t offset);  // arena slices (owner only)
    void copyToGPU();    // owner only; no-op for shared
    void copyToCPU();
    void update();       // weights.cuh:150-164, one fused kernel
    void markUpdated() { _numUpdates = 0; }   // the net-wide single-launch update did this tensor
    // grad-check support: element (i, j) of the logical matrix <-> device offset
    size_t deviceIndex(int i, int j) const { return deviceFortran() ? (size_t)j * getNumRows() + i : (size_t)i * getNumCols() + j; }
    void releaseHost();  // GIL required

private:
    const Weights& owner() const { return _src ? _src->owner() : *this; }
    HostMat *_hW, *_hInc;
    Weights* _src = nullptr;
    float *_w = nullptr, *_inc = nullptr, *_grad = nullptr;
    size_t _offset = 0;
    float _eps, _wc, _mom;
    bool _devFortran = false;
    int _numUpdates = 0;
    int _arenaIndex = -1, _contributors = 0;
};

class WeightList {
public:
    Weights& operator[](int i) const { return *_list[i]; }
    ~WeightList() { for (Weights* w : _list) delete w; }
    void addWeights(Weights& w) { _list.push_back(&w); }
    int getSize() const { return (int)_list.size(); }
    void update() { for (Weights* w : _list) w->update(); }
    void copyToCPU() { for (Weights* w : _list) w->copyToCPU(); }
    void copyToGPU() { for (Weights* w : _list) w->copyToGPU(); }

private:
    std::vector<Weights*> _list;
};

// ------------------------------------------------------------------------------------------
// Layer (layer.cuh:49-108, layer.cu:45-244)
// ------------------------------------------------------------------------------------------
class Layer {
public:
    Layer(ConvNet* convNet, PyObject* paramsDict);
    virtual ~Layer() {}

    virtual void fprop(PASS_TYPE passType);                 // arrival-counted (layer.cu:76-85)
    void fprop(DevMat& v, PASS_TYPE passType);
    virtual void fprop(DevMatV& v, PASS_TYPE passType);     // layer.cu:93-115
    virtual void bprop(PASS_TYPE passType);                 // layer.cu:117-123
    void bprop(DevMat& v, PASS_TYPE passType);              // layer.cu:125-167
    virtual void reset();
    // a pass was abandoned half way (an exception during a step capture): forget everything that only holds within one pass
    virtual void abortPass() { reset(); }
    int getRcvdFInputs() const { return _rcvdFInputs; }
    int getRcvdBInputs() const { return _rcvdBInputs; }
    int incRcvdBInputs() { return ++_rcvdBInputs; }
    const std::string& getName() const { return _name; }
    const std::string& getType() const { return _type; }
    void addNext(Layer* l);
    void addPrev(Layer* l) { _prev.push_back(l); }
    std::vector<Layer*>& getPrev() { return _prev; }
    std::vector<Layer*>& getNext() { return _next; }
    virtual DevMat& getActs();
    virtual DevMat& getActsGrad();
    int getNumOutputs() const { return _numOutputs; }
    bool conservesMem() const { return _conserveMem; }
    virtual void postInit();
    bool isGradConsumer();
    virtual bool isGradProducer() { return true; }
    // true for layers whose backward-data product can read the relu gradient mask of its input as bits (dense products), when
    // the producer lays the bits out for a (rows, ld) matrix: the layout is tied to that view (32-row blocks), and a conv
    // output (filters, modules * cases) read by an fc layer as (features, cases) is a different one.
    virtual bool readsReluBits(int rows, int ld, int numCases) const { return false; }

    // bytes of library-side split-K workspace this layer's forward / backward-data calls need at `numCases` (0: none); the
    // net reserves the largest at initModel so that no training step allocates (mc_reserve_conv_split_workspace)
    virtual size_t splitWorkspaceBytes(int numCases) const { return 0; }
    // true when bpropActs uses this layer's OUTPUT matrix as scratch (the normalisation layers, as in the reference): whoever
    // still reads those outputs on another stream -- the weight gradients of the layers above, when the step forks -- has to
    // be waited for first (Layer::bprop)
    virtual bool bpropOverwritesActs() const { return false; }

    // weight layers override these
    virtual void updateWeights() {}
    virtual void checkGradients() {}
    virtual void copyToCPU() {}
    virtual void copyToGPU() {}
    virtual void releaseHost() {}   // drop Python references (GIL held)

protected:
    virtual void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) = 0;
    virtual void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) = 0;
    virtual void bpropCommon(DevMat& v, PASS_TYPE passType) {}
    virtual void truncBwdActs();
    void fpropNext(PASS_TYPE passType);
    int numCases() const { return _numCases; }
    // Backward fusion: when input i is a relu neuron layer that gets its gradient from this layer alone and
    // passes it on in place, returns that relu's output (the mask for mc_*_relu_grad) and tells the neuron
    // layer to skip its own pass over the gradient; otherwise nullptr.
    const float* claimReluGradMask(int inpIdx);
    // The same claim when the relu's producer wrote the mask as bits this pass (mc_kernels.h "relu bit masks"); nullptr otherwise
    // (nothing is claimed then: ask claimReluGradMask next).
    const unsigned* claimReluGradBits(int inpIdx);

    ConvNet* _convNet;
    std::vector<Layer*> _prev, _next;
    int _rcvdFInputs = 0, _rcvdBInputs = 0;
    DevMatV _inputs;
    DevMat *_outputs, *_actsGrad;
    DevMat _ownOutputs, _ownActsGrad;
    bool _gradConsumer, _foundGradConsumers = false, _conserveMem;
    int _numGradProducersNext = 0;
    int _actsTarget, _actsGradTarget;
    int _numOutputs, _numCases = 0;
    std::string _name, _type;
};

class NeuronLayer : public Layer {
public:
    NeuronLayer(ConvNet* convNet, PyObject* paramsDict);
    Neuron& getNeuron() { return *_neuron; }
    // set by the producing layer when it already applied this (in-place relu) neuron in its epilogue
    void setFusedIntoProducer(bool f) { _fused = f; }
    bool inPlace() const { return _actsTarget == 0; }
    // true when the consumer of this layer's output can apply the relu gradient mask itself
    bool canFuseGradMask();
    void setGradMasked() { _gradMasked = true; }
    // Relu masks as bits: the producing layer asks wantsBits() (the one consumer is a dense product that can read them and
    // this relu passes its gradient on in place), writes them into bitsBuffer() from its epilogue and says so every pass.
    bool wantsBits(int rows, int ld, int numCases);
    unsigned* bitsBuffer(int rows, int ld);
    void setBitsValid(bool v) { _bitsValid = v; }
    const unsigned* validBits() const { return _bitsValid ? reinterpret_cast<const unsigned*>(_bits.data()) : nullptr; }
    void abortPass() override { Layer::abortPass(); _gradMasked = false; _bitsValid = false; _dropFwd = _dropBwd = false; _reluFwd = _reluBwd = false; }
    // Dropout fused into the pool layer below (MaxPoolLayer): the pool asks in its forward step; a yes means it writes this
    // layer's output itself (this layer's own forward step is then skipped) and, when the gradient side allows it -- this layer
    // hands its gradient down in place and nothing accumulates -- also masks the gradient in its max-undo (backward step skipped).
    bool fuseDropoutIntoPool(size_t numel, PASS_TYPE passType, float* prob, unsigned long long* seed, unsigned long long* offset);
    bool poolUndoesDropout() const { return _dropBwd; }
    // An in-place relu on a max-pool layer (`neuron=relu` in the pool's definition): the pool applies it in its own store
    // (setFusedIntoProducer) and, when this layer hands its gradient down in place and nothing accumulates into it, also takes over
    // the relu's backward step in its max-undo (the gradient gated by the relu'd pooled value it routes by anyway).
    bool fuseReluIntoPool(PASS_TYPE passType);
    bool poolGatesRelu() const { return _reluBwd; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    Neuron* _neuron;
    bool _dropFwd = false, _dropBwd = false;
    bool _reluFwd = false, _reluBwd = false;   // this pass: the pool below wrote this relu's output / applies its backward step
    bool _fused = false;
    bool _gradMasked = false;   // this pass's gradient already carries the relu mask (see claimReluGradMask)
    bool _bitsValid = false;
    DevMat _bits;               // 32-bit words held in a float matrix: ((rows + 31) / 32, ld)
};

class WeightLayer : public Layer {
public:
    // deviceFortran: see Weights.  (The reference's `trans` / `useGrad` arguments are gone:
    // layout is explicit and every tensor has a gradient buffer.)
    WeightLayer(ConvNet* convNet, PyObject* paramsDict, bool deviceFortran);
    void updateWeights() override;
    void copyToCPU() override;
    void copyToGPU() override;
    void checkGradients() override;
    void releaseHost() override;
    Weights& getWeights(int idx) { return _weights[idx]; }
    Weights& getBiases() { return *_biases; }
    int getNumWeights() const { return _weights.getSize(); }

protected:
    void bpropCommon(DevMat& v, PASS_TYPE passType) override;   // layer.cu:318-334
    virtual void bpropBiases(DevMat& v, PASS_TYPE passType) = 0;
    // withBiases: also deliver the bias gradient, from the same pass over `v` (asked only of layers whose
    // fusesBiasGrad() is true, for input 0, and instead of bpropBiases)
    virtual void bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases = false) = 0;
    virtual bool fusesBiasGrad() const { return false; }
    // If the only consumer is an in-place plain relu, returns it (epilogue fusion candidate).
    NeuronLayer* fusableRelu();
    // scratch for split-K / partial sums, shared by all layers of the net
    float* workspace(size_t bytes);

    WeightList _weights;
    Weights* _biases;
    float _wStep = 0.001f, _bStep = 0.002f;
};

class FCLayer : public WeightLayer {
public:
    FCLayer(ConvNet* convNet, PyObject* paramsDict);
    bool readsReluBits(int rows, int ld, int numCases) const override { return ld == numCases; }   // its view: (features, cases)
    bool fusesBiasGrad() const override { return true; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropBiases(DevMat& v, PASS_TYPE passType) override;
    void bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases = false) override;
};

class LocalLayer : public WeightLayer {
public:
    LocalLayer(ConvNet* convNet, PyObject* paramsDict);
    ~LocalLayer();

    void copyToGPU() override;

protected:
    // Random sparse connectivity (layer.cu:437-457; filterConns of Python/layer.py:848-871): input i's filter group g
    // sees channels filterConns[i][g*filterChannels + c].  sparseInput(i) gathers those channels into a dense
    // (groups*filterChannels)-channel tensor the grouped contractions run on; sparseChannels(i) is its channel count.
    bool sparse(int i) const { return _randSparse->at(i) != 0; }
    int sparseChannels(int i) const { return _groups->at(i) * _filterChannels->at(i); }
    const float* sparseInput(int i, const DevMat& input);
    // backward data of a sparse input: `gathered` holds the gradient of the gathered tensor
    void sparseScatter(int i, const DevMat& gathered, DevMat& target, float scaleTargets);

    intv *_padding, *_stride, *_filterSize, *_channels, *_imgSize, *_groups;
    intv *_imgPixels, *_filterPixels, *_filterChannels, *_overSample, *_randSparse;
    int _modulesX, _modules, _numFilters;
    std::vector<std::vector<int>> _conns, _inverse;   // per input; empty when dense
    std::vector<int*> _dConns, _dInverse;             // device copies
    std::vector<DevMat*> _gathered, _gatheredGrad;
};

class ConvLayer : public LocalLayer {
public:
    ConvLayer(ConvNet* convNet, PyObject* paramsDict);
    size_t splitWorkspaceBytes(int numCases) const override;
    bool fusesBiasGrad() const override { return _sharedBiases; }   // per-filter biases: the sum of v over modules and cases

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropBiases(DevMat& v, PASS_TYPE passType) override;
    void bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases = false) override;
    int _partialSum;
    bool _sharedBiases;
};

class LocalUnsharedLayer : public LocalLayer {
public:
    LocalUnsharedLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropBiases(DevMat& v, PASS_TYPE passType) override;
    void bpropWeights(DevMat& v, int inpIdx, PASS_TYPE passType, bool withBiases = false) override;
};

class SoftmaxLayer : public Layer {
public:
    SoftmaxLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {}
    void abortPass() override { Layer::abortPass(); _gradDone = false; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    bool _gradDone = false;   // this pass's gradient was written by the fused classifier tail (mc_softmax_logreg) during fprop
};

class EltwiseSumLayer : public Layer {
public:
    EltwiseSumLayer(ConvNet* convNet, PyObject* paramsDict);
    ~EltwiseSumLayer() { delete _coeffs; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    floatv* _coeffs;
};

// layer.cuh:208-215, layer.cu:667-680
class EltwiseMaxLayer : public Layer {
public:
    EltwiseMaxLayer(ConvNet* convNet, PyObject* paramsDict) : Layer(convNet, paramsDict) {}

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
};

// Image-preprocessing layers (layer.cuh:267-328, layer.cu:770-867).  Resize and the colour transforms have no
// backward pass (the reference asserts; its parser refuses gradient consumers below them, layer.py:399-403).
class NailbedLayer : public Layer {
public:
    NailbedLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    int _channels, _start, _stride, _outputsX, _imgSize;
};

class GaussianBlurLayer : public Layer {
public:
    GaussianBlurLayer(ConvNet* convNet, PyObject* paramsDict);
    void copyToGPU() override;

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    int _channels, _imgSize, _radius;
    std::vector<float> _taps;   // host copy of the (1, 2*radius+1) filter, taken in the constructor
    DevMat _filter, _actGradsTmp;
};

class ResizeLayer : public Layer {
public:
    ResizeLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    int _channels, _imgSize, _tgtSize;
    float _scale;
};

class RGBToYUVLayer : public Layer {
public:
    RGBToYUVLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    int _imgPixels;
};

class RGBToLABLayer : public RGBToYUVLayer {
public:
    RGBToLABLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    bool _center;
};

class DataLayer : public Layer {
public:
    DataLayer(ConvNet* convNet, PyObject* paramsDict);
    void fprop(PASS_TYPE passType) override;
    void fprop(DevMatV& data, PASS_TYPE passType) override;
    bool isGradProducer() override { return false; }
    int dataIdx() const { return _dataIdx; }

protected:
    void fpropActs(int, float, PASS_TYPE) override {}
    void bpropActs(DevMat&, int, float, PASS_TYPE) override {}
    int _dataIdx;
};

class PoolLayer : public Layer {
public:
    PoolLayer(ConvNet* convNet, PyObject* paramsDict);
    static PoolLayer& makePoolLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    int _channels, _sizeX, _start, _stride, _outputsX, _imgSize;
    std::string _pool;
};

class AvgPoolLayer : public PoolLayer {
public:
    AvgPoolLayer(ConvNet* convNet, PyObject* paramsDict) : PoolLayer(convNet, paramsDict) {}

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
};

class MaxPoolLayer : public PoolLayer {
public:
    MaxPoolLayer(ConvNet* convNet, PyObject* paramsDict) : PoolLayer(convNet, paramsDict) {}
    void abortPass() override { PoolLayer::abortPass(); _dropNext = nullptr; _reluNext = nullptr; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    NeuronLayer* _dropNext = nullptr;   // this pass: the dropout layer above whose forward step ran inside this layer's
    NeuronLayer* _reluNext = nullptr;   // this pass: the in-place relu above, applied in this layer's store
    DevMat _keep;                       // its keep bytes (one per four elements), held in a float matrix
};

class ResponseNormLayer : public Layer {
public:
    ResponseNormLayer(ConvNet* convNet, PyObject* paramsDict);
    bool bpropOverwritesActs() const override { return true; }   // mc_conv_response_norm*_undo, all of this family

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void truncBwdActs() override;
    int _channels, _size, _imgSize;
    float _scale, _pow;
    DevMat _denoms;
};

class CrossMapResponseNormLayer : public ResponseNormLayer {
public:
    CrossMapResponseNormLayer(ConvNet* convNet, PyObject* paramsDict);

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    bool _blocked;
    float _k;
};

class ContrastNormLayer : public ResponseNormLayer {
public:
    ContrastNormLayer(ConvNet* convNet, PyObject* paramsDict) : ResponseNormLayer(convNet, paramsDict) {}

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void truncBwdActs() override;
    DevMat _meanDiffs;
};

// ------------------------------------------------------------------------------------------
// Cost layers (layer.cuh:330-383, layer.cu:948-1036).  A cost layer writes its values for
// minibatch m into device slot m; ConvNet reads all slots once per batch.
// ------------------------------------------------------------------------------------------
class CostLayer : public Layer {
public:
    CostLayer(ConvNet* convNet, PyObject* paramsDict);
    ~CostLayer();
    void bprop(PASS_TYPE passType) override;
    bool isGradProducer() override { return _coeff != 0; }
    float getCoeff() const { return _coeff; }
    virtual int numCostValues() const = 0;
    // Host-side fix-up of the raw device sums of one minibatch (e.g. negate, numCases - x).
    virtual void finishCost(double* vals, int numCases) const {}
    double* costSlot();          // device address of this minibatch's values
    static CostLayer& makeCostLayer(ConvNet* convNet, const std::string& type, PyObject* paramsDict);
    static const int MAX_SLOTS = 4096;
    // two more slots for minibatches that run as a captured graph (ConvNet::trainStep): the slot such a step writes, and
    // the running total the graph adds it to.  finishCost must therefore be additive over minibatches (all of them are).
    static const int GRAPH_SLOT = MAX_SLOTS, GRAPH_TOTAL = MAX_SLOTS + 1;
    double* deviceSlots() const { return _slots; }

protected:
    float _coeff;
    double* _slots = nullptr;    // MAX_SLOTS * numCostValues() doubles on the device
};

class LogregCostLayer : public CostLayer {
public:
    LogregCostLayer(ConvNet* convNet, PyObject* paramsDict) : CostLayer(convNet, paramsDict) {}
    int numCostValues() const override { return 2; }
    void finishCost(double* vals, int numCases) const override;
    // The softmax layer below runs softmax + this cost (+ the gradient into `grad`, nullable) as one launch and tells the
    // cost so: its own fpropActs then has nothing left to do this pass.
    void fusedTail(const float* labels, const float* in, float* probs, float* grad, int numCases, int numOut);
    void abortPass() override { CostLayer::abortPass(); _fusedThisPass = false; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    DevMat _correctProbs;
    bool _fusedThisPass = false;
};

class SumOfSquaresCostLayer : public CostLayer {
public:
    SumOfSquaresCostLayer(ConvNet* convNet, PyObject* paramsDict) : CostLayer(convNet, paramsDict) {}
    int numCostValues() const override { return 1; }
    void finishCost(double* vals, int numCases) const override { vals[0] *= 0.5; }

protected:
    void fpropActs(int inpIdx, float scaleTargets, PASS_TYPE passType) override;
    void bpropActs(DevMat& v, int inpIdx, float scaleTargets, PASS_TYPE passType) override;
};

// ------------------------------------------------------------------------------------------
// Uniform [0,1) numbers for stochastic neurons (the reference's NVMatrix::randomizeUniform,
// nvmatrix.cu:271-300: XORWOW states seeded from the wall clock).  Here: a counter-based Philox
// stream (seed, running offset), so a run is reproducible from its seed; tests may queue
// explicit masks, which are consumed first, in fprop order.
// ------------------------------------------------------------------------------------------
void fillUniform(DevMat& m);
// The same stream without the matrix: reserves n positions and returns where they start, for kernels that
// generate the uniforms themselves (mc_dropout_counter).  False when an injected mask is pending: the caller
// must then take the fillUniform path, which consumes it.
bool reserveUniform(size_t n, unsigned long long* seed, unsigned long long* offset);
unsigned long long randDraws();   // how many times fillUniform / reserveUniform have been called
void injectUniform(const float* host, size_t n);
void setRandSeed(unsigned long long seed);

// ------------------------------------------------------------------------------------------
// Plugin ABI (Kernel/include/convnet/plugin.cuh:33-41): a Plugins/*.so exports
//   extern "C" std::map<std::string, layerConFunc>  layerConstructor();
//   extern "C" std::map<std::string, neuronConFunc> neuronConstructor();
// keyed by the `type` strings of the layer-def file.
// ------------------------------------------------------------------------------------------
typedef Layer* (*layerConFunc)(ConvNet*, PyObject*);
typedef Neuron* (*neuronConFunc)(PyObject*);
std::map<std::string, layerConFunc>& pluginLayers();
std::map<std::string, neuronConFunc>& pluginNeurons();
// Scans `dir` (default "Plugins" next to the module, then ./Plugins as the reference does,
// plugin.cu:40-70) and merges every plugin's constructors.  Returns the number of files loaded.
int loadPlugins(const char* moduleDir);

}  // namespace rt
