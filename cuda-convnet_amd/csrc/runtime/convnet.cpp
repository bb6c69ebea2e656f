// convnet.cpp -- model object, device thread, data staging, cost read-back, gradient check,
// work items, gradient all-reduce.  Reference: Kernel/src/convnet/convnet.cu, worker.cu,
// Kernel/src/data/data.cu (cited per block).
#include "convnet.h"

#include <dlfcn.h>
#include <math.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <functional>
#include <iostream>

namespace rt {

static const size_t kBucketElems = 256 * 1024;   // smallest all-reduce worth a launch of its own: 1 MiB of fp32
static inline mcStream_t ST() { return reinterpret_cast<mcStream_t>(computeStream()); }

// ==========================================================================================
// random numbers for stochastic neurons
// ==========================================================================================
namespace {
std::mutex g_randMutex;
std::deque<std::vector<float>> g_injected;
unsigned long long g_seed = 0, g_offset = 0;
bool g_seeded = false;
std::atomic<unsigned long long> g_randDraws{0};   // calls that took (or injected) random numbers: a step that draws cannot be replayed
}  // namespace

unsigned long long randDraws() { return g_randDraws.load(std::memory_order_relaxed); }

// A dry run (ConvNet::warmUp) must leave the random stream where it found it: the hold sets injected masks aside (the dry
// run draws from the counter-based stream instead) and its destructor puts masks, stream position and draw count back.
namespace {
struct RandHold {
    std::deque<std::vector<float>> injected;
    unsigned long long offset, draws;
    RandHold() {
        std::lock_guard<std::mutex> g(g_randMutex);
        injected.swap(g_injected);
        offset = g_offset;
        draws = g_randDraws.load(std::memory_order_relaxed);
    }
    ~RandHold() {
        std::lock_guard<std::mutex> g(g_randMutex);
        injected.swap(g_injected);
        g_offset = offset;
        g_randDraws.store(draws, std::memory_order_relaxed);
    }
};
}  // namespace

void setRandSeed(unsigned long long seed) {
    std::lock_guard<std::mutex> g(g_randMutex);
    g_seed = seed;
    g_offset = 0;
    g_seeded = true;
}

void injectUniform(const float* host, size_t n) {
    std::lock_guard<std::mutex> g(g_randMutex);
    g_injected.emplace_back(host, host + n);
}

void fillUniform(DevMat& m) {
    g_randDraws.fetch_add(1, std::memory_order_relaxed);
    std::unique_lock<std::mutex> g(g_randMutex);
    if (!g_injected.empty()) {
        std::vector<float> mask;
        mask.swap(g_injected.front());
        g_injected.pop_front();
        g.unlock();
        if (mask.size() != m.numel()) throw std::runtime_error(fmt("injected mask has %zu elements, the layer needs %zu", mask.size(), m.numel()));
        RT_HIP(hipMemcpyAsync(m.data(), mask.data(), mask.size() * sizeof(float), hipMemcpyHostToDevice, computeStream()));
        RT_HIP(hipStreamSynchronize(computeStream()));   // `mask` dies at scope exit
        return;
    }
    if (!g_seeded) {
        g_seed = (unsigned long long)time(nullptr);
        g_seeded = true;
    }
    const unsigned long long off = g_offset;
    g_offset += (m.numel() + 3) & ~(size_t)3;   // stream positions stay multiples of 4 (one Philox block)
    g.unlock();
    RT_MC(mc_rand_uniform(m.data(), m.numel(), g_seed, off, ST()));
}

bool reserveUniform(size_t n, unsigned long long* seed, unsigned long long* offset) {
    g_randDraws.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> g(g_randMutex);
    if (!g_injected.empty()) return false;
    if (!g_seeded) {
        g_seed = (unsigned long long)time(nullptr);
        g_seeded = true;
    }
    *seed = g_seed;
    *offset = g_offset;
    g_offset += (n + 3) & ~(size_t)3;
    return true;
}

// ==========================================================================================
// Cost
// ==========================================================================================
void Cost::add(const std::string& name, float coeff, const double* vals, int n) {
    std::vector<double>& v = _costMap[name];
    if ((int)v.size() < n) v.resize(n, 0.0);
    for (int i = 0; i < n; ++i) v[i] += vals[i];
    _coeffs[name] = coeff;
}

double Cost::getValue() const {
    double val = 0;
    for (auto& kv : _costMap) val += _coeffs.at(kv.first) * kv.second.at(0);
    return val;
}

// ==========================================================================================
// DataProvider                                                             (data.cu:32-121)
// ==========================================================================================
DataProvider::~DataProvider() {
    for (DevMat* m : _data) delete m;
    for (DevMat* m : _mini) delete m;
}

void DataProvider::setData(HostMatV& hData) {
    _hData = &hData;
    if (hData.empty()) throw std::runtime_error("empty data list");
    const int cases = hData[0]->cols();
    for (size_t i = 0; i < hData.size(); i++) {
        if (hData[i]->fortran()) throw std::runtime_error("data matrices must be C-ordered (dims, cases)");
        if (hData[i]->cols() != cases) throw std::runtime_error("data matrices disagree on the number of cases");
        if (i >= _data.size()) {
            _data.push_back(new DevMat());
            _mini.push_back(new DevMat());
        }
        _data[i]->copyFromHost(hData[i]->data(), hData[i]->rows(), hData[i]->cols());
    }
}

int DataProvider::getNumCases() const {
    if (_data.empty() || _data[0]->cols() <= 0) throw std::runtime_error("no data on the device");
    return _data[0]->cols();
}

int DataProvider::getNumMinibatches() const { return (getNumCases() + _minibatchSize - 1) / _minibatchSize; }

int DataProvider::getNumCasesInMinibatch(int idx) const {
    return std::min(_minibatchSize, std::max(0, getNumCases() - idx * _minibatchSize));
}

DevMatV& DataProvider::getMinibatch(int idx) {
    if (idx < 0 || idx >= getNumMinibatches()) throw std::runtime_error(fmt("minibatch %d out of range", idx));
    return getDataSlice(idx * _minibatchSize, (idx + 1) * _minibatchSize);
}

DevMatV& DataProvider::getDataSlice(int startCase, int endCase) {
    const int cases = getNumCases();
    endCase = std::min(endCase, cases);
    const int n = endCase - startCase;
    if (n <= 0) throw std::runtime_error("empty data slice");
    _miniV.clear();
    for (size_t i = 0; i < _data.size(); i++) {
        DevMat& src = *_data[i];
        DevMat& dst = *_mini[i];
        dst.resize(src.rows(), n);
        // column range of a row-major matrix -> dense (rows, n)
        RT_HIP(hipMemcpy2DAsync(dst.data(), (size_t)n * sizeof(float), src.data() + startCase, (size_t)cases * sizeof(float), (size_t)n * sizeof(float),
                                src.rows(), hipMemcpyDeviceToDevice, computeStream()));
        _miniV.push_back(&dst);
    }
    return _miniV;
}

// ==========================================================================================
// GradComm: RCCL through dlopen, so the module also loads on machines without RCCL and shares
// whichever librccl the process already has (PyTorch bundles one with the same soname).
// ==========================================================================================
namespace {
struct NcclUniqueId { char internal[128]; };
typedef int (*fnGetUniqueId)(NcclUniqueId*);
typedef int (*fnCommInitRank)(void**, int, NcclUniqueId, int);
typedef int (*fnAllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fnCommDestroy)(void*);
typedef const char* (*fnGetErrorString)(int);
struct Rccl {
    void* h = nullptr;
    fnGetUniqueId getUniqueId = nullptr;
    fnCommInitRank commInitRank = nullptr;
    fnAllReduce allReduce = nullptr;
    fnCommDestroy commDestroy = nullptr;
    fnGetErrorString getErrorString = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    if (r.h) return r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) throw std::runtime_error(std::string("unable to load librccl: ") + dlerror());
    r.getUniqueId = (fnGetUniqueId)dlsym(r.h, "ncclGetUniqueId");
    r.commInitRank = (fnCommInitRank)dlsym(r.h, "ncclCommInitRank");
    r.allReduce = (fnAllReduce)dlsym(r.h, "ncclAllReduce");
    r.commDestroy = (fnCommDestroy)dlsym(r.h, "ncclCommDestroy");
    r.getErrorString = (fnGetErrorString)dlsym(r.h, "ncclGetErrorString");
    if (!r.getUniqueId || !r.commInitRank || !r.allReduce || !r.commDestroy) throw std::runtime_error("librccl lacks an expected symbol");
    return r;
}
void nccl_check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(fmt("%s failed: %s", what, rccl().getErrorString ? rccl().getErrorString(rc) : "?"));
}
const int kNcclFloat = 7, kNcclSum = 0;   // ncclFloat32, ncclSum (rccl.h)
}  // namespace

std::string GradComm::newUniqueId() {
    NcclUniqueId id;
    nccl_check(rccl().getUniqueId(&id), "ncclGetUniqueId");
    return std::string(id.internal, sizeof(id.internal));
}

void GradComm::init(int rank, int world, const std::string& uniqueId, int device) {
    if (_comm) throw std::runtime_error("data-parallel communicator already initialised");
    if (uniqueId.size() != sizeof(NcclUniqueId)) throw std::runtime_error("RCCL unique id must be 128 bytes");
    NcclUniqueId id;
    memcpy(id.internal, uniqueId.data(), sizeof(id.internal));
    RT_HIP(hipSetDevice(device));
    nccl_check(rccl().commInitRank(&_comm, world, id, rank), "ncclCommInitRank");
    RT_HIP(hipStreamCreateWithFlags(&_stream, hipStreamNonBlocking));
    RT_HIP(hipEventCreateWithFlags(&_ready, hipEventDisableTiming));
    _rank = rank;
    _world = world;
}

GradComm::~GradComm() {
    // the communicator lives for the life of the process, like the model (pyconvnet.cu:47)
}

void GradComm::allReduceAsync(float* p, size_t n, hipStream_t producer, hipStream_t alsoAfter) {
    if (!_comm || n == 0) return;
    // the side stream picks up after everything the producing stream has queued so far (and, when a step has gradients
    // queued on two streams, after the other one too)
    if (alsoAfter && alsoAfter != producer) {
        RT_HIP(hipEventRecord(_ready, alsoAfter));
        RT_HIP(hipStreamWaitEvent(_stream, _ready, 0));
    }
    RT_HIP(hipEventRecord(_ready, producer));
    RT_HIP(hipStreamWaitEvent(_stream, _ready, 0));
    nccl_check(rccl().allReduce(p, p, n, kNcclFloat, kNcclSum, _comm, _stream), "ncclAllReduce");
    _stats.calls++;
    _stats.bytes += (double)n * sizeof(float);
    _pending = true;
}

void GradComm::waitAll(hipStream_t consumer) {
    if (!_pending) return;
    if (_events.empty()) {
        hipEvent_t e;
        RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        _events.push_back(e);
    }
    // how long the compute stream stalls here = what back-propagation did not hide; read back in takeStats()
    const bool timed = _waitUsed < _waitPairs.size() || _waitPairs.size() < 4096;
    if (timed && _waitUsed == _waitPairs.size()) {
        std::pair<hipEvent_t, hipEvent_t> pr;
        RT_HIP(hipEventCreate(&pr.first));
        RT_HIP(hipEventCreate(&pr.second));
        _waitPairs.push_back(pr);
    }
    if (timed) RT_HIP(hipEventRecord(_waitPairs[_waitUsed].first, consumer));
    RT_HIP(hipEventRecord(_events[0], _stream));
    RT_HIP(hipStreamWaitEvent(consumer, _events[0], 0));
    if (timed) RT_HIP(hipEventRecord(_waitPairs[_waitUsed++].second, consumer));
    _stats.steps++;
    _pending = false;
}

GradComm::Stats GradComm::takeStats() {
    for (size_t i = 0; i < _waitUsed; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, _waitPairs[i].first, _waitPairs[i].second) == hipSuccess) _stats.waitMs += ms;
    }
    _waitUsed = 0;
    Stats out = _stats;
    _stats = Stats();
    return out;
}

// ==========================================================================================
// ConvNet                                                                 (convnet.cu:46-257)
// ==========================================================================================
ConvNet::ConvNet(PyObject* layerParams, int minibatchSize, int deviceID) : _deviceID(deviceID) {
    _seed = (unsigned long long)time(nullptr);
    const int numLayers = (int)PyList_GET_SIZE(layerParams);
    for (int i = 0; i < numLayers; i++) {
        PyObject* paramsDict = PyList_GET_ITEM(layerParams, i);
        if (!PyDict_Check(paramsDict)) throw std::runtime_error(fmt("layer %d is not a dictionary", i));
        const std::string layerType = pyDictGetString(paramsDict, "type");
        Layer* l = initLayer(layerType, paramsDict);
        intv* inputLayers = pyDictGetIntV(paramsDict, "inputs");
        if (inputLayers) {
            for (int idx : *inputLayers) {
                if (idx < 0 || idx >= i) throw std::runtime_error(fmt("layer '%s': input index %d out of range", l->getName().c_str(), idx));
                l->addPrev(&getLayer(idx));
            }
        }
        delete inputLayers;
    }
    for (Layer* l : _layers)
        for (Layer* p : l->getPrev()) p->addNext(l);
    for (Layer* l : _layers) l->postInit();
    for (Layer* l : _layers) _conserveMem |= l->conservesMem();
    _dp = new DataProvider(minibatchSize);
}

ConvNet::~ConvNet() {}

Layer* ConvNet::initLayer(const std::string& layerType, PyObject* paramsDict) {
    Layer* l;
    if (layerType == "fc") l = new FCLayer(this, paramsDict);
    else if (layerType == "conv") l = new ConvLayer(this, paramsDict);
    else if (layerType == "local") l = new LocalUnsharedLayer(this, paramsDict);
    else if (layerType == "pool") l = &PoolLayer::makePoolLayer(this, paramsDict);
    else if (layerType == "rnorm") l = new ResponseNormLayer(this, paramsDict);
    else if (layerType == "cmrnorm") l = new CrossMapResponseNormLayer(this, paramsDict);
    else if (layerType == "cnorm") l = new ContrastNormLayer(this, paramsDict);
    else if (layerType == "softmax") l = new SoftmaxLayer(this, paramsDict);
    else if (layerType == "eltsum") l = new EltwiseSumLayer(this, paramsDict);
    else if (layerType == "eltmax") l = new EltwiseMaxLayer(this, paramsDict);
    else if (layerType == "nailbed") l = new NailbedLayer(this, paramsDict);
    else if (layerType == "blur") l = new GaussianBlurLayer(this, paramsDict);
    else if (layerType == "resize") l = new ResizeLayer(this, paramsDict);
    else if (layerType == "rgb2yuv") l = new RGBToYUVLayer(this, paramsDict);
    else if (layerType == "rgb2lab") l = new RGBToLABLayer(this, paramsDict);
    else if (layerType == "neuron") l = new NeuronLayer(this, paramsDict);
    else if (layerType == "data") {
        DataLayer* d = new DataLayer(this, paramsDict);
        _dataLayers.push_back(d);
        l = d;
    } else if (layerType.compare(0, 5, "cost.") == 0) {
        CostLayer* c = &CostLayer::makeCostLayer(this, layerType, paramsDict);
        _costs.push_back(c);
        l = c;
    } else if (pluginLayers().count(layerType)) {
        l = pluginLayers()[layerType](this, paramsDict);
    } else {
        throw std::runtime_error("Unknown layer type " + layerType +
                                 " (plugin types such as cccp and cost.gsinlabel need Plugins/*.so next to libkernel.so)");
    }
    _layers.push_back(l);
    return l;
}

void ConvNet::start() { _thread = std::thread([this] { run(); }); _thread.detach(); }

void ConvNet::initDevice() {
    int dev = _deviceID;
    if (dev < 0) dev = 0;   // "pick for me": devices of one node are identical
    RT_HIP(hipSetDevice(dev));
    _deviceID = dev;
    char arch[64] = "";
    int cus = 0;
    RT_MC(mc_device_info(arch, sizeof(arch), &cus));
    if (strncmp(arch, "gfx950", 6) != 0)
        throw std::runtime_error(fmt("device %d is '%s'; the kernels of libmckernels.so are built for gfx950 (MI355X) only", dev, arch));
    RT_HIP(hipStreamCreateWithFlags(&_stream, hipStreamNonBlocking));
    setComputeStream(_stream);
    copyToGPU();
    size_t split = 0;
    for (Layer* l : _layers) split = std::max(split, l->splitWorkspaceBytes(_dp->getMinibatchSize()));
    RT_MC(mc_reserve_conv_split_workspace(split, ST()));
    // MC_NO_STEP_FORK=1: every step on one stream.  MC_STEP_FORK=graph: captured steps fork too (measured slower, see below).
    const char* fork = getenv("MC_STEP_FORK");
    _forkEager = getenv("MC_NO_STEP_FORK") == nullptr;
    _forkGraph = _forkEager && fork && strstr(fork, "graph");
    if (_forkEager || _forkGraph) {
        RT_HIP(hipStreamCreateWithFlags(&_branchStream, hipStreamNonBlocking));
        RT_HIP(hipEventCreateWithFlags(&_forkEvent, hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&_joinEvent, hipEventDisableTiming));
    }
}

void ConvNet::run() {
    bool ready = false;
    std::string initError;
    try {
        initDevice();
        ready = true;
    } catch (std::exception& e) {
        initError = e.what();
        std::cout << "Error initialising the device: " << initError << std::endl;
    }
    while (true) {
        Worker* worker = _workerQueue.dequeue();
        try {
            if (!ready) throw std::runtime_error("device initialisation failed: " + initError);
            worker->run();
        } catch (std::exception& e) {
            _resultQueue.enqueue(new WorkResult(WorkResult::ERROR, nullptr, e.what()));
        }
        // the item is deleted by the Python thread, which owns the references it holds
    }
}

void ConvNet::copyToCPU() {
    for (Layer* l : _layers) l->copyToCPU();
}

// Lays every owned tensor out in three arenas (weights, increments, gradients), in the order
// back-propagation completes gradients (last layer first; biases before weights, as
// WeightLayer::bpropCommon runs), each tensor aligned to 256 bytes.
void ConvNet::copyToGPU() {
    std::vector<Weights*> order;
    for (auto it = _layers.rbegin(); it != _layers.rend(); ++it) {
        WeightLayer* wl = dynamic_cast<WeightLayer*>(*it);
        if (!wl) continue;
        order.push_back(&wl->getBiases());
        for (int i = 0; i < wl->getNumWeights(); i++)
            if (wl->getWeights(i).isOwner()) order.push_back(&wl->getWeights(i));
    }
    size_t total = 0;
    std::vector<size_t> offs;
    for (Weights* w : order) {
        offs.push_back(total);
        total += (w->numel() + 63) / 64 * 64;
    }
    _arenaElems = total;
    if (total) {
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_arenaW), total * sizeof(float)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_arenaInc), total * sizeof(float)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_arenaGrad), total * sizeof(float)));
        RT_HIP(hipMemset(_arenaW, 0, total * sizeof(float)));
        RT_HIP(hipMemset(_arenaInc, 0, total * sizeof(float)));
        RT_HIP(hipMemset(_arenaGrad, 0, total * sizeof(float)));
    }
    for (size_t i = 0; i < order.size(); i++) order[i]->bind(_arenaW + offs[i], _arenaInc + offs[i], _arenaGrad + offs[i], offs[i]);
    _allWeights = order;
    std::vector<size_t> ends;
    std::vector<int> contributors;
    for (size_t i = 0; i < order.size(); i++) {
        order[i]->setArenaIndex((int)i);
        ends.push_back(i + 1 < order.size() ? offs[i + 1] : total);
        contributors.push_back(order[i]->contributors());
    }
    _buckets.configure(ends, contributors, kBucketElems);
    for (Layer* l : _layers) l->copyToGPU();
    buildUpdateSpans();
    RT_HIP(hipDeviceSynchronize());
}

void ConvNet::releaseHost() {
    for (Layer* l : _layers) l->releaseHost();
}

float* ConvNet::workspace(size_t bytes) {
    if (_onBranch) {
        if (bytes > _branchWorkspaceBytes) {
            if (_capturing) throw std::runtime_error("weight-gradient branch needs a larger workspace than the eager steps did");
            if (_branchWorkspace) {
                RT_HIP(hipStreamSynchronize(_branchStream));
                RT_HIP(hipFree(_branchWorkspace));
            }
            _branchWorkspaceBytes = (bytes + (4u << 20)) / (4u << 20) * (4u << 20);
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&_branchWorkspace), _branchWorkspaceBytes));
            // captured steps reference the branch workspace only when they fork too (MC_STEP_FORK=graph): then a graph of another
            // shape holds the address just freed and must go; otherwise no graph knows this buffer
            if (_forkGraph) DeviceAllocator::get().noteForeignAllocation();
        }
        return _branchWorkspace;
    }
    if (bytes > _workspaceBytes) {
        if (_workspace) {
            RT_HIP(hipStreamSynchronize(_stream));
            RT_HIP(hipFree(_workspace));
        }
        _workspaceBytes = (bytes + (4u << 20)) / (4u << 20) * (4u << 20);
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_workspace), _workspaceBytes));
        DeviceAllocator::get().noteForeignAllocation();
    }
    return _workspace;
}

unsigned long long ConvNet::nextRandOffset(size_t n) {
    const unsigned long long o = _randOffset;
    _randOffset += n;
    return o;
}

void ConvNet::reset() {
    for (Layer* l : _layers) l->reset();
}

void ConvNet::fprop(PASS_TYPE passType) {
    if (!_data) throw std::runtime_error("fprop without data");
    reset();
    for (DataLayer* d : _dataLayers) d->fprop(*_data, passType);
}

void ConvNet::fprop(DevMatV& data, PASS_TYPE passType) {
    _data = &data;
    fprop(passType);
}

void ConvNet::fprop(int miniIdx, PASS_TYPE passType) {
    _data = &_dp->getMinibatch(miniIdx);
    fprop(passType);
}

void ConvNet::beginWeightGradBranch() {
    RT_HIP(hipEventRecord(_forkEvent, _stream));
    RT_HIP(hipStreamWaitEvent(_branchStream, _forkEvent, 0));
    setComputeStream(_branchStream);
    _onBranch = _branchOpen = true;
}

void ConvNet::endWeightGradBranch() {
    setComputeStream(_stream);
    _onBranch = false;
}

void ConvNet::joinWeightGradBranch() {
    endWeightGradBranch();
    if (!_branchOpen) return;
    _branchOpen = false;
    RT_HIP(hipEventRecord(_joinEvent, _branchStream));
    RT_HIP(hipStreamWaitEvent(_stream, _joinEvent, 0));
}

void ConvNet::bprop(PASS_TYPE passType) {
    _buckets.beginStep();
    for (CostLayer* c : _costs) c->bprop(passType);
    reset();
    if (_branchOpen) joinWeightGradBranch();   // the update (and whoever reads gradients next) waits for the branch
    if (_comm.active() && passType == PASS_TRAIN && !_warming) {
        // whatever has not been sent yet goes now; the update then waits for the side stream
        size_t lo, hi;
        if (_buckets.finish(&lo, &hi)) _comm.allReduceAsync(_arenaGrad + lo, hi - lo, _stream);
        _comm.waitAll(_stream);
    }
}

// A tensor's gradient is complete once every layer that adds into it has queued its product; complete tensors are reduced
// on the side stream in buckets of >= kBucketElems while the rest of back-propagation runs (buckets.h has the rules).
void ConvNet::gradientReady(Weights& w) {
    if (!_comm.active() || _warming || w.arenaIndex() < 0) return;
    size_t lo, hi;
    // (the producer is the stream the gradient was just queued on: the weight-gradient branch when a step forks; a bucket
    // may also hold tensors queued earlier on that same branch, and the branch runs in order)
    // (a first layer's gradient is queued on the main stream while the branch is open, see WeightLayer::bpropCommon: then
    // the bucket may hold tensors of both streams and the all-reduce waits for both)
    if (_buckets.arrived((size_t)w.arenaIndex(), &lo, &hi))
        _comm.allReduceAsync(_arenaGrad + lo, hi - lo, computeStream(), _branchOpen ? _branchStream : nullptr);
}

// One launch updates every tensor (spans of <= 4096 elements, built once); tensors that do not
// learn (eps == 0) have no spans, exactly as Weights::update skips them.
void ConvNet::buildUpdateSpans() {
    std::vector<mcUpdateSpan> spans;
    for (Weights* w : _allWeights) {
        if (w->getEps() <= 0) continue;
        const float wceps = w->getWC() > 0 ? -w->getWC() * w->getEps() : 0.f;
        for (size_t done = 0; done < w->numel(); done += 4096) {
            mcUpdateSpan sp;
            sp.start = w->arenaOffset() + done;
            sp.count = (unsigned)std::min<size_t>(4096, w->numel() - done);
            sp.mom = w->getMom();
            sp.wceps = wceps;
            sp.pad = 0;
            spans.push_back(sp);
        }
    }
    _numSpans = (int)spans.size();
    if (_numSpans) {
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_spans), spans.size() * sizeof(mcUpdateSpan)));
        RT_HIP(hipMemcpy(_spans, spans.data(), spans.size() * sizeof(mcUpdateSpan), hipMemcpyHostToDevice));
    }
}

void ConvNet::updateWeights() {
    if (_numSpans) {
        profileSetLayer("net");
        profileSetPhase("update");
        RT_MC(mc_weights_update_spans(_arenaW, _arenaInc, _arenaGrad, _spans, _numSpans, reinterpret_cast<mcStream_t>(_stream)));
        for (Weights* w : _allWeights) w->markUpdated();
        return;
    }
    for (Layer* l : _layers) l->updateWeights();
}

void ConvNet::endMinibatchCosts(int numCases) {
    _slotCases.push_back(numCases);
    _slot++;
}

Cost* ConvNet::collectCosts() {
    RT_HIP(hipStreamSynchronize(_stream));
    int total = _graphCases;
    for (int n : _slotCases) total += n;
    Cost* cost = new Cost(total);
    for (CostLayer* c : _costs) {
        const int nv = c->numCostValues();
        std::vector<double> host((size_t)_slot * nv);
        if (_slot && c->deviceSlots()) RT_HIP(hipMemcpy(host.data(), c->deviceSlots(), host.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> zero(nv, 0.0);
        cost->add(c->getName(), c->getCoeff(), zero.data(), nv);
        for (int s = 0; s < _slot; s++) {
            c->finishCost(&host[(size_t)s * nv], _slotCases[s]);
            cost->add(c->getName(), c->getCoeff(), &host[(size_t)s * nv], nv);
        }
        if (_graphCases && c->deviceSlots()) {   // minibatches that ran as replayed graphs: one running total (finishCost is additive)
            std::vector<double> sum(nv);
            RT_HIP(hipMemcpy(sum.data(), c->deviceSlots() + (size_t)CostLayer::GRAPH_TOTAL * nv, nv * sizeof(double), hipMemcpyDeviceToHost));
            c->finishCost(sum.data(), _graphCases);
            cost->add(c->getName(), c->getCoeff(), sum.data(), nv);
        }
    }
    return cost;
}

int ConvNet::costSlotIndex() const { return _capturing ? (int)CostLayer::GRAPH_SLOT : _slot; }

void ConvNet::beginBatchCosts() {
    _slot = 0;
    _slotCases.clear();
    _graphCases = 0;
    for (CostLayer* c : _costs)
        if (c->deviceSlots())
            RT_HIP(hipMemsetAsync(c->deviceSlots() + (size_t)CostLayer::GRAPH_TOTAL * c->numCostValues(), 0, sizeof(double) * c->numCostValues(), _stream));
}

// ------------------------------------------------------------------------------------------
// One training step.  Small nets are bound by launches (layers-80sec: ~35 kernels of 3-30 us per step), so a step whose
// launches are the same every time -- same minibatch shape, no random numbers, no communicator, no per-launch profiling --
// is captured into a HIP graph the third time its shape comes up (the first two eager runs size every buffer and let the
// contraction autotuner settle) and replayed from then on: one launch per step plus the staging copies of the minibatch.
// Whatever differs from step to step stays outside the graph: the minibatch is staged into fixed matrices before the
// launch, and the costs go to a fixed slot that the graph adds to a running total (read once per batch).
// Whether replaying pays is measured, not assumed: the four steps after the capture alternate between the graph and an eager
// step, timed with events, and the faster way is kept for the shape (both are training steps with bit-identical results,
// so nothing is run twice).  The graph wins where the host cannot keep up with ~35 launches of 3-30 us; eager steps win on
// larger nets because they fork (next paragraph): imagenet-224 6.16 ms replayed, 5.97 eager; conv-local 0.475 / 0.452.
// MC_NO_STEP_GRAPH=1 keeps every step eager, MC_STEP_GRAPH=1 replays whenever a graph exists.
//
// Eager steps fork: a layer's weight and bias gradient and
// the rest of back-propagation below it read the same gradient matrix and write disjoint ones, so the weight-gradient
// products run on a second stream (WeightLayer::bpropCommon forks it off the main stream with an event, ConvNet::bprop
// joins it before the update; its split-K workspace is its own).  The k-contiguous weight gradients keep one workgroup per
// CU waiting on its loads and the backward-data products fill the matrix pipe, so the two overlap well: NIN 1.914 -> 1.852
// ms/step, imagenet-224 6.17 -> 5.87.  A step forks once its shape has run twice without allocating (every matrix sized,
// contraction tiles tuned on an undisturbed stream) and never while every launch is being timed (setProfile("*")).
// Results are bit-identical: the same kernels write the same buffers in the same order per buffer
// (test_forked_steps_equal_single_stream_steps).  Captured steps do NOT fork by default: as branches of a HIP graph the same
// split costs more in cross-branch dependencies than it gains on these small nets (layers-80sec 0.297 -> 0.325 ms/step), although
// two such products issued on two plain streams do overlap (tools/overlap_probe.py: 49 -> 35 us per pair).
// MC_NO_STEP_FORK=1 keeps every step on one stream.
// ------------------------------------------------------------------------------------------
bool ConvNet::stepGraphsAllowed() const {
    static const bool off = getenv("MC_NO_STEP_GRAPH") != nullptr;
    return !off && !_stochastic && !_comm.active() && !profileOn() && !_conserveMem;
}

void ConvNet::eagerStep(DevMatV& mb, int cases) {
    fprop(mb, PASS_TRAIN);
    endMinibatchCosts(cases);
    bprop(PASS_TRAIN);
    updateWeights();
}

bool ConvNet::captureStep(StepGraph& g, DevMatV& mb) {
    hipGraph_t graph = nullptr;
    if (_forkGraph && _branchWorkspaceBytes < _workspaceBytes) {   // as large as anything the eager steps asked for
        if (_branchWorkspace) RT_HIP(hipFree(_branchWorkspace));
        _branchWorkspaceBytes = _workspaceBytes;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&_branchWorkspace), _branchWorkspaceBytes));
        DeviceAllocator::get().noteForeignAllocation();
    }
    _forkThisStep = _forkGraph;
    bool ok = hipStreamBeginCapture(_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        _capturing = true;
        (void)mc_set_autotune(0);
        try {
            fprop(mb, PASS_TRAIN);
            for (CostLayer* c : _costs)
                if (c->deviceSlots())
                    RT_MC(mc_accumulate_f64(c->deviceSlots() + (size_t)CostLayer::GRAPH_TOTAL * c->numCostValues(),
                                            c->deviceSlots() + (size_t)CostLayer::GRAPH_SLOT * c->numCostValues(), c->numCostValues(), ST()));
            bprop(PASS_TRAIN);
            if (getenv("MC_TEST_FAIL_CAPTURE")) throw std::runtime_error("MC_TEST_FAIL_CAPTURE");   // test hook: fail after bprop, before the update
            updateWeights();
        } catch (std::exception& e) {
            ok = false;
            g.why = e.what();
        }
        if (_branchOpen || _onBranch) {   // an exception left the branch open: join it so that the capture can end
            try { joinWeightGradBranch(); } catch (std::exception&) { endWeightGradBranch(); _branchOpen = false; }
        }
        _forkThisStep = false;
        (void)mc_set_autotune(1);
        _capturing = false;
        if (hipStreamEndCapture(_stream, &graph) != hipSuccess || !graph) ok = false;
    }
    if (ok && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        ok = false;
        g.exec = nullptr;
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();   // the failed capture's sticky error is not the next launch's
        g.bad = true;
        // Nothing ran, but the host side walked part of a step: put back what an eager step relies on.  Without this a weight
        // tensor whose gradient product was "queued" keeps _numUpdates > 0 and the eager step that follows ADDS onto the stale
        // gradient of the step before; arrival counters left half way skip or repeat layers.
        for (Layer* l : _layers) l->abortPass();
        for (Weights* w : _allWeights) w->markUpdated();
        if (getenv("MC_STEP_GRAPH_VERBOSE")) fprintf(stderr, "[libkernel] step graph not captured (%s): steps of this shape stay eager\n", g.why.c_str());
    }
    return ok;
}

// Dry run of one training step per minibatch shape of a batch, before its first real step (TrainingWorker / ResidentWorker):
// forward and backward on the shape's first minibatch, no update.  It sizes every matrix and lets the contraction autotuner
// time its candidates for the full AND the ragged-tail shape up front, so that no step of the batch carries timing trials or
// allocations.  Nothing it does survives: gradients are overwritten by the next step (accumulation flags cleared), the cost
// slot is cleared, the random stream and injected masks are put back, nothing is all-reduced.  MC_NO_WARMUP=1 skips it.
void ConvNet::warmUpShapes() {
    static const bool off = getenv("MC_NO_WARMUP") != nullptr;
    if (off || profileOn()) return;
    for (int i = 0; i < _dp->getNumMinibatches(); i++) {
        StepGraph& g = _stepGraphs[_dp->getNumCasesInMinibatch(i)];
        if (g.warmed || g.eagerRuns > 0) continue;
        g.warmed = true;
        RandHold hold;
        _warming = true;
        try {
            fprop(i, PASS_TRAIN);
            bprop(PASS_TRAIN);
        } catch (...) {
            _warming = false;
            throw;
        }
        _warming = false;
        _warmedShapes++;
        for (Weights* w : _allWeights) w->markUpdated();
        for (CostLayer* c : _costs)
            if (c->deviceSlots())
                RT_HIP(hipMemsetAsync(c->deviceSlots() + (size_t)_slot * c->numCostValues(), 0, sizeof(double) * c->numCostValues(), _stream));
    }
}

void ConvNet::runEagerStep(StepGraph& g, DevMatV& mb, int cases, unsigned long long stamp) {
    const unsigned long long draws = randDraws();
    // an eager step forks once its shape has run twice without allocating (matrices sized, contraction tiles tuned)
    _forkThisStep = _forkEager && g.eagerRuns >= 2 && g.allocStamp == stamp && !_conserveMem && !profileEveryLaunch();
    try {
        eagerStep(mb, cases);
    } catch (...) {
        if (_branchOpen || _onBranch) { endWeightGradBranch(); _branchOpen = false; }
        _forkThisStep = false;
        throw;
    }
    _forkedSteps += _forkThisStep;
    _forkThisStep = false;
    // With a captured graph alive the stamp must stay the capture's: re-stamping after an eager step that (re)allocated would make
    // the graph's stale addresses look current to the next trainStep, which compares the two and drops the graph on a mismatch.
    if (!g.exec) g.allocStamp = DeviceAllocator::get().allocations();
    if (randDraws() != draws) _stochastic = true;   // dropout & co. take new random numbers every step
    g.eagerRuns++;
}

void ConvNet::trainStep(int miniIdx) {
    const int cases = _dp->getNumCasesInMinibatch(miniIdx);
    DevMatV& mb = _dp->getMinibatch(miniIdx);   // staged into the provider's fixed matrices: same addresses every step
    StepGraph& g = _stepGraphs[cases];
    const unsigned long long stamp = DeviceAllocator::get().allocations();
    if (g.exec && g.allocStamp != stamp) {   // some matrix has moved since the capture: the graph's addresses are stale
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        g.eagerRuns = 1;
        g.mode = StepGraph::UNDECIDED;
        g.trials = 0;
        g.graphMs = g.eagerMs = 1e30f;
    }
    // MC_STEP_GRAPH=1: replay whenever a graph exists (no trial); MC_NO_STEP_GRAPH=1: never capture
    static const bool forceGraph = getenv("MC_STEP_GRAPH") != nullptr;
    bool replay = false, trial = false;
    if (stepGraphsAllowed() && !g.bad) {
        if (!g.exec && g.eagerRuns >= 2) {
            captureStep(g, mb);   // nothing runs during the capture
            g.allocStamp = DeviceAllocator::get().allocations();
            if (g.exec && g.allocStamp != stamp) {   // the step still allocates: not a shape to replay yet
                (void)hipGraphExecDestroy(g.exec);
                g.exec = nullptr;
                g.eagerRuns = 1;
            }
        }
        if (g.exec) {
            if (forceGraph) g.mode = StepGraph::GRAPH;
            trial = g.mode == StepGraph::UNDECIDED;
            replay = trial ? g.trials % 2 == 0 : g.mode == StepGraph::GRAPH;
        }
    }
    if (trial) {
        if (!_trialBegin) {
            RT_HIP(hipEventCreate(&_trialBegin));
            RT_HIP(hipEventCreate(&_trialEnd));
        }
        RT_HIP(hipEventRecord(_trialBegin, _stream));
    }
    if (replay) {
        RT_HIP(hipGraphLaunch(g.exec, _stream));
        _graphCases += cases;
        _graphReplays++;
        for (Weights* w : _allWeights) w->markUpdated();
    } else {
        runEagerStep(g, mb, cases, stamp);
    }
    if (trial) {   // two timed steps each way (a step is a step: both update the weights), then the faster way for good
        float ms = 0.f;
        RT_HIP(hipEventRecord(_trialEnd, _stream));
        RT_HIP(hipEventSynchronize(_trialEnd));
        RT_HIP(hipEventElapsedTime(&ms, _trialBegin, _trialEnd));
        float& best = replay ? g.graphMs : g.eagerMs;
        best = std::min(best, ms);
        if (++g.trials == 4) {
            g.mode = g.graphMs <= g.eagerMs ? StepGraph::GRAPH : StepGraph::EAGER;
            if (getenv("MC_STEP_GRAPH_VERBOSE"))
                fprintf(stderr, "[libkernel] steps of %d cases: replayed graph %.3f ms, eager %.3f ms -> %s\n", cases, g.graphMs, g.eagerMs,
                        g.mode == StepGraph::GRAPH ? "graph" : "eager");
        }
    }
}

double ConvNet::getCostValue() {
    RT_HIP(hipStreamSynchronize(_stream));
    double val = 0;
    const int numCases = _data ? (*_data)[0]->cols() : 0;
    for (CostLayer* c : _costs) {
        const int nv = c->numCostValues();
        std::vector<double> host(nv, 0.0);
        if (c->deviceSlots()) RT_HIP(hipMemcpy(host.data(), c->deviceSlots() + (size_t)_slot * nv, nv * sizeof(double), hipMemcpyDeviceToHost));
        c->finishCost(host.data(), numCases);
        val += c->getCoeff() * host[0];
    }
    return val;
}

// ------------------------------------------------------------------------------------------
// Numerical gradient check                                               (convnet.cu:262-335)
// ------------------------------------------------------------------------------------------
static const float GC_REL_ERR_THRESH = 0.02f;   // Kernel/include/util.cuh:46

int ConvNet::checkGradients() {
    _numFailures = 0;
    _numTests = 0;
    beginBatchCosts();
    fprop(0, PASS_GC);
    _baseErr = getCostValue();
    bprop(PASS_GC);
    for (Layer* l : _layers) l->checkGradients();
    std::cout << "------------------------" << std::endl;
    if (_numFailures > 0) std::cout << _numFailures << "/" << _numTests << " TESTS FAILED" << std::endl;
    else std::cout << "ALL " << _numTests << " TESTS PASSED" << std::endl;
    return _numFailures;
}

bool ConvNet::checkGradient(const std::string& name, float eps, Weights& weights) {
    const int rows = weights.getNumRows(), cols = weights.getNumCols();
    const size_t n = weights.numel();
    const int numCases = (*_data)[0]->cols();
    std::vector<float> w(n), numGrad(n, 0.f), grad(n);
    RT_HIP(hipStreamSynchronize(_stream));
    RT_HIP(hipMemcpy(w.data(), weights.getW(), n * sizeof(float), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(grad.data(), weights.getGrad(), n * sizeof(float), hipMemcpyDeviceToHost));   // before any re-fprop
    for (int i = 0; i < rows; i++) {
        for (int j = 0; j < cols; j++) {
            const size_t d = weights.deviceIndex(i, j);
            const float bumped = w[d] + eps;
            RT_HIP(hipMemcpy(weights.getW() + d, &bumped, sizeof(float), hipMemcpyHostToDevice));
            fprop(PASS_GC);
            const double err = getCostValue();
            numGrad[d] = (float)((err - _baseErr) / (numCases * eps));
            if (std::isnan(numGrad[d]) || std::isinf(numGrad[d])) {
                std::cout << "Numerical computation produced nan or inf when checking '" << name << "': " << numGrad[d] << std::endl;
                std::cout << "Consider reducing the sizes of the weights or finite difference steps." << std::endl;
                throw std::runtime_error("gradient check aborted");
            }
            RT_HIP(hipMemcpy(weights.getW() + d, &w[d], sizeof(float), hipMemcpyHostToDevice));
        }
    }
    double analNorm = 0, numNorm = 0, diffNorm = 0;
    for (size_t k = 0; k < n; k++) {
        const double a = -(double)grad[k] / numCases;
        analNorm += a * a;
        numNorm += (double)numGrad[k] * numGrad[k];
        diffNorm += (numGrad[k] - a) * (numGrad[k] - a);
    }
    analNorm = sqrt(analNorm); numNorm = sqrt(numNorm); diffNorm = sqrt(diffNorm);
    const double relErr = diffNorm / analNorm;
    const bool fail = !(relErr < GC_REL_ERR_THRESH);
    std::cout << "========================" << std::endl;
    printf("(%s) %s GRADIENT CHECK\n", fail ? "****FAIL****" : "PASS", name.c_str());
    std::cout << "========================" << std::endl;
    printf("Analytic norm: %e\n", analNorm);
    printf("Numeric norm:  %e\n", numNorm);
    printf("Relative error: %e\n", relErr);
    fflush(stdout);
    _numTests++;
    _numFailures += fail;
    return fail;
}

std::map<std::string, DevMat*> ConvNet::debugActs() {
    std::map<std::string, DevMat*> m;
    for (Layer* l : _layers) {
        try { m[l->getName()] = &l->getActs(); } catch (...) {}
    }
    return m;
}

// ==========================================================================================
// Work items                                                               (worker.cu:61-214)
// ==========================================================================================
DataWorker::~DataWorker() {
    if (_data) {
        for (HostMat* m : *_data) delete m;
        delete _data;
    }
}

void DataWorker::releaseHost() {
    if (_data)
        for (HostMat* m : *_data) m->release();
}

void TrainingWorker::run() {
    DataProvider& dp = _convNet->getDataProvider();
    dp.setData(*_data);
    _convNet->beginBatchCosts();
    const long gmb = (long)dp.getMinibatchSize() * _convNet->world();
    for (int i = 0; i < dp.getNumMinibatches(); i++) {
        // a ragged global tail is dealt out unevenly (dp.shard_columns): its gradient scale is 1 / (its true global size)
        if (_globalCases > 0) _convNet->setGlobalMinibatchCases((int)std::min(gmb, _globalCases - i * gmb));
        if (_test) {
            _convNet->fprop(i, PASS_TEST);
            _convNet->endMinibatchCosts(dp.getNumCasesInMinibatch(i));
        } else {
            if (i == 0) _convNet->warmUpShapes();
            _convNet->trainStep(i);
        }
    }
    _convNet->setGlobalMinibatchCases(0);
    Cost* cost = _convNet->collectCosts();   // synchronises the stream
    dp.clearData();
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::BATCH_DONE, cost));
}

void ResidentWorker::run() {
    DataProvider& dp = _convNet->getDataProvider();
    const int nmb = dp.getNumMinibatches();
    _convNet->beginBatchCosts();
    if (!_test && _count > 0) _convNet->warmUpShapes();
    for (int k = 0; k < _count; k++) {
        const int i = (_first + k) % nmb;
        if (_test) {
            _convNet->fprop(i, PASS_TEST);
            _convNet->endMinibatchCosts(dp.getNumCasesInMinibatch(i));
        } else {
            _convNet->trainStep(i);
        }
    }
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::BATCH_DONE, _convNet->collectCosts()));
}

void SyncWorker::run() {
    _convNet->copyToCPU();
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::SYNC_DONE));
}

void GradCheckWorker::run() {
    _convNet->getDataProvider().setData(*_data);
    const int failures = _convNet->checkGradients();
    WorkResult* r = new WorkResult(WorkResult::BATCH_DONE, new Cost(0));
    r->intValue = failures;
    _convNet->getResultQueue().enqueue(r);
}

void MultiviewTestWorker::run() {
    DataProvider& dp = _convNet->getDataProvider();
    dp.setData(*_data);
    if (dp.getNumCases() % _numViews != 0) throw std::runtime_error("multiview test: cases not divisible by the number of views");
    Layer& logregLayer = _convNet->getLayer(_logregIdx);
    const int numCasesReal = dp.getNumCases() / _numViews;
    const int mb = dp.getMinibatchSize();
    const int numMiniReal = (numCasesReal + mb - 1) / mb;
    _convNet->beginBatchCosts();
    DevMat softmaxActs, labels;
    for (int i = 0; i < numMiniReal; i++) {
        int n = 0;
        for (int v = 0; v < _numViews; v++) {
            DevMatV& mini = dp.getDataSlice(v * numCasesReal + i * mb, std::min((v + 1) * numCasesReal, v * numCasesReal + (i + 1) * mb));
            _convNet->fprop(mini, PASS_TEST);
            DevMat& probs = logregLayer.getPrev()[1]->getActs();
            n = probs.cols();
            if (v == 0) {
                softmaxActs.copyFrom(probs);
                labels.copyFrom(logregLayer.getPrev()[0]->getActs());
            } else {
                RT_MC(mc_axpby(softmaxActs.data(), probs.data(), 1.f, 1.f, probs.numel(), ST()));
            }
        }
        RT_MC(mc_scale(softmaxActs.data(), softmaxActs.data(), 1.f / _numViews, softmaxActs.numel(), ST()));
        DevMatV logregInput;
        logregInput.push_back(&labels);
        logregInput.push_back(&softmaxActs);
        logregLayer.fprop(logregInput, PASS_TEST);
        _convNet->endMinibatchCosts(n);
    }
    Cost* cost = _convNet->collectCosts();
    dp.clearData();
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::BATCH_DONE, cost));
}

void FeatureWorker::releaseHost() {
    DataWorker::releaseHost();
    if (_ftrs) {
        _ftrs->release();
        delete _ftrs;
        _ftrs = nullptr;
    }
}

void FeatureWorker::run() {
    DataProvider& dp = _convNet->getDataProvider();
    dp.setData(*_data);
    Layer& ftrLayer = _convNet->getLayer(_layerIdx);
    if (_ftrs->fortran() || _ftrs->rows() != dp.getNumCases()) throw std::runtime_error("feature matrix must be C-ordered (numCases, layerOutputs)");
    _convNet->beginBatchCosts();
    std::vector<float> host;
    for (int i = 0; i < dp.getNumMinibatches(); i++) {
        _convNet->fprop(i, PASS_TEST);
        const int n = dp.getNumCasesInMinibatch(i);
        _convNet->endMinibatchCosts(n);
        DevMat& acts = ftrLayer.getActs();   // (outputs, n)
        if (acts.rows() != _ftrs->cols()) throw std::runtime_error(fmt("feature matrix has %d columns, layer produces %d outputs", _ftrs->cols(), acts.rows()));
        host.resize(acts.numel());
        acts.copyToHost(host.data());
        float* dst = _ftrs->data() + (size_t)i * dp.getMinibatchSize() * _ftrs->cols();
        for (int c = 0; c < n; c++)
            for (int f = 0; f < acts.rows(); f++) dst[(size_t)c * acts.rows() + f] = host[(size_t)f * n + c];
    }
    Cost* cost = _convNet->collectCosts();
    dp.clearData();
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::BATCH_DONE, cost));
}

void CallWorker::run() {
    _fn();
    _convNet->getResultQueue().enqueue(new WorkResult(WorkResult::SYNC_DONE));
}

}  // namespace rt
