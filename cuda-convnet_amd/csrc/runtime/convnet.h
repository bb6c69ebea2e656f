// convnet.h -- the model object, its worker thread and the work items the Python thread queues.
// Follows Kernel/include/convnet/{convnet,worker,queue,thread,data}.cuh of the reference:
// Python enqueues a Worker, one device thread dequeues and runs it, results come back through
// a second queue (convnet.cu:149-158, queue.h:84-110).
#pragma once
#include "buckets.h"
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "layer.h"

namespace rt {

template <class T>
class Queue {
public:
    void enqueue(T v) {
        {
            std::lock_guard<std::mutex> g(_m);
            _q.push(v);
        }
        _cv.notify_one();
    }
    T dequeue() {
        std::unique_lock<std::mutex> g(_m);
        _cv.wait(g, [&] { return !_q.empty(); });
        T v = _q.front();
        _q.pop();
        return v;
    }

private:
    std::queue<T> _q;
    std::mutex _m;
    std::condition_variable _cv;
};

// name -> values, summed over the minibatches of a batch (Kernel/src/layers/cost.cu:38-107)
typedef std::map<std::string, std::vector<double>> CostMap;
class Cost {
public:
    explicit Cost(int numCases = 0) : _numCases(numCases) {}
    void add(const std::string& name, float coeff, const double* vals, int n);
    void addCases(int n) { _numCases += n; }
    int getNumCases() const { return _numCases; }
    CostMap& getCostMap() { return _costMap; }
    double getValue() const;   // sum_k coeff_k * cost_k[0]
private:
    int _numCases;
    CostMap _costMap;
    std::map<std::string, float> _coeffs;
};

// Host batch -> device minibatches (Kernel/src/data/data.cu:47-121).  The whole batch is
// uploaded once; a minibatch is a column range gathered into a dense (dims, n) staging matrix
// (the kernels take dense (features, numCases) operands).
class DataProvider {
public:
    explicit DataProvider(int minibatchSize) : _minibatchSize(minibatchSize) {}
    ~DataProvider();
    void setData(HostMatV& data);
    void clearData() { _hData = nullptr; }
    DevMatV& getMinibatch(int idx);
    DevMatV& getDataSlice(int startCase, int endCase);
    int getNumMinibatches() const;
    int getMinibatchSize() const { return _minibatchSize; }
    int getNumCases() const;
    int getNumCasesInMinibatch(int idx) const;

private:
    int _minibatchSize;
    HostMatV* _hData = nullptr;
    std::vector<DevMat*> _data;   // whole batch on the device
    std::vector<DevMat*> _mini;   // staging
    DevMatV _miniV;
};

// Gradient all-reduce over RCCL (new; the reference is single-GPU).  The gradient arena is cut
// into buckets in back-propagation order; a bucket is reduced on a side stream as soon as the
// last layer contributing to it has finished its bpropCommon, overlapped with the rest of
// backprop; the weight update waits for all buckets.
class GradComm {
public:
    ~GradComm();
    // uniqueId: the 128 bytes of ncclGetUniqueId() from rank 0.  Must be called on the worker thread.
    void init(int rank, int world, const std::string& uniqueId, int device);
    static std::string newUniqueId();
    bool active() const { return _world > 1 || _forced; }
    int world() const { return _world; }
    int rank() const { return _rank; }
    void allReduceAsync(float* p, size_t n, hipStream_t producer, hipStream_t alsoAfter = nullptr);   // sum, in place
    void waitAll(hipStream_t consumer);
    void forceSingleRank(bool f) { _forced = f; }   // exercise the RCCL path with world == 1 (tests)
    // Counters since the last report: all-reduce calls and bytes, steps (waitAll calls), and the time the compute stream
    // spent blocked on the side stream in waitAll (event pairs around the wait; the caller must have drained the stream).
    struct Stats { long calls = 0, steps = 0; double bytes = 0, waitMs = 0; };
    Stats takeStats();

private:
    void* _comm = nullptr;
    hipStream_t _stream = nullptr;
    std::vector<hipEvent_t> _events;
    size_t _nextEvent = 0;
    hipEvent_t _ready = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> _waitPairs;   // timed (before, after) the wait, one pair per step
    size_t _waitUsed = 0;
    Stats _stats;
    int _rank = 0, _world = 1;
    bool _forced = false, _pending = false;
};

class Worker;
class WorkResult {
public:
    enum RESULTS { BATCH_DONE, SYNC_DONE, ERROR };
    WorkResult(RESULTS t, Cost* c = nullptr, const std::string& msg = "") : type(t), cost(c), message(msg) {}
    ~WorkResult() { delete cost; }
    RESULTS type;
    Cost* cost;
    std::string message;
    int intValue = 0;
};

class ConvNet {
public:
    ConvNet(PyObject* layerParams, int minibatchSize, int deviceID);
    ~ConvNet();
    void start();   // spawns the device thread (convnet.cu:149-158)

    Queue<Worker*>& getWorkerQueue() { return _workerQueue; }
    Queue<WorkResult*>& getResultQueue() { return _resultQueue; }
    DataProvider& getDataProvider() { return *_dp; }
    Layer& operator[](int idx) { return *_layers[idx]; }
    Layer& getLayer(int idx) { return *_layers.at(idx); }
    int getNumLayers() const { return (int)_layers.size(); }
    GradComm& comm() { return _comm; }

    void copyToCPU();
    void copyToGPU();
    void updateWeights();
    void reset();
    void fprop(PASS_TYPE passType);
    void fprop(int miniIdx, PASS_TYPE passType);
    void fprop(DevMatV& data, PASS_TYPE passType);
    void bprop(PASS_TYPE passType);
    // cost handling: slots on the device, read once per batch
    int costSlotIndex() const;
    void beginBatchCosts();
    // while a step is being captured into a graph the costs go to a fixed slot (CostLayer::GRAPH_SLOT)
    bool capturing() const { return _capturing; }
    void warmUpShapes();           // dry run per minibatch shape of the current batch (sizes matrices, tunes tiles); see convnet.cpp
    void trainStep(int miniIdx);   // fprop + costs + bprop + update of one minibatch, eager or as a replayed graph
    long stepGraphReplays() const { return _graphReplays; }
    long warmedShapes() const { return _warmedShapes; }   // minibatch shapes dry-run by warmUpShapes since initModel
    long forkedSteps() const { return _forkedSteps; }   // eager training steps whose weight gradients ran on the second stream
    void endMinibatchCosts(int numCases);
    Cost* collectCosts();
    double getCostValue();   // synchronous, last minibatch only (grad check)
    int checkGradients();
    bool checkGradient(const std::string& name, float eps, Weights& weights);

    // plumbing for layers
    DevMat* currentData(int dataIdx) { return _data && dataIdx >= 0 && dataIdx < (int)_data->size() ? (*_data)[dataIdx] : nullptr; }   // this pass's input matrices
    void registerWeights(Weights* w) { _allWeights.push_back(w); }
    void gradientReady(Weights& w);           // called from WeightLayer::bpropCommon
    float* workspace(size_t bytes);
    // Weight-gradient branch of a training step: between beginWeightGradBranch() and endWeightGradBranch() the thread's
    // compute stream is a second stream that has waited for everything queued so far; joinWeightGradBranch() makes the
    // main stream wait for it (ConvNet::bprop, before the update).  See "One training step" in convnet.cpp.
    bool forksWeightGrads() const { return _forkThisStep; }
    void beginWeightGradBranch();
    void endWeightGradBranch();
    void joinWeightGradBranch();
    // the 1/N of the gradient scale: the cases of the GLOBAL minibatch (all ranks).  Ranks hold equal shares except in a ragged
    // tail, where the driver hands the true global count down (startBatch's third argument); see dp.shard_columns.
    int gradScaleCases(int localCases) const { return _globalMinibatchCases > 0 ? _globalMinibatchCases : localCases * _comm.world(); }
    void setGlobalMinibatchCases(int n) { _globalMinibatchCases = n; }
    unsigned long long nextRandOffset(size_t n);
    unsigned long long seed() const { return _seed; }
    int world() const { return _comm.world(); }
    void setSeed(unsigned long long s) { _seed = s; _randOffset = 0; }
    void releaseHost();
    int deviceID() const { return _deviceID; }
    // dropout mask injection for parity tests: name -> host mask consumed on the next fprop
    std::map<std::string, std::vector<float>> injectedMasks;
    std::map<std::string, DevMat*> debugActs();   // layer name -> output matrix

private:
    struct StepGraph {
        hipGraphExec_t exec = nullptr;
        int eagerRuns = 0;
        unsigned long long allocStamp = 0;   // DeviceAllocator::allocations() when captured
        bool warmed = false;   // warmUpShapes has run this shape
        bool bad = false;      // a capture failed: steps of this shape stay eager
        std::string why;
        enum Mode { UNDECIDED, GRAPH, EAGER };   // which way steps of this shape run once a graph exists (trainStep's trial)
        Mode mode = UNDECIDED;
        int trials = 0;
        float graphMs = 1e30f, eagerMs = 1e30f;
    };
    bool stepGraphsAllowed() const;
    void eagerStep(DevMatV& mb, int cases);
    bool captureStep(StepGraph& g, DevMatV& mb);
    void runEagerStep(StepGraph& g, DevMatV& mb, int cases, unsigned long long stamp);
    hipEvent_t _trialBegin = nullptr, _trialEnd = nullptr;
    std::map<int, StepGraph> _stepGraphs;   // by cases in the minibatch
    bool _capturing = false, _stochastic = false, _conserveMem = false;
    int _graphCases = 0;                    // cases of this batch that ran as replayed graphs
    long _graphReplays = 0;                 // steps replayed since initModel
    long _forkedSteps = 0, _warmedShapes = 0;

    void run();
    void initDevice();
    Layer* initLayer(const std::string& layerType, PyObject* paramsDict);
    void buildUpdateSpans();

    std::vector<Layer*> _layers;
    std::vector<DataLayer*> _dataLayers;
    std::vector<CostLayer*> _costs;
    std::vector<Weights*> _allWeights;
    DevMatV* _data = nullptr;
    DataProvider* _dp;
    int _deviceID;
    std::thread _thread;
    Queue<Worker*> _workerQueue;
    Queue<WorkResult*> _resultQueue;
    hipStream_t _stream = nullptr;
    // arenas
    float *_arenaW = nullptr, *_arenaInc = nullptr, *_arenaGrad = nullptr;
    size_t _arenaElems = 0;
    BucketPlanner _buckets;                // which slices of the gradient arena are reduced when (buckets.h)
    mcUpdateSpan* _spans = nullptr;
    int _numSpans = 0;
    float* _workspace = nullptr;
    size_t _workspaceBytes = 0;
    hipStream_t _branchStream = nullptr;   // the weight-gradient branch of a step
    hipEvent_t _forkEvent = nullptr, _joinEvent = nullptr;
    float* _branchWorkspace = nullptr;     // its own split-K workspace: both branches may hold slabs at once
    size_t _branchWorkspaceBytes = 0;
    bool _forkEager = false, _forkGraph = false;   // MC_STEP_FORK
    bool _forkThisStep = false, _onBranch = false, _branchOpen = false;
    bool _warming = false;
    GradComm _comm;
    int _slot = 0;
    std::vector<int> _slotCases;
    double _baseErr = 0;
    int _numTests = 0, _numFailures = 0;
    unsigned long long _seed, _randOffset = 0;
    int _globalMinibatchCases = 0;
};

// ------------------------------------------------------------------------------------------
// Work items (Kernel/src/convnet/worker.cu:61-214)
// ------------------------------------------------------------------------------------------
class Worker {
public:
    explicit Worker(ConvNet& net) : _convNet(&net) {}
    virtual ~Worker() {}
    virtual void run() = 0;
    // Python references held by the item; dropped by the Python thread after the result arrives.
    virtual void releaseHost() {}

protected:
    ConvNet* _convNet;
};

class DataWorker : public Worker {
public:
    DataWorker(ConvNet& net, HostMatV* data) : Worker(net), _data(data) {}
    ~DataWorker();
    void releaseHost() override;

protected:
    HostMatV* _data;
};

class TrainingWorker : public DataWorker {
public:
    // globalCases: cases of this batch over all ranks (0 = local cases x world size)
    TrainingWorker(ConvNet& net, HostMatV* data, bool test, long globalCases = 0) : DataWorker(net, data), _test(test), _globalCases(globalCases) {}
    void run() override;

private:
    bool _test;
    long _globalCases;
};

// Re-runs minibatches of the batch that is already on the device (left there by the previous
// startBatch): `count` steps starting at minibatch `first`, cycling.  Not in the reference; it
// lets a benchmark time training steps with the inputs resident in HBM.
class ResidentWorker : public Worker {
public:
    ResidentWorker(ConvNet& net, bool test, int first, int count) : Worker(net), _test(test), _first(first), _count(count) {}
    void run() override;

private:
    bool _test;
    int _first, _count;
};

class SyncWorker : public Worker {
public:
    explicit SyncWorker(ConvNet& net) : Worker(net) {}
    void run() override;
};

class GradCheckWorker : public DataWorker {
public:
    GradCheckWorker(ConvNet& net, HostMatV* data) : DataWorker(net, data) {}
    void run() override;
};

class MultiviewTestWorker : public DataWorker {
public:
    MultiviewTestWorker(ConvNet& net, HostMatV* data, int numViews, int logregIdx)
        : DataWorker(net, data), _numViews(numViews), _logregIdx(logregIdx) {}
    void run() override;

private:
    int _numViews, _logregIdx;
};

class FeatureWorker : public DataWorker {
public:
    FeatureWorker(ConvNet& net, HostMatV* data, HostMat* ftrs, int layerIdx)
        : DataWorker(net, data), _ftrs(ftrs), _layerIdx(layerIdx) {}
    void run() override;
    void releaseHost() override;

private:
    HostMat* _ftrs;
    int _layerIdx;
};

// Runs `fn` on the device thread and waits for it (used by the extra, non-reference entry
// points: data-parallel set-up, test hooks).
class CallWorker : public Worker {
public:
    CallWorker(ConvNet& net, std::function<void()> fn) : Worker(net), _fn(fn) {}
    void run() override;

private:
    std::function<void()> _fn;
};

}  // namespace rt
