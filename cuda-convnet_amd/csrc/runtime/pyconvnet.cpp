// pyconvnet.cpp -- the `libkernel` CPython extension: the seven entry points of the reference's
// Kernel/src/convnet/pyconvnet.cu:49-209 with the same names, argument formats and return
// values, plus a few additions (marked NEW) for data parallelism, benchmarking and tests.
//
//   initModel(list layerDicts, int minibatchSize, int deviceID) -> 0
//   startBatch(list ndarrays[, int test=0]) -> 0              (asynchronous)
//   finishBatch() -> ({costName: [values]}, numCases)          (blocks)
//   checkGradients(list ndarrays) -> 0                         (blocks; prints the report)
//   startMultiviewTest(list ndarrays, int numViews, int logregIdx) -> 0
//   startFeatureWriter(list ndarrays + [ftrs], int layerIdx) -> 0
//   syncWithHost() -> 0                                        (blocks)
//
// Error behaviour: argument errors are TypeErrors as before; runtime faults, which the reference
// turns into assert()/exit(1) (convnet.cu:79-82, nvmatrix.cuh:54-59), are raised as RuntimeError
// from the blocking call that observes them.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <dlfcn.h>
#include <libgen.h>

#include <deque>
#include <iostream>

#include "convnet.h"

using namespace rt;

static ConvNet* model = nullptr;
static std::deque<Worker*> pending;   // work items whose Python references are still held

#define REQUIRE_MODEL()                                                              \
    if (!model) {                                                                    \
        PyErr_SetString(PyExc_RuntimeError, "initModel has not been called");        \
        return nullptr;                                                              \
    }

// Synchronous entry points submit one item and wait for ITS result; the result queue is first-in first-out, so with a
// startBatch / startResident still outstanding they would swallow that batch's result instead.
#define REQUIRE_IDLE(what)                                                                                  \
    if (!pending.empty()) {                                                                                 \
        PyErr_SetString(PyExc_RuntimeError, what ": a started batch is outstanding (call finishBatch first)"); \
        return nullptr;                                                                                     \
    }

static PyObject* fail(const std::exception& e) {
    PyErr_SetString(PyExc_RuntimeError, e.what());
    return nullptr;
}

static void submit(Worker* w) {
    pending.push_back(w);
    model->getWorkerQueue().enqueue(w);
}

// Waits for the oldest item's result without holding the GIL, then drops that item's references.
static WorkResult* await_result() {
    WorkResult* res;
    Py_BEGIN_ALLOW_THREADS
    res = model->getResultQueue().dequeue();
    Py_END_ALLOW_THREADS
    if (!pending.empty()) {
        Worker* w = pending.front();
        pending.pop_front();
        w->releaseHost();
        delete w;
    }
    return res;
}

static PyObject* check(WorkResult* res, WorkResult::RESULTS want) {
    if (res->type == WorkResult::ERROR) {
        PyErr_SetString(PyExc_RuntimeError, res->message.c_str());
        delete res;
        return nullptr;
    }
    if (res->type != want) {
        PyErr_SetString(PyExc_RuntimeError, "unexpected result type (is a startBatch without finishBatch outstanding?)");
        delete res;
        return nullptr;
    }
    return Py_None;   // borrowed marker: "ok"
}

static PyObject* initModel(PyObject*, PyObject* args) {
    PyObject* layers;
    int minibatch, device;
    if (!PyArg_ParseTuple(args, "O!ii", &PyList_Type, &layers, &minibatch, &device)) return nullptr;
    if (model) {
        PyErr_SetString(PyExc_RuntimeError, "one model per process (initModel was already called)");
        return nullptr;
    }
    if (minibatch <= 0) {
        PyErr_SetString(PyExc_ValueError, "minibatch size must be positive");
        return nullptr;
    }
    try {
        model = new ConvNet(layers, minibatch, device);
        model->start();
    } catch (std::exception& e) {
        std::cout << "Error creating ConvNet: " << e.what() << std::endl;
        model = nullptr;
        return fail(e);
    }
    return Py_BuildValue("i", 0);
}

static PyObject* startBatch(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    PyObject* data;
    int test = 0;
    long globalCases = 0;   // data parallel: cases of this batch over all ranks (optional; 0 = local cases x world size)
    if (!PyArg_ParseTuple(args, "O!|il", &PyList_Type, &data, &test, &globalCases)) return nullptr;
    try {
        submit(new TrainingWorker(*model, getMatrixV(data), test != 0, globalCases));
    } catch (std::exception& e) { return fail(e); }
    return Py_BuildValue("i", 0);
}

static PyObject* startMultiviewTest(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    PyObject* data;
    int numViews, logregIdx;
    if (!PyArg_ParseTuple(args, "O!ii", &PyList_Type, &data, &numViews, &logregIdx)) return nullptr;
    if (numViews <= 0 || logregIdx < 0 || logregIdx >= model->getNumLayers()) {
        PyErr_SetString(PyExc_ValueError, "bad numViews / logregIdx");
        return nullptr;
    }
    try {
        submit(new MultiviewTestWorker(*model, getMatrixV(data), numViews, logregIdx));
    } catch (std::exception& e) { return fail(e); }
    return Py_BuildValue("i", 0);
}

static PyObject* startFeatureWriter(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    PyObject* data;
    int layerIdx;
    if (!PyArg_ParseTuple(args, "O!i", &PyList_Type, &data, &layerIdx)) return nullptr;
    if (layerIdx < 0 || layerIdx >= model->getNumLayers() || PyList_GET_SIZE(data) < 2) {
        PyErr_SetString(PyExc_ValueError, "bad layerIdx / data list");
        return nullptr;
    }
    try {
        HostMatV* mvec = getMatrixV(data);
        HostMat* ftrs = mvec->back();
        mvec->pop_back();
        submit(new FeatureWorker(*model, mvec, ftrs, layerIdx));
    } catch (std::exception& e) { return fail(e); }
    return Py_BuildValue("i", 0);
}

static PyObject* finishBatch(PyObject*, PyObject*) {
    REQUIRE_MODEL();
    if (pending.empty()) {
        PyErr_SetString(PyExc_RuntimeError, "finishBatch without a started batch");
        return nullptr;
    }
    WorkResult* res = await_result();
    if (!check(res, WorkResult::BATCH_DONE)) return nullptr;
    PyObject* dict = PyDict_New();
    for (auto& kv : res->cost->getCostMap()) {
        PyObject* v = PyList_New(0);
        for (double d : kv.second) {
            PyObject* f = PyFloat_FromDouble(d);
            PyList_Append(v, f);
            Py_DECREF(f);
        }
        PyDict_SetItemString(dict, kv.first.c_str(), v);
        Py_DECREF(v);
    }
    PyObject* ret = Py_BuildValue("Ni", dict, res->cost->getNumCases());
    delete res;
    return ret;
}

static PyObject* checkGradients(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    PyObject* data;
    if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &data)) return nullptr;
    REQUIRE_IDLE("checkGradients");
    try {
        submit(new GradCheckWorker(*model, getMatrixV(data)));
    } catch (std::exception& e) { return fail(e); }
    WorkResult* res = await_result();
    if (!check(res, WorkResult::BATCH_DONE)) return nullptr;
    const int failures = res->intValue;
    delete res;
    // The reference's worker calls exit(0) here (worker.cu:126-130); the driver does that now,
    // and a caller that wants the verdict gets the number of failed tensors.
    return Py_BuildValue("i", failures);
}

static PyObject* syncWithHost(PyObject*, PyObject*) {
    REQUIRE_MODEL();
    REQUIRE_IDLE("syncWithHost");
    submit(new SyncWorker(*model));
    WorkResult* res = await_result();
    if (!check(res, WorkResult::SYNC_DONE)) return nullptr;
    delete res;
    return Py_BuildValue("i", 0);
}

// ---- NEW entry points ---------------------------------------------------------------------
static PyObject* run_on_device(std::function<void()> fn) {
    REQUIRE_IDLE("this call");
    submit(new CallWorker(*model, fn));
    WorkResult* res = await_result();
    if (!check(res, WorkResult::SYNC_DONE)) return nullptr;
    delete res;
    return Py_BuildValue("i", 0);
}

// startResident(test, firstMinibatch, count): re-run minibatches of the batch already on the device
static PyObject* startResident(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    int test, first, count;
    if (!PyArg_ParseTuple(args, "iii", &test, &first, &count)) return nullptr;
    if (first < 0 || count <= 0 || count > CostLayer::MAX_SLOTS) {
        PyErr_SetString(PyExc_ValueError, "bad firstMinibatch / count");
        return nullptr;
    }
    submit(new ResidentWorker(*model, test != 0, first, count));
    return Py_BuildValue("i", 0);
}

// commUniqueId() -> bytes: rank 0 creates the RCCL id that every rank passes to initComm
static PyObject* commUniqueId(PyObject*, PyObject*) {
    try {
        const std::string id = GradComm::newUniqueId();
        return PyBytes_FromStringAndSize(id.data(), id.size());
    } catch (std::exception& e) { return fail(e); }
}

// initComm(rank, worldSize, idBytes[, force=0]): join the data-parallel group (collective call)
static PyObject* initComm(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    int rank, world, force = 0;
    const char* id;
    Py_ssize_t idLen;
    if (!PyArg_ParseTuple(args, "iiy#|i", &rank, &world, &id, &idLen, &force)) return nullptr;
    if (world < 1 || rank < 0 || rank >= world) {
        PyErr_SetString(PyExc_ValueError, "bad rank / world size");
        return nullptr;
    }
    const std::string ids(id, idLen);
    return run_on_device([=] {
        model->comm().init(rank, world, ids, model->deviceID());
        model->comm().forceSingleRank(force != 0);
    });
}

// setDropoutMasks(list of float32 ndarrays): uniforms consumed, in fprop order, instead of the RNG
static PyObject* setDropoutMasks(PyObject*, PyObject* args) {
    PyObject* list;
    if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &list)) return nullptr;
    try {
        HostMatV* v = getMatrixV(list);
        for (HostMat* m : *v) {
            std::vector<float> rm(m->numel());
            m->toRowMajor(rm.data());
            injectUniform(rm.data(), rm.size());
            m->release();
            delete m;
        }
        delete v;
    } catch (std::exception& e) { return fail(e); }
    return Py_BuildValue("i", 0);
}

static PyObject* setSeed(PyObject*, PyObject* args) {
    unsigned long long seed;
    if (!PyArg_ParseTuple(args, "K", &seed)) return nullptr;
    setRandSeed(seed);
    return Py_BuildValue("i", 0);
}

// getActs(layerName, grad=0) -> float32 ndarray (outputs, numCases): a layer's current output
// (or activity gradient); for tests and debugging
static PyObject* getActs(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    const char* name;
    int grad = 0;
    if (!PyArg_ParseTuple(args, "s|i", &name, &grad)) return nullptr;
    std::vector<float>* host = new std::vector<float>();
    int rows = 0, cols = 0;
    const std::string lname(name);
    PyObject* ok = run_on_device([&, lname, grad] {
        for (int i = 0; i < model->getNumLayers(); i++) {
            Layer& l = model->getLayer(i);
            if (l.getName() != lname) continue;
            DevMat& m = grad ? l.getActsGrad() : l.getActs();
            rows = m.rows();
            cols = m.cols();
            host->resize(m.numel());
            m.copyToHost(host->data());
            return;
        }
        throw std::runtime_error("no layer named '" + lname + "'");
    });
    if (!ok) {
        delete host;
        return nullptr;
    }
    Py_DECREF(ok);
    PyObject* np = PyImport_ImportModule("numpy");
    PyObject* bytes = PyBytes_FromStringAndSize(reinterpret_cast<const char*>(host->data()), host->size() * sizeof(float));
    delete host;
    PyObject* arr = PyObject_CallMethod(np, "frombuffer", "Os", bytes, "float32");
    Py_DECREF(bytes);
    PyObject* shaped = arr ? PyObject_CallMethod(arr, "reshape", "ii", rows, cols) : nullptr;
    Py_XDECREF(arr);
    PyObject* copy = shaped ? PyObject_CallMethod(shaped, "copy", nullptr) : nullptr;
    Py_XDECREF(shaped);
    Py_DECREF(np);
    return copy;
}

// setProfile(filter): time every kernel launch whose "<layer>:<mc_function>" key contains `filter`
// ("*" = all, "" = off) with HIP events on the compute stream
static PyObject* setProfile(PyObject*, PyObject* args) {
    REQUIRE_MODEL();
    const char* filter;
    if (!PyArg_ParseTuple(args, "s", &filter)) return nullptr;
    const std::string f(filter);
    return run_on_device([=] { profileSetFilter(f.c_str()); });
}

// getProfile() -> {key: (launches, total_ms)}; clears the records.  With a data-parallel communicator it also carries the
// "net/step_graph:replays" -> (training steps replayed as captured graphs since initModel, 0) and the
// counters since the last call: "dp/allreduce:bytes" -> (all-reduce calls, bytes reduced), "dp/wait:ms" -> (steps,
// milliseconds the compute stream was blocked waiting for the reductions, i.e. what back-propagation did not hide).
static PyObject* getProfile(PyObject*, PyObject*) {
    REQUIRE_MODEL();
    std::map<std::string, std::pair<long, double>> rep;
    PyObject* ok = run_on_device([&] {
        profileReport(rep, true);
        rep["net/step_graph:replays"] = std::make_pair(model->stepGraphReplays(), 0.0);   // training steps replayed as HIP graphs so far
        rep["net/step_warm:shapes"] = std::make_pair(model->warmedShapes(), 0.0);         // minibatch shapes dry-run before their first step
        rep["net/step_fork:steps"] = std::make_pair(model->forkedSteps(), 0.0);           // eager steps that ran their weight gradients on the second stream
        if (model->comm().active()) {
            const GradComm::Stats st = model->comm().takeStats();
            rep["dp/allreduce:bytes"] = std::make_pair(st.calls, st.bytes);
            rep["dp/wait:ms"] = std::make_pair(st.steps, st.waitMs);
        }
    });
    if (!ok) return nullptr;
    Py_DECREF(ok);
    PyObject* dict = PyDict_New();
    for (auto& kv : rep) {
        PyObject* v = Py_BuildValue("(ld)", kv.second.first, kv.second.second);
        PyDict_SetItemString(dict, kv.first.c_str(), v);
        Py_DECREF(v);
    }
    return dict;
}

static PyMethodDef methods[] = {{"initModel", initModel, METH_VARARGS, nullptr},
                                {"startBatch", startBatch, METH_VARARGS, nullptr},
                                {"finishBatch", finishBatch, METH_VARARGS, nullptr},
                                {"checkGradients", checkGradients, METH_VARARGS, nullptr},
                                {"startMultiviewTest", startMultiviewTest, METH_VARARGS, nullptr},
                                {"startFeatureWriter", startFeatureWriter, METH_VARARGS, nullptr},
                                {"syncWithHost", syncWithHost, METH_VARARGS, nullptr},
                                {"startResident", startResident, METH_VARARGS, nullptr},
                                {"commUniqueId", commUniqueId, METH_NOARGS, nullptr},
                                {"initComm", initComm, METH_VARARGS, nullptr},
                                {"setDropoutMasks", setDropoutMasks, METH_VARARGS, nullptr},
                                {"setSeed", setSeed, METH_VARARGS, nullptr},
                                {"getActs", getActs, METH_VARARGS, nullptr},
                                {"setProfile", setProfile, METH_VARARGS, nullptr},
                                {"getProfile", getProfile, METH_NOARGS, nullptr},
                                {nullptr, nullptr, 0, nullptr}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "libkernel", "MI355X conv-net training runtime (cuda-convnet extension API)", -1, methods};

PyMODINIT_FUNC PyInit_libkernel(void) {
    PyObject* m = PyModule_Create(&moduledef);
    if (!m) return nullptr;
    initNumpy();
    if (PyErr_Occurred()) return nullptr;
    // plugins live next to this shared object (dist layout: libkernel.so + Plugins/*.so)
    Dl_info info;
    std::string dir;
    if (dladdr(reinterpret_cast<void*>(&PyInit_libkernel), &info) && info.dli_fname) {
        std::string path(info.dli_fname);
        const size_t slash = path.rfind('/');
        dir = slash == std::string::npos ? "." : path.substr(0, slash);
    }
    loadPlugins(dir.c_str());
    return m;
}
