// devmat.h -- device memory for the host runtime: a caching allocator and DevMat, the
// replacement for the reference's NVMatrix (Kernel/include/nvmatrix/nvmatrix.cuh:61-457) on the
// training path.
//
// What is kept from NVMatrix: an owning or non-owning fp32 2-D buffer whose resize() preserves
// the allocation when the element count fits (nvmatrix.cu:458-491), host<->device copies, views.
// What is dropped on purpose: the `isTrans` soft transpose.  Every activation here is stored
// (features, numCases) row-major -- the bytes the reference also has, whichever way it labels
// them (nvmatrix.cu:540-553) -- so layers pass explicit GEMM operand flags instead of flipping a
// flag on shared matrices.  All math goes through the C-ABI of libmckernels.so.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mc_kernels.h"

namespace rt {

std::string fmt(const char* f, ...);

#define RT_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            throw std::runtime_error(::rt::fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__)); \
    } while (0)

// Optional per-launch timing (bench.py's roofline leg): when a filter is set, every RT_MC call
// whose "<layer>:<function>" key contains the filter is bracketed by two HIP events on the
// compute stream; profileReport() turns them into (launches, total ms) per key.
class ProfScope {
public:
    explicit ProfScope(const char* callText);
    ~ProfScope();
private:
    int _slot = -1;
};
void profileSetFilter(const char* filter);            // NULL / "" switches profiling off
bool profileOn();
bool profileEveryLaunch();                            // the filter is "*": launches are timed one by one, steps stay on one stream
void profileSetLayer(const char* layerName);
void profileSetPhase(const char* phase);              // "fwd" / "wgrad" / "bwd": part of the key
void profileReport(std::map<std::string, std::pair<long, double>>& out, bool reset);

// Every kernel call of the runtime goes through this: a rejected call is a bug in the caller's
// shapes, reported with the library's own message.
#define RT_MC(call)                                                                                \
    do {                                                                                           \
        ::rt::ProfScope prof__(#call);                                                             \
        int rc__ = (call);                                                                         \
        if (rc__ != MC_OK)                                                                         \
            throw std::runtime_error(::rt::fmt("%s -> %d: %s (%s:%d)", #call, rc__, mc_last_error(), __FILE__, __LINE__)); \
    } while (0)

// The stream all of this thread's kernels and copies are issued on (set by the worker thread).
hipStream_t computeStream();
void setComputeStream(hipStream_t s);

// Size-bucketed caching allocator: the reference pays one cublasAlloc/cublasFree per
// NVMatrix::resize (nvmatrix.cu:462-491); here freed blocks are kept and reused, so a steady
// training loop performs no hipMalloc at all after the first minibatch.
class DeviceAllocator {
public:
    static DeviceAllocator& get();
    float* alloc(size_t elems, size_t* capacityOut);
    void release(float* p, size_t capacity);
    void trim();  // hipFree everything cached
    // how many blocks have been handed out so far: a matrix that (re)allocates changes address, which invalidates
    // anything that baked addresses in (ConvNet's captured training steps compare this stamp before every replay)
    unsigned long long allocations() const { return _allocations; }
    void noteForeignAllocation() { _allocations++; }   // buffers the net allocates itself
    size_t bytesInUse() const { return _inUse; }
    size_t bytesCached() const { return _cached; }

private:
    std::multimap<size_t, float*> _free;  // capacity (elements) -> block
    size_t _inUse = 0, _cached = 0;
    unsigned long long _allocations = 0;
};

class DevMat {
public:
    DevMat() {}
    DevMat(int rows, int cols) { resize(rows, cols); }
    ~DevMat() { free(); }
    DevMat(const DevMat&) = delete;
    DevMat& operator=(const DevMat&) = delete;

    // (rows, cols) row-major.  Keeps the allocation when it is large enough; contents are
    // preserved only when the allocation is kept.
    void resize(int rows, int cols);
    void resize(const DevMat& like) { resize(like._rows, like._cols); }
    // Same bytes, new shape (element count must match).
    void reshape(int rows, int cols);
    // Non-owning window onto someone else's memory.
    void view(float* p, int rows, int cols);
    void free();  // NVMatrix::truncate (nvmatrix.cu:493-495)

    float* data() const { return _p; }
    int rows() const { return _rows; }
    int cols() const { return _cols; }
    size_t numel() const { return (size_t)_rows * _cols; }
    bool empty() const { return numel() == 0; }
    bool sameDims(const DevMat& o) const { return _rows == o._rows && _cols == o._cols; }

    void copyFromHost(const float* h, int rows, int cols);  // async on computeStream()
    void copyToHost(float* h) const;                        // synchronous
    void copyFrom(const DevMat& src);                       // async D2D, resizes
    void zero();

private:
    float* _p = nullptr;
    size_t _cap = 0;
    int _rows = 0, _cols = 0;
    bool _own = true;
};

typedef std::vector<DevMat*> DevMatV;

}  // namespace rt
