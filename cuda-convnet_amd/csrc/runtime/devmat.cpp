#include "devmat.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace rt {

std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return std::string(buf);
}

static thread_local hipStream_t t_stream = nullptr;
hipStream_t computeStream() { return t_stream; }
void setComputeStream(hipStream_t s) { t_stream = s; }

// ---- per-launch timing ---------------------------------------------------------------------
namespace {
struct ProfRec { std::string key; hipEvent_t a, b; };
std::vector<ProfRec> g_prof;
std::string g_profFilter, g_profLayer, g_profPhase;
bool g_profOn = false;
}  // namespace

void profileSetFilter(const char* filter) {
    g_profFilter = filter ? filter : "";
    g_profOn = !g_profFilter.empty();
}

bool profileOn() { return g_profOn; }
bool profileEveryLaunch() { return g_profOn && g_profFilter == "*"; }

void profileSetLayer(const char* layerName) {
    if (g_profOn) g_profLayer = layerName ? layerName : "";
}

void profileSetPhase(const char* phase) {
    if (g_profOn) g_profPhase = phase ? phase : "";
}

ProfScope::ProfScope(const char* callText) {
    if (!g_profOn) return;
    const char* paren = strchr(callText, '(');
    std::string key = g_profLayer + (g_profPhase.empty() ? "" : "/" + g_profPhase) + ":" + std::string(callText, paren ? (size_t)(paren - callText) : strlen(callText));
    if (g_profFilter != "*" && key.find(g_profFilter) == std::string::npos) return;
    ProfRec r;
    r.key = key;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, computeStream());
    g_prof.push_back(r);
    _slot = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
    if (_slot >= 0) (void)hipEventRecord(g_prof[_slot].b, computeStream());
}

void profileReport(std::map<std::string, std::pair<long, double>>& out, bool reset) {
    (void)hipStreamSynchronize(computeStream());
    for (ProfRec& r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = out[r.key];
            e.first += 1;
            e.second += ms;
        }
        if (reset) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
    }
    if (reset) g_prof.clear();
}

DeviceAllocator& DeviceAllocator::get() {
    static DeviceAllocator a;
    return a;
}

// Blocks are rounded up to 64 KiB below 8 MiB and to 2 MiB above, so the few distinct tensor
// sizes of a net map onto few distinct block sizes and tail minibatches reuse the full-size blocks.
static size_t round_capacity(size_t elems) {
    const size_t bytes = elems * sizeof(float);
    const size_t q = bytes < (8u << 20) ? (64u << 10) : (2u << 20);
    return ((bytes + q - 1) / q * q) / sizeof(float);
}

float* DeviceAllocator::alloc(size_t elems, size_t* capacityOut) {
    _allocations++;
    const size_t cap = round_capacity(elems ? elems : 1);
    auto it = _free.lower_bound(cap);
    // reuse a cached block unless it would waste more than half of itself
    if (it != _free.end() && it->first <= 2 * cap) {
        float* p = it->second;
        *capacityOut = it->first;
        _cached -= it->first * sizeof(float);
        _inUse += it->first * sizeof(float);
        _free.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap * sizeof(float));
    if (e != hipSuccess) {
        trim();
        e = hipMalloc(&p, cap * sizeof(float));
    }
    if (e != hipSuccess) throw std::runtime_error(fmt("hipMalloc of %zu bytes failed: %s", cap * sizeof(float), hipGetErrorString(e)));
    *capacityOut = cap;
    _inUse += cap * sizeof(float);
    return static_cast<float*>(p);
}

void DeviceAllocator::release(float* p, size_t capacity) {
    if (!p) return;
    _free.insert({capacity, p});
    _inUse -= capacity * sizeof(float);
    _cached += capacity * sizeof(float);
}

void DeviceAllocator::trim() {
    (void)hipDeviceSynchronize();
    for (auto& kv : _free) (void)hipFree(kv.second);
    _free.clear();
    _cached = 0;
}

void DevMat::free() {
    if (_own && _p) DeviceAllocator::get().release(_p, _cap);
    _p = nullptr;
    _cap = 0;
    _rows = _cols = 0;
    _own = true;
}

void DevMat::resize(int rows, int cols) {
    if (rows < 0 || cols < 0) throw std::runtime_error(fmt("DevMat::resize(%d, %d)", rows, cols));
    const size_t need = (size_t)rows * cols;
    if (!_own) {
        if (need > _cap) throw std::runtime_error("DevMat::resize: a view cannot grow");
    } else if (need > _cap) {
        // Blocks are recycled without synchronisation: every user of this memory is ordered on
        // the one compute stream, so a later owner's kernels run after the earlier owner's.
        if (_p) DeviceAllocator::get().release(_p, _cap);
        _p = DeviceAllocator::get().alloc(need, &_cap);
    }
    _rows = rows;
    _cols = cols;
}

void DevMat::reshape(int rows, int cols) {
    if ((size_t)rows * cols != numel()) throw std::runtime_error(fmt("DevMat::reshape: %dx%d -> %dx%d changes the element count", _rows, _cols, rows, cols));
    _rows = rows;
    _cols = cols;
}

void DevMat::view(float* p, int rows, int cols) {
    free();
    _own = false;
    _p = p;
    _cap = (size_t)rows * cols;
    _rows = rows;
    _cols = cols;
}

void DevMat::copyFromHost(const float* h, int rows, int cols) {
    resize(rows, cols);
    if (numel()) RT_HIP(hipMemcpyAsync(_p, h, numel() * sizeof(float), hipMemcpyHostToDevice, computeStream()));
}

void DevMat::copyToHost(float* h) const {
    if (!numel()) return;
    RT_HIP(hipMemcpyAsync(h, _p, numel() * sizeof(float), hipMemcpyDeviceToHost, computeStream()));
    RT_HIP(hipStreamSynchronize(computeStream()));
}

void DevMat::copyFrom(const DevMat& src) {
    resize(src.rows(), src.cols());
    if (numel()) RT_HIP(hipMemcpyAsync(_p, src.data(), numel() * sizeof(float), hipMemcpyDeviceToDevice, computeStream()));
}

void DevMat::zero() {
    if (numel()) RT_HIP(hipMemsetAsync(_p, 0, numel() * sizeof(float), computeStream()));
}

}  // namespace rt
