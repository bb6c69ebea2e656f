// api.hip -- extern "C" entry points for the contraction kernels (conv / local / dense),
// error reporting, device query and the split-K reduction.  The unshared (local) layers run on the same
// fp32-MFMA engine as batched products (igemm.hip: one product per module forward, one per pixel backward).
#include <stdlib.h>
#include <string.h>
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include "mc_common.h"

namespace mc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int num_cus() {
    static thread_local int dev_cached = -1;
    static thread_local int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cus;
    if (dev != dev_cached) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        dev_cached = dev;
    }
    return cus;
}

// implemented in igemm.hip
int conv_filter_acts_impl(const float*, const float*, float*, const float*, int, int, int, int, int, int, int, int, int, int,
                          int, int, int, float, float, float*, int, unsigned*, hipStream_t);
int conv_img_acts_impl(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, float,
                       float, float*, int, hipStream_t);
int conv_split_plan(int, long, int, int, int*);
int conv_band_kind(int, int, int, int, int, int, int, int, int, int, int);
int conv_band_launch(const float*, const float*, float*, const float*, int, int, int, int, int, int, int, int, int, int, float, unsigned*,
                     hipStream_t);
void describe_write(const char*);
int conv_weight_acts_impl(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, int,
                          float, float, float*, long, float, hipStream_t);
int conv_weight_grad_plan(int, int, int, int, int, int, int*);
void set_tune_inhibit(bool);
bool tune_inhibited();
void set_describe(char*, int);
bool describing();
int gemm_impl(const float*, const float*, float*, int, int, int, int, int, int, int, int, float, float, const float*, int,
              const float*, int, int, long, unsigned*, const unsigned*, float*, long, float, hipStream_t);

}  // namespace mc
// (elementwise.hip) two column sums over the same number of rows in one launch; internal, not part of the C-ABI contract
extern "C" int mc__sum_cols_pair(const float* v1, float* out1, int cols1, float sT1, float sO1, const float* v2, float* out2, int cols2,
                                 float sO2, int rows, int tallMin, hipStream_t st);
namespace mc {
namespace {

constexpr int TPB = 256;

// out[i][j] = beta*C + alpha*sum_q slab[q][i][j] (+bias[i]) (relu)
__global__ void k_splitk_reduce(const float* __restrict__ slabs, float* C, int m, int n, int ldc, int chunks, float alpha,
                                float beta, const float* __restrict__ bias, int act, const float* __restrict__ mask, int ldmask) {
    const size_t total = (size_t)m * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
        float acc = 0.f;
        for (int q = 0; q < chunks; ++q) acc += slabs[(size_t)q * total + e];
        float v = alpha * acc;
        const size_t idx = (size_t)i * ldc + j;
        if (beta != 0.f) v += beta * C[idx];
        if (bias) v += bias[i];
        if (act == MC_NEURON_RELU) v = v < 0.f ? 0.f : v;
        if (mask && !(mask[(size_t)i * ldmask + j] > 0.f)) v = 0.f;
        C[idx] = v;
    }
}

// Same reduction for many slabs: workgroup = 16 float4 columns x 16 slab groups; thread (cg, qg) adds
// slabs qg, qg+16, ... in order, then the 16 partials are added in a fixed LDS tree (deterministic).
// Requires n % 4 == 0, ldc % 4 == 0 and 16-byte aligned pointers.
__global__ void __launch_bounds__(256) k_splitk_reduce_tall(const float4* __restrict__ slabs, float* C, int m, int n4, int ldc, int chunks,
                                                              float alpha, float beta, const float* __restrict__ bias, int act,
                                                              const float* __restrict__ mask, int ldmask) {
    __shared__ float4 sh[16][17];
    const int cg = threadIdx.x & 15, qg = threadIdx.x >> 4;
    const size_t total4 = (size_t)m * n4;
    const size_t e = (size_t)blockIdx.x * 16 + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total4) {
        int q = qg;   // sixteen slabs' loads in flight, added in slab order (same bits as one at a time)
        for (; q + 15 * 16 < chunks; q += 16 * 16) {
            float4 x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = slabs[(size_t)(q + 16 * u) * total4 + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        for (; q < chunks; q += 16) {
            const float4 x = slabs[(size_t)q * total4 + e];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
    }
    sh[qg][cg] = acc;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (qg < half) {
            const float4 a = sh[qg][cg], b = sh[qg + half][cg];
            sh[qg][cg] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (qg == 0 && e < total4) {
        const float4 t = sh[0][cg];
        const int i = (int)(e / n4), j4 = (int)(e - (size_t)i * n4);
        float4* dst = reinterpret_cast<float4*>(C + (size_t)i * ldc) + j4;
        float4 v = make_float4(alpha * t.x, alpha * t.y, alpha * t.z, alpha * t.w);
        if (beta != 0.f) {
            const float4 o = *dst;
            v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
        }
        if (bias) { const float b = bias[i]; v.x += b; v.y += b; v.z += b; v.w += b; }
        if (act == MC_NEURON_RELU) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
        if (mask) {
            const float* mk = mask + (size_t)i * ldmask + 4 * j4;
            if (!(mk[0] > 0.f)) v.x = 0.f;
            if (!(mk[1] > 0.f)) v.y = 0.f;
            if (!(mk[2] > 0.f)) v.z = 0.f;
            if (!(mk[3] > 0.f)) v.w = 0.f;
        }
        *dst = v;
    }
}

// float4 form of k_splitk_reduce for a few slabs (n % 4 == 0, ldc % 4 == 0, 16-byte aligned pointers)
__global__ void k_splitk_reduce4(const float4* __restrict__ slabs, float* C, int m, int n4, int ldc, int chunks, float alpha, float beta,
                                 const float* __restrict__ bias, int act) {
    const size_t total4 = (size_t)m * n4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        float4 acc = slabs[e];
        for (int q = 1; q < chunks; ++q) {
            const float4 x = slabs[(size_t)q * total4 + e];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        const int i = (int)(e / n4), j4 = (int)(e - (size_t)i * n4);
        float4* dst = reinterpret_cast<float4*>(C + (size_t)i * ldc) + j4;
        float4 v = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
        if (beta != 0.f) {
            const float4 o = *dst;
            v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
        }
        if (bias) { const float b = bias[i]; v.x += b; v.y += b; v.z += b; v.w += b; }
        if (act == MC_NEURON_RELU) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
        *dst = v;
    }
}

// ---- relu bit masks (layout: mc_kernels.h "relu bit masks"): word (i >> 5) * ld + j, bit p = 16 * h + r <-> row
// (i & ~31) + (r & 3) + 8 * (r >> 2) + 4 * h.  Thread = (32-row block, column): walks the block's rows in bit order.
__device__ __forceinline__ int bit_row(int p) { const int h = p >> 4, r = p & 15; return (r & 3) + 8 * (r >> 2) + 4 * h; }

// split-K reduction that also writes the bit mask of its (relu) output: out = alpha * sum_q slab[q] (+bias) (relu), bit = out > 0
__global__ void k_splitk_reduce_bits(const float* __restrict__ slabs, float* C, unsigned* bits, int m, int n, int chunks, float alpha,
                                     const float* __restrict__ bias, int act) {
    const size_t total = (size_t)m * n;
    const int blocks32 = (m + 31) >> 5;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (size_t)blocks32 * n; t += (size_t)gridDim.x * blockDim.x) {
        const int rb = (int)(t / n), j = (int)(t - (size_t)rb * n);
        unsigned w = 0;
        for (int p = 0; p < 32; ++p) {
            const int i = rb * 32 + bit_row(p);
            if (i >= m) continue;
            const size_t e = (size_t)i * n + j;
            float acc = 0.f;
            for (int q = 0; q < chunks; ++q) acc += slabs[(size_t)q * total + e];
            float v = alpha * acc;
            if (bias) v += bias[i];
            if (act == MC_NEURON_RELU) v = v < 0.f ? 0.f : v;
            C[e] = v;
            w |= (v > 0.f) ? (1u << p) : 0u;
        }
        bits[t] = w;
    }
}

// float4 form (n % 4 == 0, 16-byte aligned): a workgroup takes 32 rows x 128 columns; thread (prow, c4) sums rows prow, prow + 8, ...
// (bit order) of column group c4 -- 512 contiguous bytes per row and wave -- and the eight threads of a column group OR their
// partial words together in LDS.
__global__ void __launch_bounds__(256) k_splitk_reduce_bits4(const float4* __restrict__ slabs, float* C, unsigned* bits, int m, int n, int chunks,
                                                             float alpha, const float* __restrict__ bias, int act) {
    __shared__ unsigned sh[128];
    const int c4 = threadIdx.x & 31, prow = threadIdx.x >> 5;
    const int colTiles = (n + 127) >> 7, blocks32 = (m + 31) >> 5, n4 = n >> 2;
    const size_t total4 = (size_t)m * n4;
    for (int tile = blockIdx.x; tile < blocks32 * colTiles; tile += gridDim.x) {
        const int rb = tile / colTiles, ct = tile - rb * colTiles, j = ct * 128 + c4 * 4;
        if (threadIdx.x < 128) sh[threadIdx.x] = 0;
        __syncthreads();
        unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (j < n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = prow + 8 * k, i = rb * 32 + bit_row(p);
                if (i < m) {
                    const size_t e = (size_t)i * n4 + (j >> 2);
                    float4 acc = slabs[e];
                    for (int q = 1; q < chunks; ++q) {
                        const float4 x = slabs[(size_t)q * total4 + e];
                        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                    }
                    float4 v = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
                    if (bias) { const float b = bias[i]; v.x += b; v.y += b; v.z += b; v.w += b; }
                    if (act == MC_NEURON_RELU) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                    reinterpret_cast<float4*>(C)[e] = v;
                    w0 |= (v.x > 0.f) ? (1u << p) : 0u; w1 |= (v.y > 0.f) ? (1u << p) : 0u;
                    w2 |= (v.z > 0.f) ? (1u << p) : 0u; w3 |= (v.w > 0.f) ? (1u << p) : 0u;
                }
            }
            atomicOr(&sh[c4 * 4 + 0], w0); atomicOr(&sh[c4 * 4 + 1], w1); atomicOr(&sh[c4 * 4 + 2], w2); atomicOr(&sh[c4 * 4 + 3], w3);
        }
        __syncthreads();
        if (threadIdx.x < 128 && ct * 128 + (int)threadIdx.x < n) bits[(size_t)rb * n + ct * 128 + threadIdx.x] = sh[threadIdx.x];
        __syncthreads();
    }
}

__global__ void k_relu_bits_from_acts(const float* __restrict__ acts, unsigned* bits, int m, int n, int ld) {
    const int blocks32 = (m + 31) >> 5;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (size_t)blocks32 * n; t += (size_t)gridDim.x * blockDim.x) {
        const int rb = (int)(t / n), j = (int)(t - (size_t)rb * n);
        unsigned w = 0;
        for (int p = 0; p < 32; ++p) {
            const int i = rb * 32 + bit_row(p);
            if (i < m && acts[(size_t)i * ld + j] > 0.f) w |= 1u << p;
        }
        bits[(size_t)rb * ld + j] = w;
    }
}

// Library-owned scratch for the split-K slabs of the conv forward / backward-data entry points, whose signatures (the
// reference's host wrappers) carry no workspace.  One grow-only buffer per (device, stream): calls on one stream are ordered,
// so they may share it; calls on different streams never do.  A buffer that has become too small is RETIRED, not freed
// (launches already queued may still be using it): nothing here synchronises or frees, and the only hipMalloc is the
// growth itself.  mc_reserve_conv_split_workspace() sizes it ahead of time (the runtime does at initModel), after
// which these entry points allocate nothing.
struct SplitBuf { int dev; hipStream_t st; float* p; size_t cap; };
static std::mutex g_splitMu;
static std::vector<SplitBuf> g_splitBufs;

float* split_scratch(size_t bytes, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_splitMu);
    SplitBuf* b = nullptr;
    for (auto& e : g_splitBufs)
        if (e.dev == dev && e.st == st) { b = &e; break; }
    if (b && b->cap >= bytes) return b->p;
    size_t cap = (bytes + ((size_t)1 << 20) - 1) >> 20 << 20;
    if (b && cap < b->cap + b->cap / 2) cap = b->cap + b->cap / 2;   // geometric growth bounds what the retired buffers hold
    float* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (b) { b->p = p; b->cap = cap; }   // the old buffer stays allocated until the process ends
    else g_splitBufs.push_back(SplitBuf{dev, st, p, cap});
    return p;
}

inline int grid_items(size_t items) {
    const size_t want = (items + TPB - 1) / TPB;
    const size_t cap = (size_t)num_cus() * 16;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

int reduce_slabs(const float* slabs, float* C, int m, long n, int chunks, float alpha, float beta, const float* bias, int act, hipStream_t st) {
    if (n % 4 == 0 && aligned16(C) && aligned16(slabs))
        hipLaunchKernelGGL(k_splitk_reduce4, dim3(grid_items((size_t)m * (n / 4))), dim3(TPB), 0, st, reinterpret_cast<const float4*>(slabs), C, m,
                           (int)(n / 4), (int)n, chunks, alpha, beta, bias, act);
    else
        hipLaunchKernelGGL(k_splitk_reduce, dim3(grid_items((size_t)m * n)), dim3(TPB), 0, st, slabs, C, m, (int)n, (int)n, chunks, alpha, beta, bias,
                           act, (const float*)nullptr, 0);
    return launch_status("conv split-K reduce");
}

// kc: the weight-gradient orientation (transA = 0, transB = 1: both operands k-contiguous), the one with 16x16x4 tiles
int pick_gemm_chunks(int m, int n, int k, bool kc) {
    static const char* forced = getenv("MC_GEMM_CHUNKS");   // tuning knob (benchmarks only)
    if (forced && forced[0]) return atoi(forced) < 1 ? 1 : atoi(forced);
    static const bool nomf = getenv("MC_IGEMM_NOMF") != nullptr;
    if (kc && !nomf && k >= 2048) {
        // A weight gradient whose whole output is one 16x16x4 tile (NIN's cccp layers): one workgroup per K chunk owns all of C,
        // so each operand element is fetched once device-wide and no matrix work is padding.  One round of workgroups: a chunk
        // per CU when k allows 256 per chunk; with fewer chunks than that the tile is halved along its rows (the launch picks).
        if (mf_whole_output(m, n)) {
            long want = num_cus();
            const long maxc = k / 256;
            if (want > maxc) want = maxc;
            // Too short a k for three quarters of the CUs to get a whole-output workgroup each: 96 x 96 quarter tiles on a
            // quarter of the CUs' worth of chunks instead (one round of workgroups, half the slabs to reduce).  Measured
            // (tools/mf_probe.py, product + reduction + row sums): 192 x 192 x 32768 (NIN cccp3 / cccp4) 39.9 us as 96 x 192 halves
            // of 128 chunks, 36.3 as quarters of 64; 192 x 192 x 8192 (cccp5) 20.3 -> 18.2.
            if (4 * want < 3L * num_cus() && m > 96 && n > 96) {
                want = num_cus() / 4;
                if (want > k / 128) want = k / 128;
            }
            return want < 1 ? 1 : (int)want;
        }
    }
    // tiles of the shape launch_rows_cols will pick for a long-k product: 128 x 192 when that pads the
    // column extent less, else 96 x 128
    const bool wide = (long)ceil_div(n, 192) * 192 < (long)ceil_div(n, 128) * 128 && m > 96;
    const long tiles = wide ? (long)ceil_div(m, 128) * ceil_div(n, 192) : (long)ceil_div(m, 96) * ceil_div(n, 128);
    // a product that fits a handful of tiles is one serial K loop per workgroup (layers-80sec fc64 forward: 64 x 128 x 1024 on
    // ONE workgroup, 64 K steps): split from k = 1024 when the tiles cover less than an eighth of the device
    if (k < (tiles * 8 <= num_cus() ? 1024 : 2048) || tiles >= num_cus()) return 1;
    long want = (2L * num_cus() + tiles - 1) / tiles;
    const long maxc = k / 256;   // at least 256 k per chunk (8-16 LDS tiles)
    if (want > maxc) want = maxc;
    if (want > 128) want = 128;
    return want < 1 ? 1 : (int)want;
}

}  // namespace
}  // namespace mc

using namespace mc;

extern "C" {

const char* mc_last_error(void) { return g_err; }

int mc_device_info(char* archName, int archNameLen, int* numCUs) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("mc_device_info: %s", hipGetErrorString(e)); return MC_ELAUNCH; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { set_error("mc_device_info: %s", hipGetErrorString(e)); return MC_ELAUNCH; }
    if (archName && archNameLen > 0) {
        strncpy(archName, prop.gcnArchName, archNameLen - 1);
        archName[archNameLen - 1] = 0;
    }
    if (numCUs) *numCUs = prop.multiProcessorCount;
    return MC_OK;
}

int mc_conv_filter_acts(const float* images, const float* filters, float* targets, int numImages, int numImgColors,
                        int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                        int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                        mcStream_t s) {
    return mc_conv_filter_acts_fused(images, filters, targets, nullptr, MC_NEURON_NONE, numImages, numImgColors, imgSizeY, imgSizeX,
                                     numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups,
                                     scaleTargets, scaleOutput, s);
}

static int conv_fwd_entry(const float* images, const float* filters, float* targets, const float* biases, int neuron,
                          int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                          int paddingStart, int moduleStride, int numModulesY, int numModulesX, int numGroups,
                          float scaleTargets, float scaleOutput, unsigned* reluBits, mcStream_t s);
static std::atomic<int> g_convBand{[] { const char* e = getenv("MC_CONV_BAND"); return e ? atoi(e) : -1; }()};
static std::map<std::array<int, 12>, int> g_bandChoice;
static std::mutex g_bandMu;

int mc_conv_filter_acts_fused(const float* images, const float* filters, float* targets, const float* biases, int neuron,
                              int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                              int paddingStart, int moduleStride, int numModulesY, int numModulesX, int numGroups,
                              float scaleTargets, float scaleOutput, mcStream_t s) {
    return conv_fwd_entry(images, filters, targets, biases, neuron, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                          paddingStart, moduleStride, numModulesY, numModulesX, numGroups, scaleTargets, scaleOutput, nullptr, s);
}

int mc_conv_filter_acts_relu_bits(const float* images, const float* filters, float* targets, const float* biases, unsigned* reluBits,
                                  int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                                  int paddingStart, int moduleStride, int numModulesY, int numModulesX, mcStream_t s) {
    MC_REQUIRE(reluBits, "mc_conv_filter_acts_relu_bits: null reluBits");
    return conv_fwd_entry(images, filters, targets, biases, MC_NEURON_RELU, numImages, numImgColors, imgSizeY, imgSizeX, numFilters,
                          filterSize, paddingStart, moduleStride, numModulesY, numModulesX, 1, 0.f, 1.f, reluBits, s);
}

static int conv_fwd_entry(const float* images, const float* filters, float* targets, const float* biases, int neuron,
                          int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                          int paddingStart, int moduleStride, int numModulesY, int numModulesX, int numGroups,
                          float scaleTargets, float scaleOutput, unsigned* reluBits, mcStream_t s) {
    MC_REQUIRE(images && filters && targets, "mc_conv_filter_acts_fused: null pointer");
    MC_REQUIRE(neuron == MC_NEURON_NONE || neuron == MC_NEURON_RELU, "mc_conv_filter_acts_fused: unknown neuron %d", neuron);
    // few columns (modules x images): split K so the grid fills the device, reduce the slabs with the epilogue
    if (numImages > 0 && numImgColors > 0 && numFilters > 0 && numGroups > 0 && filterSize > 0 && numModulesY > 0 && numModulesX > 0 &&
        numFilters % numGroups == 0 && numImgColors % numGroups == 0) {
        const long cols = (long)numModulesY * numModulesX * numImages;
        int kc = 0;
        const int chunks = conv_split_plan(numFilters / numGroups, cols, numImgColors / numGroups * filterSize * filterSize, numGroups, &kc);
        static float describeOnly[4];
        float* slabs = chunks > 1 ? (describing() ? describeOnly : split_scratch((size_t)chunks * numFilters * cols * sizeof(float), S(s))) : nullptr;
        if (slabs) {
            int rc = conv_filter_acts_impl(images, filters, targets, nullptr, MC_NEURON_NONE, numImages, numImgColors, imgSizeY, imgSizeX,
                                           numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0,
                                           0.f, 1.f, slabs, kc, nullptr, S(s));
            if (rc || describing()) return rc;
            if (reluBits) {
                if (cols % 4 == 0 && aligned16(targets) && aligned16(slabs)) {
                    const long tiles = (long)ceil_div(numFilters, 32) * ceil_div(cols, 128);
                    hipLaunchKernelGGL(k_splitk_reduce_bits4, dim3((unsigned)(tiles < 16L * num_cus() ? tiles : 16L * num_cus())), dim3(256), 0, S(s),
                                       reinterpret_cast<const float4*>(slabs), targets, reluBits, numFilters, (int)cols, chunks, scaleOutput, biases, neuron);
                } else
                hipLaunchKernelGGL(k_splitk_reduce_bits, dim3(grid_items((size_t)ceil_div(numFilters, 32) * cols)), dim3(TPB), 0, S(s), slabs, targets,
                                   reluBits, numFilters, (int)cols, chunks, scaleOutput, biases, neuron);
                return launch_status("conv split-K reduce (bits)");
            }
            return reduce_slabs(slabs, targets, numFilters, cols, chunks, scaleOutput, scaleTargets, biases, neuron, S(s));
        }
    }
    auto engine = [&]() {
        return conv_filter_acts_impl(images, filters, targets, biases, neuron, numImages, numImgColors, imgSizeY, imgSizeX,
                                     numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0,
                                     scaleTargets, scaleOutput, nullptr, 0, reluBits, S(s));
    };
    // Small-filter layers (conv_band.hip): the band kernel gives the same bits as the engine, so which of the two serves a shape is
    // decided by one timed trial per shape and device, on the caller's stream, at the first call that may be repeated
    // (scaleTargets == 0) and may block (not under mc_set_autotune(0)).  g_convBand: -1 trial, 0 engine only, 1 band wherever it exists.
    const int bandKind = (scaleTargets == 0.f && paddingStart <= 0 && aligned16(images) && aligned16(filters) && aligned16(targets))
                             ? conv_band_kind(numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize, -paddingStart, moduleStride,
                                              numModulesY, numModulesX, numGroups)
                             : 0;
    if (!bandKind || g_convBand == 0) return engine();
    auto band = [&]() {
        return conv_band_launch(images, filters, targets, biases, neuron, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                                -paddingStart, numModulesY, numModulesX, scaleOutput, reluBits, S(s));
    };
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::array<int, 12> key{numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize, paddingStart, numModulesY, numModulesX,
                                  reluBits ? 1 : 0, neuron, dev};
    int choice = g_convBand == 1 ? 1 : -1;
    if (choice < 0) {
        std::lock_guard<std::mutex> lk(g_bandMu);
        auto it = g_bandChoice.find(key);
        if (it != g_bandChoice.end()) choice = it->second;
    }
    if (describing()) {
        if (choice != 1) return engine();
        char buf[128];
        snprintf(buf, sizeof buf, "k_conv_band kind %d (filter size, channel chunk, 32-filter blocks) tile F x (2x2 pixels x 32 images)", bandKind);
        describe_write(buf);
        return MC_OK;
    }
    if (choice == 1) return band();
    if (choice == 0 || tune_inhibited()) return engine();
    // the trial: the engine first (it tunes its own tile on this call), then two timed runs of each
    int rc = engine();
    if (rc != MC_OK) return rc;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MC_OK;
    float ms[2] = {1e30f, 1e30f};
    for (int which = 0; which < 2 && rc == MC_OK; ++which) {
        rc = which ? band() : engine();   // warm
        if (rc != MC_OK) break;
        (void)hipEventRecord(e0, S(s));
        for (int r = 0; r < 2 && rc == MC_OK; ++r) rc = which ? band() : engine();
        (void)hipEventRecord(e1, S(s));
        float el = 0.f;
        if (rc == MC_OK && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&el, e0, e1) == hipSuccess) ms[which] = el * 0.5f;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != MC_OK) return rc;
    choice = ms[1] < 0.97f * ms[0] ? 1 : 0;   // the engine stays unless the band kernel is at least 3 % faster
    static const bool verbose = getenv("MC_CONV_BAND_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[mc conv band] C %d %dx%d F %d N %d: engine %.1f us, band %.1f us -> %s\n", numImgColors, imgSizeY, imgSizeX, numFilters,
                numImages, ms[0] * 1e3f, ms[1] * 1e3f, choice ? "band" : "engine");
    {
        std::lock_guard<std::mutex> lk(g_bandMu);
        g_bandChoice[key] = choice;
    }
    return choice ? MC_OK : engine();   // the last run left the band kernel's output, which is the same bits
}

int mc_set_conv_band(int mode) {
    MC_REQUIRE(mode >= -1 && mode <= 1, "mc_set_conv_band: mode %d (want -1 trial, 0 engine only, 1 band kernel wherever it exists)", mode);
    g_convBand = mode;
    return MC_OK;
}

int mc_conv_img_acts(const float* hidActs, const float* filters, float* targets, int numImages, int numImgColors,
                     int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                     int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(hidActs && filters && targets, "mc_conv_img_acts: null pointer");
    if (numImages > 0 && numImgColors > 0 && numFilters > 0 && numGroups > 0 && filterSize > 0 && imgSizeY > 0 && imgSizeX > 0 &&
        numFilters % numGroups == 0 && numImgColors % numGroups == 0) {
        const long cols = (long)imgSizeY * imgSizeX * numImages;
        int kc = 0;
        const int chunks = conv_split_plan(numImgColors / numGroups, cols, filterSize * filterSize * (numFilters / numGroups), numGroups, &kc);
        float* slabs = chunks > 1 ? split_scratch((size_t)chunks * numImgColors * cols * sizeof(float), S(s)) : nullptr;
        if (slabs) {
            int rc = conv_img_acts_impl(hidActs, filters, targets, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                                        paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0, 0.f, 1.f, slabs, kc, S(s));
            if (rc) return rc;
            return reduce_slabs(slabs, targets, numImgColors, cols, chunks, scaleOutput, scaleTargets, nullptr, MC_NEURON_NONE, S(s));
        }
    }
    return conv_img_acts_impl(hidActs, filters, targets, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                              paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0, scaleTargets, scaleOutput,
                              nullptr, 0, S(s));
}

int mc_conv_weight_acts(const float* images, const float* hidActs, float* targets, int numImages, int numImgColors,
                        int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                        int numModulesY, int numModulesX, int numGroups, int partialSum, float scaleTargets,
                        float scaleOutput, mcStream_t s) {
    MC_REQUIRE(images && hidActs && targets, "mc_conv_weight_acts: null pointer");
    return conv_weight_acts_impl(images, hidActs, targets, numImages, numImgColors, imgSizeY, imgSizeX, numFilters,
                                 filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, partialSum, 0,
                                 scaleTargets, scaleOutput, nullptr, 0, 0.f, S(s));
}

int mc_local_weight_acts(const float* images, const float* hidActs, float* targets, int numImages, int numImgColors,
                         int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                         int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                         mcStream_t s) {
    MC_REQUIRE(images && hidActs && targets, "mc_local_weight_acts: null pointer");
    // unshared weight gradient == shared one with partialSum = 1 (weight_acts.cu:1103-1108): every
    // module's chunk IS that module's filter bank.
    return conv_weight_acts_impl(images, hidActs, targets, numImages, numImgColors, imgSizeY, imgSizeX, numFilters,
                                 filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 1, 0,
                                 scaleTargets, scaleOutput, nullptr, 0, 0.f, S(s));
}

int mc_conv_weight_acts_pick_partial_sum(int numImages, int numImgColors, int numFilters, int filterSize, int numModules,
                                         int numGroups) {
    if (numImages <= 0 || numImgColors <= 0 || numFilters <= 0 || filterSize <= 0 || numModules <= 0 || numGroups <= 0) return 0;
    const long rows = (long)(numImgColors / numGroups) * filterSize * filterSize;
    const long tiles = (long)ceil_div(rows, 96) * ceil_div(numFilters / numGroups, 256) * numGroups;
    long want = (2L * num_cus() + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > numModules) want = numModules;
    // keep each chunk's k extent (partialSum*numImages) >= 512 where possible
    while (want > 1 && (long)(numModules / want) * numImages < 512) --want;
    int best = numModules;  // partialSum candidates: divisors of numModules
    for (int ps = 1; ps <= numModules; ++ps)
        if (numModules % ps == 0 && numModules / ps >= want) best = ps;
    return best;
}

int mc_local_filter_acts(const float* images, const float* filters, float* targets, int numImages, int numImgColors,
                         int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                         int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                         mcStream_t s) {
    MC_REQUIRE(images && filters && targets, "mc_local_filter_acts: null pointer");
    return conv_filter_acts_impl(images, filters, targets, nullptr, MC_NEURON_NONE, numImages, numImgColors, imgSizeY, imgSizeX,
                                 numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 1,
                                 scaleTargets, scaleOutput, nullptr, 0, nullptr, S(s));
}

int mc_local_img_acts(const float* hidActs, const float* filters, float* targets, int numImages, int numImgColors,
                      int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                      int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(hidActs && filters && targets, "mc_local_img_acts: null pointer");
    return conv_img_acts_impl(hidActs, filters, targets, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                              paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 1, scaleTargets, scaleOutput,
                              nullptr, 0, S(s));
}

size_t mc_gemm_workspace_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int c0 = pick_gemm_chunks(m, n, k, false), c1 = pick_gemm_chunks(m, n, k, true);   // the caller does not say which orientation
    const int chunks = c0 > c1 ? c0 : c1;
    return chunks > 1 ? (size_t)chunks * ((size_t)m * n + m) * sizeof(float) : 0;
}
// (Measured and dropped: cutting the 160 / 192 columns of the dense weight gradients into 128 + remainder, two products on
// exact tiles instead of one padded to 256 columns -- cccp1 121 -> 115 us, but cccp2 67 -> 75 and cccp3 42 -> 48: the second
// launch and reduction cost more than the padding, and these products are bound by their k-contiguous loaders anyway.)

static int gemm_entry(const char* who, const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB, int lda,
                      int ldb, int ldc, float scaleTargets, float scaleOutput, const float* rowBias, int neuron, const float* mask,
                      int ldmask, void* workspace, size_t workspaceBytes, mcStream_t s, unsigned* bitsOut = nullptr,
                      const unsigned* bitsIn = nullptr, float* rowSums = nullptr, float sumScale = 0.f) {
    MC_REQUIRE(A && B && C, "%s: null pointer", who);
    MC_REQUIRE(m > 0 && n > 0 && k > 0, "%s: non-positive dimension (%d,%d,%d)", who, m, n, k);
    MC_REQUIRE(lda >= (transA ? m : k) && ldb >= (transB ? k : n) && ldc >= n, "%s: leading dimension too small", who);
    MC_REQUIRE(neuron == MC_NEURON_NONE || neuron == MC_NEURON_RELU, "%s: unknown neuron %d", who, neuron);
    MC_REQUIRE(!mask || ldmask >= n, "%s: mask leading dimension too small", who);
    MC_REQUIRE((size_t)(transA ? k : m) * lda < (1ull << 30) && (size_t)(transB ? n : k) * ldb < (1ull << 30) &&
                   (size_t)m * ldc < (1ull << 30) && (!mask || (size_t)m * ldmask < (1ull << 30)),
               "%s: operand exceeds the 4 GiB buffer-descriptor range", who);
    int chunks = (bitsOut || bitsIn) ? 1 : pick_gemm_chunks(m, n, k, !transA && transB);   // the bit-mask epilogues live in the unsplit product
    // with row sums the chunks' m partial sums each follow the slabs
    const size_t slabBytes = (size_t)chunks * ((size_t)m * n + (rowSums ? (size_t)m : 0)) * sizeof(float);
    if (chunks > 1 && workspace && workspaceBytes >= slabBytes && aligned16(workspace)) {
        float* slabs = static_cast<float*>(workspace);
        float* sumSlabs = slabs + (size_t)chunks * m * n;
        const long sumStride = m;
        int rc = gemm_impl(A, B, slabs, m, n, k, transA, transB, lda, ldb, n, 0.f, 1.f, nullptr, MC_NEURON_NONE, nullptr, 0, chunks,
                           (long)m * n, nullptr, nullptr, rowSums ? sumSlabs : nullptr, sumStride, 1.f, S(s));
        if (rc) return rc;
        // gemm_impl may round the chunk size up; recompute how many slabs were written
        int kChunk = (ceil_div(k, chunks) + 31) / 32 * 32;
        const int written = ceil_div(k, kChunk);
        if (rowSums && ldc == n && !rowBias && neuron == MC_NEURON_NONE && !mask)   // plain contiguous C: slabs and row sums in one launch
            return mc__sum_cols_pair(slabs, C, m * n, scaleTargets, scaleOutput, sumSlabs, rowSums, m, sumScale, written, 16, S(s));
        if (n % 4 == 0 && ldc % 4 == 0 && aligned16(C) && written >= 16)
            hipLaunchKernelGGL(k_splitk_reduce_tall, dim3(ceil_div((long)m * (n / 4), 16)), dim3(256), 0, S(s),
                               reinterpret_cast<const float4*>(slabs), C, m, n / 4, ldc, written, scaleOutput, scaleTargets, rowBias, neuron,
                               mask, ldmask);
        else
            hipLaunchKernelGGL(k_splitk_reduce, dim3(grid_items((size_t)m * n)), dim3(TPB), 0, S(s), slabs, C, m, n, ldc,
                               written, scaleOutput, scaleTargets, rowBias, neuron, mask, ldmask);
        if (launch_status("mc_gemm(split-k reduce)") != MC_OK) return MC_ELAUNCH;
        if (rowSums)   // the chunks' partial row sums: a (written x m) matrix summed over its rows (parallel over the chunks too)
            return mc_sum_cols(sumSlabs, rowSums, written, m, 0.f, sumScale, s);
        return launch_status("mc_gemm(split-k reduce)");
    }
    return gemm_impl(A, B, C, m, n, k, transA, transB, lda, ldb, ldc, scaleTargets, scaleOutput, rowBias, neuron, mask, ldmask, 1, 0,
                     bitsOut, bitsIn, rowSums, 0, sumScale, S(s));
}

size_t mc_gemm_row_sums_workspace_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int chunks = pick_gemm_chunks(m, n, k, true);
    return chunks > 1 ? (size_t)chunks * ((size_t)m * n + m) * sizeof(float) : 0;
}

int mc_gemm_row_sums(const float* A, const float* B, float* C, int m, int n, int k, int lda, int ldb, int ldc, float scaleTargets,
                     float scaleOutput, float* rowSums, float sumScale, void* workspace, size_t workspaceBytes, mcStream_t s) {
    MC_REQUIRE(rowSums, "mc_gemm_row_sums: null rowSums");
    return gemm_entry("mc_gemm_row_sums", A, B, C, m, n, k, 0, 1, lda, ldb, ldc, scaleTargets, scaleOutput, nullptr, MC_NEURON_NONE, nullptr, 0,
                      workspace, workspaceBytes, s, nullptr, nullptr, rowSums, sumScale);
}

int mc_gemm(const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB, int lda, int ldb,
            int ldc, float scaleTargets, float scaleOutput, const float* rowBias, int neuron, void* workspace,
            size_t workspaceBytes, mcStream_t s) {
    return gemm_entry("mc_gemm", A, B, C, m, n, k, transA, transB, lda, ldb, ldc, scaleTargets, scaleOutput, rowBias, neuron, nullptr, 0,
                      workspace, workspaceBytes, s);
}

int mc_gemm_relu_grad(const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB, int lda, int ldb,
                      int ldc, float scaleTargets, float scaleOutput, const float* reluActs, int ldr, void* workspace,
                      size_t workspaceBytes, mcStream_t s) {
    MC_REQUIRE(reluActs, "mc_gemm_relu_grad: null reluActs");
    return gemm_entry("mc_gemm_relu_grad", A, B, C, m, n, k, transA, transB, lda, ldb, ldc, scaleTargets, scaleOutput, nullptr,
                      MC_NEURON_NONE, reluActs, ldr, workspace, workspaceBytes, s);
}

// Dry run: the kernel instantiation and grid mc_conv_filter_acts_fused / _relu_bits would launch for this shape right now (the
// autotuner's cached choice if the shape has been tuned in this process, else the static rule's).  Launches nothing.
int mc_describe_conv_filter_acts(char* out, int outLen, int reluBits, int numImages, int numImgColors, int imgSizeY, int imgSizeX,
                                 int numFilters, int filterSize, int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                                 int numGroups) {
    MC_REQUIRE(out && outLen > 0, "mc_describe_conv_filter_acts: no output buffer");
    out[0] = 0;
    // never dereferenced: nothing is launched.  16-byte aligned on purpose: the entry picks its vector / scalar instantiation (and
    // with it the autotune cache it consults) from the operands' alignment, and the description is of the aligned case.
    alignas(16) static float dummy[4];
    set_describe(out, outLen);
    const int rc = conv_fwd_entry(dummy, dummy, dummy, nullptr, reluBits ? MC_NEURON_RELU : MC_NEURON_NONE, numImages, numImgColors, imgSizeY, imgSizeX,
                                  numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0.f, 1.f,
                                  reluBits ? reinterpret_cast<unsigned*>(dummy) : nullptr, nullptr);
    set_describe(nullptr, 0);
    return rc;
}

int mc_set_autotune(int enabled) {
    set_tune_inhibit(enabled == 0);
    return MC_OK;
}

size_t mc_relu_bits_words(int rows, int ld) { return rows > 0 && ld > 0 ? (size_t)((rows + 31) / 32) * ld : 0; }

int mc_gemm_relu_bits(const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB, int lda, int ldb, int ldc,
                      const float* rowBias, unsigned* reluBits, mcStream_t s) {
    MC_REQUIRE(reluBits, "mc_gemm_relu_bits: null reluBits");
    MC_REQUIRE(!transA && !transB, "mc_gemm_relu_bits: only the forward orientation (transA = transB = 0) writes bit masks");
    return gemm_entry("mc_gemm_relu_bits", A, B, C, m, n, k, transA, transB, lda, ldb, ldc, 0.f, 1.f, rowBias, MC_NEURON_RELU, nullptr, 0,
                      nullptr, 0, s, reluBits, nullptr);
}

int mc_gemm_relu_grad_bits(const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB, int lda, int ldb,
                           int ldc, float scaleOutput, const unsigned* reluBits, mcStream_t s) {
    MC_REQUIRE(reluBits, "mc_gemm_relu_grad_bits: null reluBits");
    MC_REQUIRE(transA && !transB, "mc_gemm_relu_grad_bits: only the backward orientation (transA = 1, transB = 0) reads bit masks");
    return gemm_entry("mc_gemm_relu_grad_bits", A, B, C, m, n, k, transA, transB, lda, ldb, ldc, 0.f, scaleOutput, nullptr, MC_NEURON_NONE,
                      nullptr, 0, nullptr, 0, s, nullptr, reluBits);
}

int mc_relu_bits_from_acts(const float* acts, unsigned* reluBits, int rows, int cols, int ld, mcStream_t s) {
    MC_REQUIRE(acts && reluBits && rows > 0 && cols > 0 && ld >= cols, "mc_relu_bits_from_acts: bad arguments");
    hipLaunchKernelGGL(k_relu_bits_from_acts, dim3(grid_items((size_t)ceil_div(rows, 32) * cols)), dim3(TPB), 0, S(s), acts, reluBits, rows, cols, ld);
    return launch_status("mc_relu_bits_from_acts");
}

size_t mc_conv_weight_grad_workspace_bytes(int numImages, int numImgColors, int numFilters, int filterSize, int numModules,
                                           int numGroups) {
    if (numImages <= 0 || numImgColors <= 0 || numFilters <= 0 || filterSize <= 0 || numModules <= 0 || numGroups <= 0) return 0;
    int kc = 0;
    const int chunks = conv_weight_grad_plan(numImages, numImgColors, numFilters, filterSize, numModules, numGroups, &kc);
    return chunks > 1 ? (size_t)chunks * (numImgColors / numGroups) * filterSize * filterSize * numFilters * sizeof(float) : 0;
}

size_t mc_conv_weight_grad_bias_workspace_bytes(int numImages, int numImgColors, int numFilters, int filterSize, int numModules,
                                                int numGroups) {
    const size_t w = mc_conv_weight_grad_workspace_bytes(numImages, numImgColors, numFilters, filterSize, numModules, numGroups);
    if (w == 0) return 0;
    const size_t wel = (size_t)(numImgColors / numGroups) * filterSize * filterSize * numFilters;
    return w + w / (wel * sizeof(float)) * numFilters * sizeof(float);   // every chunk: its slab and numFilters partial sums
}

static int conv_weight_grad_entry(const char* who, const float* images, const float* hidActs, float* weightGrad, float* biasGrad, int numImages,
                                  int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart,
                                  int moduleStride, int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                                  float biasScale, void* workspace, size_t workspaceBytes, mcStream_t s) {
    MC_REQUIRE(images && hidActs && weightGrad, "%s: null pointer", who);
    MC_REQUIRE(numImages > 0 && numImgColors > 0 && numFilters > 0 && filterSize > 0 && numModulesY > 0 && numModulesX > 0 &&
                   numGroups > 0 && numImgColors % numGroups == 0 && numFilters % numGroups == 0,
               "%s: bad dimensions", who);
    const int M = numModulesY * numModulesX;
    int kc = 0;
    const int chunks = conv_weight_grad_plan(numImages, numImgColors, numFilters, filterSize, M, numGroups, &kc);
    if (chunks <= 1)
        return conv_weight_acts_impl(images, hidActs, weightGrad, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                                     paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0, 0, scaleTargets, scaleOutput,
                                     biasGrad, 0, biasScale, S(s));
    const size_t wel = (size_t)(numImgColors / numGroups) * filterSize * filterSize * numFilters;
    const size_t need = (size_t)chunks * (wel + (biasGrad ? numFilters : 0)) * sizeof(float);
    MC_REQUIRE(workspace && aligned16(workspace) && workspaceBytes >= need, "%s: workspace of %zu bytes needed (%s_workspace_bytes)", who, need, who);
    float* slabs = static_cast<float*>(workspace);
    float* sumSlabs = slabs + (size_t)chunks * wel;   // (chunks x numFilters), after the weight slabs
    int rc = conv_weight_acts_impl(images, hidActs, slabs, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                                   paddingStart, moduleStride, numModulesY, numModulesX, numGroups, 0, kc, 0.f, 1.f,
                                   biasGrad ? sumSlabs : nullptr, numFilters, 1.f, S(s));
    if (rc) return rc;
    if (!biasGrad) return mc_sum_cols(slabs, weightGrad, chunks, (int)wel, scaleTargets, scaleOutput, s);
    return mc__sum_cols_pair(slabs, weightGrad, (int)wel, scaleTargets, scaleOutput, sumSlabs, biasGrad, numFilters, biasScale, chunks, 32, S(s));
}

int mc_conv_weight_grad(const float* images, const float* hidActs, float* weightGrad, int numImages, int numImgColors,
                        int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                        int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                        void* workspace, size_t workspaceBytes, mcStream_t s) {
    return conv_weight_grad_entry("mc_conv_weight_grad", images, hidActs, weightGrad, nullptr, numImages, numImgColors, imgSizeY, imgSizeX,
                                  numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups, scaleTargets,
                                  scaleOutput, 0.f, workspace, workspaceBytes, s);
}

int mc_conv_weight_grad_bias(const float* images, const float* hidActs, float* weightGrad, float* biasGrad, int numImages,
                             int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart,
                             int moduleStride, int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                             float biasScale, void* workspace, size_t workspaceBytes, mcStream_t s) {
    MC_REQUIRE(biasGrad, "mc_conv_weight_grad_bias: null biasGrad");
    return conv_weight_grad_entry("mc_conv_weight_grad_bias", images, hidActs, weightGrad, biasGrad, numImages, numImgColors, imgSizeY,
                                  imgSizeX, numFilters, filterSize, paddingStart, moduleStride, numModulesY, numModulesX, numGroups,
                                  scaleTargets, scaleOutput, biasScale, workspace, workspaceBytes, s);
}


size_t mc_conv_split_workspace_bytes(int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                                     int numModulesY, int numModulesX, int numGroups) {
    if (numImages <= 0 || numImgColors <= 0 || numFilters <= 0 || numGroups <= 0 || filterSize <= 0 || numModulesY <= 0 || numModulesX <= 0 ||
        imgSizeY <= 0 || imgSizeX <= 0 || numFilters % numGroups || numImgColors % numGroups)
        return 0;
    int kc = 0;
    const long fcols = (long)numModulesY * numModulesX * numImages, bcols = (long)imgSizeY * imgSizeX * numImages;
    const int fch = conv_split_plan(numFilters / numGroups, fcols, numImgColors / numGroups * filterSize * filterSize, numGroups, &kc);
    const int bch = conv_split_plan(numImgColors / numGroups, bcols, filterSize * filterSize * (numFilters / numGroups), numGroups, &kc);
    const size_t f = fch > 1 ? (size_t)fch * numFilters * fcols * sizeof(float) : 0;
    const size_t b = bch > 1 ? (size_t)bch * numImgColors * bcols * sizeof(float) : 0;
    return f > b ? f : b;
}

int mc_reserve_conv_split_workspace(size_t bytes, mcStream_t s) {
    if (bytes == 0) return MC_OK;
    if (!split_scratch(bytes, S(s))) { set_error("mc_reserve_conv_split_workspace: cannot allocate %zu bytes", bytes); return MC_EWORKSPACE; }
    return MC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The entry point that allocates nothing.  mc_conv_weight_grad's split-K slabs (the workspace argument there) live in a
// file-scope device array instead; its device address is looked up once per device with hipGetSymbolAddress and only
// that pointer is ever handed to kernels.
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
constexpr size_t kStaticScratchBytes = MC_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) float scratch[kStaticScratchBytes / sizeof(float)];
constexpr int kMaxDevices = 64;
static std::mutex g_scratchMu;
static float* g_scratchPtr[kMaxDevices];   // device address of `scratch` on each device, filled on first use

// `scratch` as a name is a host-side handle: only the address hipGetSymbolAddress returns for the CURRENT device is real
int static_scratch(float** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("mc_conv_weight_grad_noalloc: hipGetDevice: %s", hipGetErrorString(e)); return MC_ELAUNCH; }
    if (dev < 0 || dev >= kMaxDevices) { set_error("mc_conv_weight_grad_noalloc: device ordinal %d not supported (max %d)", dev, kMaxDevices - 1); return MC_EINVAL; }
    std::lock_guard<std::mutex> lk(g_scratchMu);
    if (!g_scratchPtr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("mc_conv_weight_grad_noalloc: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
            return MC_ELAUNCH;
        }
        g_scratchPtr[dev] = static_cast<float*>(p);
    }
    *out = g_scratchPtr[dev];
    return MC_OK;
}
}  // namespace

extern "C" {

size_t mc_noalloc_scratch_bytes(void) { return kStaticScratchBytes; }

int mc_conv_weight_grad_noalloc(const float* images, const float* hidActs, float* weightGrad, int numImages, int numImgColors,
                                int imgSizeY, int imgSizeX, int numFilters, int filterSize, int paddingStart, int moduleStride,
                                int numModulesY, int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                                mcStream_t s) {
    MC_REQUIRE(images && hidActs && weightGrad, "mc_conv_weight_grad_noalloc: null pointer");
    const size_t need = mc_conv_weight_grad_workspace_bytes(numImages, numImgColors, numFilters, filterSize,
                                                            numModulesY > 0 && numModulesX > 0 ? numModulesY * numModulesX : 0, numGroups);
    if (need > kStaticScratchBytes) {
        set_error("mc_conv_weight_grad_noalloc: this shape needs %zu bytes of split-K scratch, the library's static array holds %zu; "
                  "use mc_conv_weight_grad with a workspace", need, kStaticScratchBytes);
        return MC_EWORKSPACE;
    }
    float* ws = nullptr;
    int rc = static_scratch(&ws);
    if (rc) return rc;
    // every slab element the reduction reads is written by the product first (chunks cover the whole (module, image) axis),
    // so nothing left in the array by an earlier call is ever read: no clearing needed
    const bool was = tune_inhibited();
    set_tune_inhibit(true);   // the timed tile trial creates events and blocks: not from this entry point
    rc = mc_conv_weight_grad(images, hidActs, weightGrad, numImages, numImgColors, imgSizeY, imgSizeX, numFilters, filterSize,
                             paddingStart, moduleStride, numModulesY, numModulesX, numGroups, scaleTargets, scaleOutput, ws,
                             kStaticScratchBytes, s);
    set_tune_inhibit(was);
    return rc;
}

}  // extern "C"
