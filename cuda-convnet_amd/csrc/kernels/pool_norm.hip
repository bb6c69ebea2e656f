// pool_norm.hip -- pooling, pooling-undo and response-normalisation kernels for gfx950.
// HBM-bound: one lane owns 4 consecutive images (16 B) of one (channel, pixel) and walks the
// window; window re-reads are served by L1/L2, each tensor crosses HBM once.
// Semantics follow Kernel/include/conv/conv_util.cuh:71-194 and Kernel/src/conv/conv_util.cu
// (kLocalMaxUndo :1502, kLocalAvgUndo :1395, kCNorm_manyfilter :831, kRNormUndoPrelims :1603,
// kRNormUndo :1639, kFCNorm :1041, kFRNormUndo :1115).
#include "mc_common.h"
#include "philox.h"

namespace mc {
namespace {

constexpr int TPB = 256;

// V = 4: float4 per lane (numImages % 4 == 0, aligned); V = 1: scalar fallback.
template <int V> struct Vec;
template <> struct Vec<4> {
    typedef float4 T;
    __device__ static T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static void st(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
    __device__ static T splat(float s) { return make_float4(s, s, s, s); }
};
template <> struct Vec<1> {
    typedef float T;
    __device__ static T ld(const float* p) { return *p; }
    __device__ static void st(float* p, T v) { *p = v; }
    __device__ static T splat(float s) { return s; }
};
#define MC_EACH4(EXPR_X, EXPR_Y, EXPR_Z, EXPR_W) EXPR_X; EXPR_Y; EXPR_Z; EXPR_W
__device__ __forceinline__ float4 vmax(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vdivs(float4 a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ float vdivs(float a, float s) { return a / s; }
__device__ __forceinline__ float4 vdiv(float4 a, float4 b) { return make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
__device__ __forceinline__ float vdiv(float a, float b) { return a / b; }
// (img == act) * grad, the reference's equality routing (conv_util.cu:1567)
__device__ __forceinline__ float4 veqmul(float4 i, float4 a, float4 g) {
    return make_float4((i.x == a.x) * g.x, (i.y == a.y) * g.y, (i.z == a.z) * g.z, (i.w == a.w) * g.w);
}
__device__ __forceinline__ float veqmul(float i, float a, float g) { return (i == a) * g; }
// relu gradient mask: v where y > 0, else 0 (neuron.cuh:173-178)
__device__ __forceinline__ float4 vposmask(float4 y, float4 v) {
    return make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
}
__device__ __forceinline__ float vposmask(float y, float v) { return y > 0.f ? v : 0.f; }
// a relu's own backward step, in the reference's arithmetic: g * (y > 0) (ReluGradientOperator, neuron.cuh:173-178)
__device__ __forceinline__ float4 vrelugrad(float4 y, float4 g) {
    return make_float4(g.x * (y.x > 0.f), g.y * (y.y > 0.f), g.z * (y.z > 0.f), g.w * (y.w > 0.f));
}
__device__ __forceinline__ float vrelugrad(float y, float g) { return g * (y > 0.f); }
__device__ __forceinline__ float4 vfma_s(float s, float4 a, float4 c) { return make_float4(s * a.x + c.x, s * a.y + c.y, s * a.z + c.z, s * a.w + c.w); }
__device__ __forceinline__ float vfma_s(float s, float a, float c) { return s * a + c; }
// a^e for a > 0 as exp2(e * log2 a) on the hardware transcendental units -- what the reference's __powf compiles to
// (conv_util.cu:895, 1086).  The library powf is ~100 instructions an element and made the norm kernels ALU-bound
// (cmrnorm forward at 2.5 TB/s); denominators are >= k = 1 here, where this form is accurate to a few 1e-7 relative.
__device__ __forceinline__ float fast_pow(float a, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(a)); }
__device__ __forceinline__ float4 vpow(float4 a, float e) { return make_float4(fast_pow(a.x, e), fast_pow(a.y, e), fast_pow(a.z, e), fast_pow(a.w, e)); }
__device__ __forceinline__ float vpow(float a, float e) { return fast_pow(a, e); }

template <class T>
__device__ __forceinline__ T blend(T old, T v, float sT, float sO) { return vadd(vscale(old, sT), vscale(v, sO)); }

// Dropout in the pool's store (DROP): the pooled value goes to `target` as always (the max-undo routes by it), and
// (u < p ? 0 : value) -- u the counter-based uniform of the element, block (flat index / 4) of stream (seed, offset), exactly
// what mc_dropout_counter draws for it -- goes to dp.out; dp.keep gets one byte per four elements, bit e = (u_e >= p), which
// the backward pass reads instead of running Philox again for every window that overlaps a pixel block.
struct DropArgs {
    float* out;
    unsigned char* keep;
    float p;
    unsigned long long seed, offset;
};
__device__ __forceinline__ float4 drop4(float4 v, const DropArgs& dp, size_t flat, unsigned& bits) {
    float u[4];
    philox_uniform4(dp.seed, dp.offset, flat >> 2, u);
    bits = (u[0] < dp.p ? 0u : 1u) | (u[1] < dp.p ? 0u : 2u) | (u[2] < dp.p ? 0u : 4u) | (u[3] < dp.p ? 0u : 8u);
    return make_float4(u[0] < dp.p ? 0.f : v.x, u[1] < dp.p ? 0.f : v.y, u[2] < dp.p ? 0.f : v.z, u[3] < dp.p ? 0.f : v.w);
}
// gradient of four elements behind a dropout whose keep bits are `bits`
__device__ __forceinline__ float4 keep4(float4 g, unsigned bits) {
    return make_float4((bits & 1u) ? g.x : 0.f, (bits & 2u) ? g.y : 0.f, (bits & 4u) ? g.z : 0.f, (bits & 8u) ? g.w : 0.f);
}

// item -> (row = channel*pixels + px, image chunk)
template <int V>
__device__ __forceinline__ bool decode(size_t item, int nChunks, int& row, int& n) {
    row = (int)(item / nChunks);
    n = (int)(item - (size_t)row * nChunks) * V;
    return true;
}

// ------------------------------------------------------------------ pooling forward
template <int V, bool MAX>
__global__ void k_pool(const float* __restrict__ imgs, float* target, int N, int C, int imgSize, int subsX, int startX,
                       int strideX, int outputsX) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int numOut = outputsX * outputsX;
    const size_t total = (size_t)C * numOut * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / numOut, o = row - c * numOut;
        const int oy = o / outputsX, ox = o - oy * outputsX;
        const int sy = startX + oy * strideX, sx = startX + ox * strideX;
        const int y0 = max(0, sy), x0 = max(0, sx);
        const int y1 = min(imgSize, sy + subsX), x1 = min(imgSize, sx + subsX);
        typename VV::T acc = VV::splat(MAX ? -2e38f : 0.f);  // conv_util.cuh:78,89-91
        const float* src = imgs + (size_t)c * imgSize * imgSize * N + n;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const typename VV::T v = VV::ld(src + (size_t)(y * imgSize + x) * N);
                acc = MAX ? vmax(acc, v) : vadd(acc, v);
            }
        if (!MAX) acc = vdivs(acc, (float)((y1 - y0) * (x1 - x0)));  // clipped region size, conv_util.cuh:169,190
        VV::st(target + (size_t)row * N + n, acc);
    }
}

__device__ __forceinline__ void out_range(int p, int start, int subs, int stride, int outputsX, int& lo, int& hi) {
    lo = (p - start < subs) ? 0 : 1 + (p - start - subs) / stride;   // conv_util.cu:1517-1520
    hi = min(outputsX, 1 + (p - start) / stride);
}

// ------------------------------------------------------------------ max-pool undo
// the gradient of element(s) o behind a fused dropout: zero where the keep bit is clear (keep == nullptr: no dropout)
__device__ __forceinline__ float4 kept(float4 g, const unsigned char* keep, size_t o) { return keep ? keep4(g, keep[o >> 2]) : g; }
__device__ __forceinline__ float kept(float g, const unsigned char* keep, size_t o) { return (!keep || ((keep[o >> 2] >> (o & 3)) & 1)) ? g : 0.f; }
template <int V, bool RELU>
__global__ void k_max_undo(const float* __restrict__ imgs, const float* __restrict__ maxGrads,
                           const float* __restrict__ maxActs, float* target, int N, int C, int imgSize, int subsX,
                           int startX, int strideX, int outputsX, float sT, float sO, const unsigned char* __restrict__ keep, int gate) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize, numOut = outputsX * outputsX;
    const int lim = startX + strideX * (outputsX - 1) + subsX;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int y = px / imgSize, x = px - y * imgSize;
        typename VV::T prod = VV::splat(0.f);
        if (x >= startX && x < lim && y >= startX && y < lim) {  // conv_util.cu:1540-1541
            int ylo, yhi, xlo, xhi;
            out_range(y, startX, subsX, strideX, outputsX, ylo, yhi);
            out_range(x, startX, subsX, strideX, outputsX, xlo, xhi);
            const typename VV::T img = VV::ld(imgs + (size_t)row * N + n);
            for (int my = ylo; my < yhi; ++my)
                for (int mx = xlo; mx < xhi; ++mx) {
                    const size_t o = ((size_t)c * numOut + my * outputsX + mx) * N + n;
                    const typename VV::T ma = VV::ld(maxActs + o);
                    typename VV::T gv = kept(VV::ld(maxGrads + o), keep, o);
                    if (gate) gv = vrelugrad(ma, gv);   // a relu sits on the pooled tensor (applied in place): its gradient step, fused
                    prod = vadd(prod, veqmul(img, ma, gv));
                }
            if (RELU) prod = vposmask(img, prod);   // the pooled images are a relu's output: its gradient mask, fused
        }
        float* t = target + (size_t)row * N + n;
        VV::st(t, sT == 0.f ? vscale(prod, sO) : blend(VV::ld(t), prod, sT, sO));
    }
}

// ------------------------------------------------------------------ 2-D mapped variants (vector path)
// Same arithmetic as k_pool / k_max_undo, different work distribution: blockIdx.y = channel, a 256-thread
// workgroup = LX image-chunk lanes x (256 / LX) columns, and every thread walks a band of rows.  No per-item
// 64-bit divisions, x-side window bounds hoisted out of the row loop, 4-8 results per thread: the flat
// kernels spend more issue slots on index arithmetic than on their 9 loads.
struct Map2D {
    int lxShift;   // log2(LX), LX = lanes along the image-chunk axis (<= 32)
    int xBlocks;   // workgroups along x
    int band;      // rows per thread
    int xcd;       // workgroups of the launch / 8 when the XCD-aware order below is on, else 0
    int relu;      // k_pool_fix: a relu on the pooled value, applied in the store
};
// Workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest), each XCD with an L2 of its own.  The bands of one
// channel overlap (a 3x3 / 2 window row is read by two bands, a max-undo block row reads the window rows of its neighbours), so in
// launch order a channel's bands land in 8 different L2s and every shared row is fetched from memory once per band: measured on NIN
// pool1 (rocprofv3 FETCH_SIZE), forward 63 MB for 50 MB of input, backward 102 MB for 79 MB.  With the launch a multiple of 8
// workgroups, workgroup w stands in for logical workgroup (w % 8) * (total / 8) + w / 8: consecutive logical workgroups -- all the
// bands of a channel, then the next channel -- share an XCD.  (block, channel) of the logical workgroup:
__device__ __forceinline__ void map2d_block(const Map2D& mp, int& blk, int& c) {
    blk = blockIdx.x;
    c = blockIdx.y;
    if (mp.xcd) {
        const int w = blockIdx.y * gridDim.x + blockIdx.x;
        const int l = (w & 7) * mp.xcd + (w >> 3);
        c = l / (int)gridDim.x;
        blk = l - c * (int)gridDim.x;
    }
}

template <bool MAX>
__global__ void __launch_bounds__(256) k_pool_2d(const float* __restrict__ imgs, float* target, int N, int imgSize, int subsX,
                                                 int startX, int strideX, int outputsX, Map2D mp) {
    typedef Vec<4> VV;
    const int LX = 1 << mp.lxShift;
    const int nl = threadIdx.x & (LX - 1), oxl = threadIdx.x >> mp.lxShift;
    int blk, c;
    map2d_block(mp, blk, c);
    const int xb = blk % mp.xBlocks, yb = blk / mp.xBlocks;
    const int ox = xb * (256 >> mp.lxShift) + oxl;
    if (ox >= outputsX) return;
    const int sx = startX + ox * strideX;
    const int x0 = max(0, sx), x1 = min(imgSize, sx + subsX);
    const int numOut = outputsX * outputsX;
    const int oyEnd = min(outputsX, (yb + 1) * mp.band);
    const float* src = imgs + (size_t)c * imgSize * imgSize * N;
    float* dst = target + (size_t)c * numOut * N;
    for (int n = nl * 4; n < N; n += LX * 4)
        for (int oy = yb * mp.band; oy < oyEnd; ++oy) {
            const int sy = startX + oy * strideX;
            const int y0 = max(0, sy), y1 = min(imgSize, sy + subsX);
            float4 acc = VV::splat(MAX ? -2e38f : 0.f);
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const float4 v = VV::ld(src + (size_t)(y * imgSize + x) * N + n);
                    acc = MAX ? vmax(acc, v) : vadd(acc, v);
                }
            if (!MAX) acc = vdivs(acc, (float)((y1 - y0) * (x1 - x0)));
            VV::st(dst + (size_t)(oy * outputsX + ox) * N + n, acc);
        }
}

// k_pool_2d for a compile-time window (2x2, 3x3: every shipped strided pool): the window loop is unrolled and branch-free, so
// all SIZE^2 loads are in flight together.  A tap outside the image re-reads the nearest tap inside (max: a duplicate changes
// nothing) or enters the sum as +0 (avg: same order of the in-image terms as the loops of k_pool_2d).
template <bool MAX, int SIZE, bool DROP = false>
__global__ void __launch_bounds__(256) k_pool_fix(const float* __restrict__ imgs, float* __restrict__ target, int N, int imgSize,
                                                  int startX, int strideX, int outputsX, Map2D mp, DropArgs dp = DropArgs{}) {
    typedef Vec<4> VV;
    const int LX = 1 << mp.lxShift;
    const int nl = threadIdx.x & (LX - 1), oxl = threadIdx.x >> mp.lxShift;
    int blk, c;
    map2d_block(mp, blk, c);
    const int xb = blk % mp.xBlocks, yb = blk / mp.xBlocks;
    const int ox = xb * (256 >> mp.lxShift) + oxl;
    if (ox >= outputsX) return;
    const int sx = startX + ox * strideX;
    const int x0 = max(0, sx), x1 = min(imgSize, sx + SIZE);
    const int numOut = outputsX * outputsX;
    const int oyEnd = min(outputsX, (yb + 1) * mp.band);
    const float* src = imgs + (size_t)c * imgSize * imgSize * N;
    for (int n = nl * 4; n < N; n += LX * 4)
        for (int oy = yb * mp.band; oy < oyEnd; ++oy) {
            const int sy = startX + oy * strideX;
            const int y0 = max(0, sy), y1 = min(imgSize, sy + SIZE);
            float4 v[SIZE][SIZE];
#pragma unroll
            for (int dy = 0; dy < SIZE; ++dy)
#pragma unroll
                for (int dx = 0; dx < SIZE; ++dx) {
                    const int y = min(max(sy + dy, y0), y1 - 1), x = min(max(sx + dx, x0), x1 - 1);
                    v[dy][dx] = VV::ld(src + (size_t)(y * imgSize + x) * N + n);
                }
            float4 acc = VV::splat(MAX ? -2e38f : 0.f);
#pragma unroll
            for (int dy = 0; dy < SIZE; ++dy)
#pragma unroll
                for (int dx = 0; dx < SIZE; ++dx) {
                    if (MAX) acc = vmax(acc, v[dy][dx]);
                    else {
                        const bool in = sy + dy >= y0 && sy + dy < y1 && sx + dx >= x0 && sx + dx < x1;
                        acc = vadd(acc, in ? v[dy][dx] : VV::splat(0.f));
                    }
                }
            if (!MAX) acc = vdivs(acc, (float)((y1 - y0) * (x1 - x0)));
            const size_t flat = (size_t)c * numOut * N + (size_t)(oy * outputsX + ox) * N + n;
            if (!(mp.relu && dp.out == target)) VV::st(target + flat, acc);
            if (mp.relu) VV::st(dp.out + flat, vmax(acc, VV::splat(0.f)));   // the relu above the pool: its output (in place when dp.out == target)
            if constexpr (DROP) {
                unsigned bits;
                VV::st(dp.out + flat, drop4(acc, dp, flat, bits));
                dp.keep[flat >> 2] = (unsigned char)bits;
            }
        }
}

// the general form of the fused forward pass: dropout of an already pooled tensor, with the keep bytes (any n, any alignment)
__global__ void k_dropout_keep(const float* __restrict__ in, float* out, unsigned char* keep, size_t n, float p, unsigned long long seed,
                               unsigned long long offset) {
    const size_t nblk = (n + 3) / 4;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (size_t)gridDim.x * blockDim.x) {
        float u[4];
        philox_uniform4(seed, offset, b, u);
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (b * 4 + e < n) {
                const bool kp = !(u[e] < p);
                bits |= kp ? (1u << e) : 0u;
                out[b * 4 + e] = kp ? in[b * 4 + e] : 0.f;
            }
        keep[b] = (unsigned char)bits;
    }
}

template <bool RELU>
__global__ void __launch_bounds__(256) k_max_undo_2d(const float* __restrict__ imgs, const float* __restrict__ maxGrads,
                                                     const float* __restrict__ maxActs, float* target, int N, int imgSize, int subsX,
                                                     int startX, int strideX, int outputsX, float sT, float sO, Map2D mp,
                                                     const unsigned char* __restrict__ keep, int gate) {
    typedef Vec<4> VV;
    const int LX = 1 << mp.lxShift;
    const int nl = threadIdx.x & (LX - 1), xl = threadIdx.x >> mp.lxShift;
    int blk, c;
    map2d_block(mp, blk, c);
    const int xb = blk % mp.xBlocks, yb = blk / mp.xBlocks;
    const int x = xb * (256 >> mp.lxShift) + xl;
    if (x >= imgSize) return;
    const int numOut = outputsX * outputsX;
    const int lim = startX + strideX * (outputsX - 1) + subsX;
    const bool xin = x >= startX && x < lim;
    int xlo = 0, xhi = 0;
    if (xin) out_range(x, startX, subsX, strideX, outputsX, xlo, xhi);
    const int yEnd = min(imgSize, (yb + 1) * mp.band);
    const size_t ibase = (size_t)c * imgSize * imgSize * N, obase = (size_t)c * numOut * N;
    for (int n = nl * 4; n < N; n += LX * 4)
        for (int y = yb * mp.band; y < yEnd; ++y) {
            const size_t e = ibase + (size_t)(y * imgSize + x) * N + n;
            float4 prod = VV::splat(0.f);
            if (xin && y >= startX && y < lim) {
                int ylo, yhi;
                out_range(y, startX, subsX, strideX, outputsX, ylo, yhi);
                const float4 img = VV::ld(imgs + e);
                for (int my = ylo; my < yhi; ++my)
                    for (int mx = xlo; mx < xhi; ++mx) {
                        const size_t o = obase + (size_t)(my * outputsX + mx) * N + n;
                        const float4 ma = VV::ld(maxActs + o);
                        float4 gv = kept(VV::ld(maxGrads + o), keep, o);
                        if (gate) gv = vrelugrad(ma, gv);
                        prod = vadd(prod, veqmul(img, ma, gv));
                    }
                if (RELU) prod = vposmask(img, prod);
            }
            VV::st(target + e, sT == 0.f ? vscale(prod, sO) : blend(VV::ld(target + e), prod, sT, sO));
        }
}

// Stride-2 pooling (every shipped max pool: 3x3 / 2 and 2x2 / 2): a thread owns a 2x2 block of input pixels and walks block
// rows.  The block's pixels share their windows (at most 2 x 2 of them for 3x3 / 2), so each window's (maxActs, maxGrads) pair
// is loaded once per block instead of once per pixel: 4 image loads + <= 8 window loads + 4 stores per 4 pixels, against
// 4 x (1 + <= 8 + 1).  Per pixel the windows are visited in the same (my, mx) order as k_max_undo: same sum, same bits.
template <bool RELU>
__global__ void __launch_bounds__(256) k_max_undo_s2(const float* __restrict__ imgs, const float* __restrict__ maxGrads,
                                                     const float* __restrict__ maxActs, float* target, int N, int imgSize, int subsX,
                                                     int startX, int outputsX, float sT, float sO, Map2D mp,
                                                     const unsigned char* __restrict__ keep, int gate) {
    typedef Vec<4> VV;
    const int LX = 1 << mp.lxShift;
    const int nl = threadIdx.x & (LX - 1), xl = threadIdx.x >> mp.lxShift;
    int blk, c;
    map2d_block(mp, blk, c);
    const int xb = blk % mp.xBlocks, yb = blk / mp.xBlocks;
    const int bx = xb * (256 >> mp.lxShift) + xl, x0 = 2 * bx;
    if (x0 >= imgSize) return;
    const int numOut = outputsX * outputsX, blocksY = (imgSize + 1) >> 1;
    const int lim = startX + 2 * (outputsX - 1) + subsX;
    const bool x1in = x0 + 1 < imgSize;
    // windows that can hold pixel column x0 or x0 + 1 (empty when the block lies outside [startX, lim))
    int xlo = 0, xhi = 0;
    {
        const int a = max(x0, startX), b = min(x0 + 1, lim - 1);
        if (a <= b) { int t0, t1; out_range(a, startX, subsX, 2, outputsX, xlo, t1); out_range(b, startX, subsX, 2, outputsX, t0, xhi); }
    }
    const int byEnd = min(blocksY, (yb + 1) * mp.band);
    const size_t ibase = (size_t)c * imgSize * imgSize * N, obase = (size_t)c * numOut * N;
    for (int n = nl * 4; n < N; n += LX * 4)
        for (int by = yb * mp.band; by < byEnd; ++by) {
            const int y0 = 2 * by;
            const bool y1in = y0 + 1 < imgSize;
            const size_t e00 = ibase + (size_t)(y0 * imgSize + x0) * N + n, rowStep = (size_t)imgSize * N;
            const float4 z = VV::splat(0.f);
            const float4 i00 = VV::ld(imgs + e00), i01 = x1in ? VV::ld(imgs + e00 + N) : z;
            const float4 i10 = y1in ? VV::ld(imgs + e00 + rowStep) : z, i11 = (y1in && x1in) ? VV::ld(imgs + e00 + rowStep + N) : z;
            float4 p00 = z, p01 = z, p10 = z, p11 = z;
            int ylo = 0, yhi = 0;
            {
                const int a = max(y0, startX), b = min(y0 + 1, lim - 1);
                if (a <= b) { int t0, t1; out_range(a, startX, subsX, 2, outputsX, ylo, t1); out_range(b, startX, subsX, 2, outputsX, t0, yhi); }
            }
            // at most 2 x 2 windows touch the block (subsX <= 4, stride 2): all eight loads go out before the first use,
            // an absent window reads the channel's first output and contributes through a zero gradient
            const int ny = yhi - ylo, nx = xhi - xlo;
            float4 a[2][2], g[2][2];
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const bool ok = ky < ny && kx < nx;
                    const size_t o = obase + (ok ? (size_t)((ylo + ky) * outputsX + xlo + kx) * N : (size_t)0) + n;
                    a[ky][kx] = VV::ld(maxActs + o);
                    g[ky][kx] = kept(VV::ld(maxGrads + o), keep, o);
                    if (gate) g[ky][kx] = vrelugrad(a[ky][kx], g[ky][kx]);
                    if (!ok) g[ky][kx] = z;
                }
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
                const int wy = startX + 2 * (ylo + ky);
                const bool r0 = ky < ny && y0 >= wy && y0 < wy + subsX, r1 = ky < ny && y0 + 1 >= wy && y0 + 1 < wy + subsX;
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const int wx = startX + 2 * (xlo + kx);
                    const bool c0 = kx < nx && x0 >= wx && x0 < wx + subsX, c1 = kx < nx && x0 + 1 >= wx && x0 + 1 < wx + subsX;
                    if (r0 && c0) p00 = vadd(p00, veqmul(i00, a[ky][kx], g[ky][kx]));
                    if (r0 && c1) p01 = vadd(p01, veqmul(i01, a[ky][kx], g[ky][kx]));
                    if (r1 && c0) p10 = vadd(p10, veqmul(i10, a[ky][kx], g[ky][kx]));
                    if (r1 && c1) p11 = vadd(p11, veqmul(i11, a[ky][kx], g[ky][kx]));
                }
            }
            if (RELU) { p00 = vposmask(i00, p00); p01 = vposmask(i01, p01); p10 = vposmask(i10, p10); p11 = vposmask(i11, p11); }
            auto put = [&](size_t e, float4 p) { VV::st(target + e, sT == 0.f ? vscale(p, sO) : blend(VV::ld(target + e), p, sT, sO)); };
            put(e00, p00);
            if (x1in) put(e00 + N, p01);
            if (y1in) put(e00 + rowStep, p10);
            if (y1in && x1in) put(e00 + rowStep + N, p11);
        }
}

// rows: items along the walked axis, cols: items along x; channels on grid.y (<= 65535)
inline Map2D make_map2d(int N, int rows, int cols, int C, dim3& grid) {
    Map2D mp;
    const int chunks = N / 4;
    mp.lxShift = 0;
    while ((1 << mp.lxShift) < chunks && mp.lxShift < 5) ++mp.lxShift;
    const int colsPerBlock = 256 >> mp.lxShift;
    mp.xBlocks = ceil_div(cols, colsPerBlock);
    // rows per thread: 8 when that still leaves >= 4 workgroups per CU, else fewer
    mp.band = 8;
    while (mp.band > 1 && (long)mp.xBlocks * ceil_div(rows, mp.band) * C < 4L * num_cus()) mp.band >>= 1;
    grid = dim3(mp.xBlocks * ceil_div(rows, mp.band), C, 1);
    static const bool noXcd = getenv("MC_POOL_NO_XCD_ORDER") != nullptr;   // A/B switch
    const long total = (long)grid.x * grid.y;
    mp.xcd = (!noXcd && total % 8 == 0 && total < (1L << 30)) ? (int)(total / 8) : 0;
    mp.relu = 0;
    return mp;
}

// ------------------------------------------------------------------ large windows, few outputs (NIN's global 8x8 average)
// One workgroup per (channel, output pixel): LX image-chunk lanes x (256 / LX) window slices; slice s takes window
// elements s, s + S, ... and the S partial results are combined in slice order (deterministic).  The flat
// kernel gives each of a few hundred threads the whole window: 64 dependent round trips.
template <bool MAX>
__global__ void __launch_bounds__(256) k_pool_bigwin(const float* __restrict__ imgs, float* target, int N, int imgSize, int subsX,
                                                     int startX, int strideX, int outputsX, int lxShift) {
    typedef Vec<4> VV;
    __shared__ float4 part[256];
    const int LX = 1 << lxShift, S = 256 >> lxShift;
    const int nl = threadIdx.x & (LX - 1), sl = threadIdx.x >> lxShift;
    const int o = blockIdx.x, c = blockIdx.y;
    const int oy = o / outputsX, ox = o - oy * outputsX;
    const int sy = startX + oy * strideX, sx = startX + ox * strideX;
    const int y0 = max(0, sy), x0 = max(0, sx);
    const int y1 = min(imgSize, sy + subsX), x1 = min(imgSize, sx + subsX);
    const int wx = x1 - x0, cnt = (y1 - y0) * wx;
    const float* src = imgs + (size_t)c * imgSize * imgSize * N;
    for (int n0 = 0; n0 < N; n0 += LX * 4) {
        const int n = n0 + nl * 4;
        float4 acc = VV::splat(MAX ? -2e38f : 0.f);
        if (n < N)
            for (int e = sl; e < cnt; e += S) {
                const int dy = e / wx, dx = e - dy * wx;
                const float4 v = VV::ld(src + (size_t)((y0 + dy) * imgSize + x0 + dx) * N + n);
                acc = MAX ? vmax(acc, v) : vadd(acc, v);
            }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (sl == 0 && n < N) {
            for (int k = 1; k < S; ++k) {
                const float4 v = part[k * LX + nl];
                acc = MAX ? vmax(acc, v) : vadd(acc, v);
            }
            if (!MAX) acc = vdivs(acc, (float)cnt);
            VV::st(target + ((size_t)c * outputsX * outputsX + o) * N + n, acc);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ avg-pool undo
// reluActs (nullable): the pooled images are a relu's output -- its gradient mask goes into the store (sT == 0 only)
template <int V>
__global__ void k_avg_undo(const float* __restrict__ avgGrads, const float* __restrict__ reluActs, float* target, int N, int C, int imgSize,
                           int subsX, int startX, int strideX, int outputsX, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize, numOut = outputsX * outputsX;
    const int lim = startX + strideX * (outputsX - 1) + subsX;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int y = px / imgSize, x = px - y * imgSize;
        typename VV::T prod = VV::splat(0.f);
        if (x >= startX && x < lim && y >= startX && y < lim) {
            int ylo, yhi, xlo, xhi;
            out_range(y, startX, subsX, strideX, outputsX, ylo, yhi);
            out_range(x, startX, subsX, strideX, outputsX, xlo, xhi);
            for (int my = ylo; my < yhi; ++my) {
                const float ry = fminf((float)imgSize, (float)(startX + my * strideX + subsX)) - fmaxf(0.f, (float)(startX + my * strideX));
                for (int mx = xlo; mx < xhi; ++mx) {
                    const float rx = fminf((float)imgSize, (float)(startX + mx * strideX + subsX)) - fmaxf(0.f, (float)(startX + mx * strideX));
                    const float inv = 1.0f / (rx * ry);  // conv_util.cu:1442
                    prod = vadd(prod, vscale(VV::ld(avgGrads + ((size_t)c * numOut + my * outputsX + mx) * N + n), inv));
                }
            }
        }
        float* t = target + (size_t)row * N + n;
        if (reluActs) VV::st(t, vposmask(VV::ld(reluActs + (size_t)row * N + n), vscale(prod, sO)));
        else VV::st(t, sT == 0.f ? vscale(prod, sO) : blend(VV::ld(t), prod, sT, sO));
    }
}

// ------------------------------------------------------------------ within-map response / contrast norm
template <int V>
__global__ void k_cnorm(const float* __restrict__ imgs, const float* __restrict__ meanDiffs, float* denoms, float* target,
                        int N, int C, int imgSize, int sizeX, float addScale, float powScale) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int y = px / imgSize, x = px - y * imgSize;
        const int sy = -sizeX / 2 + y, sx = -sizeX / 2 + x;   // conv_util.cu:844-845
        const int y0 = max(0, sy), x0 = max(0, sx), y1 = min(imgSize, sy + sizeX), x1 = min(imgSize, sx + sizeX);
        typename VV::T s = VV::splat(0.f);
        const float* md = meanDiffs + (size_t)c * imgPix * N + n;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                const typename VV::T v = VV::ld(md + (size_t)(yy * imgSize + xx) * N);
                s = vadd(s, vmul(v, v));
            }
        const typename VV::T d = vfma_s(addScale, s, VV::splat(1.0f));  // 1 + addScale*sum, :889
        const size_t i = (size_t)row * N + n;
        VV::st(denoms + i, d);
        VV::st(target + i, vmul(VV::ld(imgs + i), vpow(d, -powScale)));
    }
}

// acts := scale * outGrads * acts / denoms   (kRNormUndoPrelims, conv_util.cu:1603-1615)
template <int V>
__global__ void k_rnorm_undo_prelims(float* acts, const float* __restrict__ denoms, const float* __restrict__ outGrads,
                                     size_t n, float scale) {
    typedef Vec<V> VV;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V)
        VV::st(acts + i, vdiv(vmul(vscale(VV::ld(outGrads + i), scale), VV::ld(acts + i)), VV::ld(denoms + i)));
}

template <int V>
__global__ void k_rnorm_undo(const float* __restrict__ outGrads, const float* __restrict__ denoms,
                             const float* __restrict__ inputs, const float* __restrict__ acts, float* target, int N, int C,
                             int imgSize, int sizeX, float powScale, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int y = px / imgSize, x = px - y * imgSize;
        const int y0 = max(0, y + sizeX / 2 - sizeX + 1), x0 = max(0, x + sizeX / 2 - sizeX + 1);  // :1653-1656
        const int y1 = min(imgSize, y + sizeX / 2 + 1), x1 = min(imgSize, x + sizeX / 2 + 1);
        typename VV::T s = VV::splat(0.f);
        const float* a = acts + (size_t)c * imgPix * N + n;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) s = vadd(s, VV::ld(a + (size_t)(yy * imgSize + xx) * N));
        const size_t i = (size_t)row * N + n;
        // inp * sum + out * den^-pow   (:1697-1701)
        const typename VV::T v = vadd(vmul(VV::ld(inputs + i), s), vmul(VV::ld(outGrads + i), vpow(VV::ld(denoms + i), -powScale)));
        VV::st(target + i, sT == 0.f ? vscale(v, sO) : blend(VV::ld(target + i), v, sT, sO));
    }
}

// ------------------------------------------------------------------ cross-map response norm
template <int V>
__global__ void k_cmrnorm(const float* __restrict__ imgs, float* denoms, float* target, int N, int C, int imgSize, int sizeF,
                          float addScale, float powScale, int blocked, float k) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int startF = blocked ? (c / sizeF) * sizeF : -sizeF / 2 + c;  // conv_util.cu:1070
        const int f0 = blocked ? startF : max(0, startF);
        const int f1 = min(C, startF + sizeF);
        typename VV::T s = VV::splat(0.f);
        for (int f = f0; f < f1; ++f) {
            const typename VV::T v = VV::ld(imgs + ((size_t)f * imgPix + px) * N + n);
            s = vadd(s, vmul(v, v));
        }
        const typename VV::T d = vfma_s(addScale, s, VV::splat(k));  // k + addScale*sum, :1085
        const size_t i = (size_t)row * N + n;
        VV::st(denoms + i, d);
        VV::st(target + i, vmul(VV::ld(imgs + i), vpow(d, -powScale)));
    }
}

// The same sums for the centred (not blocked) window of SF maps, walking the map axis: one lane owns a (pixel, image)
// column and a segment of maps, keeps the window's SF inputs in registers and re-adds their squares in map order at
// every step, so each input is read once (plus SF-1 halo maps per segment) instead of SF times, and the sum has the
// order -- and the bits -- of k_cmrnorm (maps outside [0, C) enter as +0).
template <int V, int SF>
__global__ void k_cmrnorm_walk(const float* __restrict__ imgs, float* __restrict__ denoms, float* __restrict__ target, int N, int C, int imgPix, int segLen,
                               float addScale, float powScale, float k) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    const int nChunks = N / V;
    const int nSeg = (C + segLen - 1) / segLen;
    const size_t total = (size_t)nSeg * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(item % nChunks) * V;
        const size_t rest = item / nChunks;
        const int px = (int)(rest % imgPix), seg = (int)(rest / imgPix);
        const int c0 = seg * segLen, c1 = min(C, c0 + segLen);
        const size_t col = (size_t)px * N + n, mapStride = (size_t)imgPix * N;
        T x[SF];
#pragma unroll
        for (int j = 0; j < SF - 1; ++j) {   // window of map c0, all but its last member
            const int f = c0 - SF / 2 + j;
            x[j + 1] = (f >= 0 && f < C) ? VV::ld(imgs + f * mapStride + col) : VV::splat(0.f);
        }
        auto fetch = [&](int f) { return f < C ? VV::ld(imgs + f * mapStride + col) : VV::splat(0.f); };
        T nxt = fetch(c0 - SF / 2 + SF - 1);
        for (int c = c0; c < c1; ++c) {
#pragma unroll
            for (int j = 0; j < SF - 1; ++j) x[j] = x[j + 1];
            x[SF - 1] = nxt;
            nxt = fetch(c + 1 - SF / 2 + SF - 1);   // requested a whole step ahead of its use
            T sum = VV::splat(0.f);
#pragma unroll
            for (int j = 0; j < SF; ++j) sum = vadd(sum, vmul(x[j], x[j]));
            const T d = vfma_s(addScale, sum, VV::splat(k));
            const size_t i = c * mapStride + col;
            VV::st(denoms + i, d);
            VV::st(target + i, vmul(x[SF / 2], vpow(d, -powScale)));
        }
    }
}

template <int V>
__global__ void k_cmrnorm_undo(const float* __restrict__ outGrads, const float* __restrict__ denoms,
                               const float* __restrict__ inputs, const float* __restrict__ acts, float* target, int N,
                               int C, int imgSize, int sizeF, float powScale, int blocked, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const int imgPix = imgSize * imgSize;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        int row, n;
        decode<V>(item, nChunks, row, n);
        const int c = row / imgPix, px = row - c * imgPix;
        const int startF = blocked ? (c / sizeF) * sizeF : -sizeF + sizeF / 2 + 1 + c;  // conv_util.cu:1144
        const int f0 = blocked ? startF : max(0, startF);
        const int f1 = min(C, startF + sizeF);
        typename VV::T s = VV::splat(0.f);
        for (int f = f0; f < f1; ++f) s = vadd(s, VV::ld(acts + ((size_t)f * imgPix + px) * N + n));
        const size_t i = (size_t)row * N + n;
        const typename VV::T v = vadd(vmul(VV::ld(inputs + i), s), vmul(VV::ld(outGrads + i), vpow(VV::ld(denoms + i), -powScale)));
        VV::st(target + i, sT == 0.f ? vscale(v, sO) : blend(VV::ld(target + i), v, sT, sO));
    }
}

// The undo for the centred window of SF maps in ONE pass, walking the map axis like k_cmrnorm_walk.  The reference runs two
// kernels: kRNormUndoPrelims overwrites acts with t = (-2 addScale pow) outGrads * acts / denoms, then kFRNormUndo sums t over
// the mirrored window (conv_util.cu:1603-1615, 1115-1181).  Here a lane owns a (pixel, image) column and a segment of maps,
// forms t of the map entering the window on the fly and keeps the window's SF terms in registers.  acts is not READ at all:
// the forward pass stored acts = inputs * denoms^-pow with this file's vpow, so the same expression gives the same bits (and
// denoms^-pow is needed for the second term anyway); that is what lets neighbouring segments recompute their halo terms
// while the owner overwrites acts (the reference's side effect, kept).  Same terms, same order of the window sum as the
// two-pass form (maps outside [0, C) enter as +0): bit-identical target and acts.  Traffic: 3 reads + 2 writes per element
// (+ (SF-1)/segLen halo) against 5 reads + 2 writes of the two passes.
template <int V, int SF>
__global__ void k_cmrnorm_undo_walk(const float* __restrict__ outGrads, const float* __restrict__ denoms, const float* __restrict__ inputs,
                                    float* __restrict__ acts, float* __restrict__ target, int N, int C, int imgPix, int segLen, float scale,
                                    float powScale, float sT, float sO) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    constexpr int LEAD = SF / 2;           // the window of map c is [c - (SF - 1 - LEAD), c + LEAD]
    constexpr int BACK = SF - 1 - LEAD;
    const int nChunks = N / V;
    const int nSeg = (C + segLen - 1) / segLen;
    const size_t total = (size_t)nSeg * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(item % nChunks) * V;
        const size_t rest = item / nChunks;
        const int px = (int)(rest % imgPix), seg = (int)(rest / imgPix);
        const int c0 = seg * segLen, c1 = min(C, c0 + segLen);
        const size_t col = (size_t)px * N + n, mapStride = (size_t)imgPix * N;
        T t[SF];            // t[j]: term of map c - BACK + j
        T xs[LEAD + 1];     // inputs of maps c .. c + LEAD
        T gp[LEAD + 1];     // outGrads * denoms^-pow of the same maps
        struct Raw { T g, d, x; };
        auto fetch = [&](int f) {   // a map outside [0, C) contributes a zero term (g = x = 0, d = 1)
            Raw r;
            if (f < 0 || f >= C) { r.g = VV::splat(0.f); r.d = VV::splat(1.f); r.x = VV::splat(0.f); return r; }
            const size_t i = f * mapStride + col;
            r.g = VV::ld(outGrads + i); r.d = VV::ld(denoms + i); r.x = VV::ld(inputs + i);
            return r;
        };
        // term of map f (and, for the centre queue, its input and grad * denom^-pow); own = this segment stores acts[f]
        auto take = [&](const Raw& r, int f, T& tf, T& xf, T& gpf, bool own) {
            const T p = vpow(r.d, -powScale);
            xf = r.x;
            tf = vdiv(vmul(vscale(r.g, scale), vmul(r.x, p)), r.d);
            gpf = vmul(r.g, p);
            if (own && f >= 0 && f < C) VV::st(acts + f * mapStride + col, tf);
        };
#pragma unroll
        for (int j = 0; j < SF - 1; ++j) {   // window of map c0 without its last member: maps c0 - BACK .. c0 + LEAD - 1
            const int f = c0 - BACK + j;
            T xf, gpf;
            take(fetch(f), f, t[j + 1], xf, gpf, f >= c0 && f < c1);
            if (j >= BACK) { xs[j - BACK + 1] = xf; gp[j - BACK + 1] = gpf; }
        }
        Raw nxt = fetch(c0 + LEAD);
        for (int c = c0; c < c1; ++c) {
            const Raw cur = nxt;
            nxt = fetch(c + 1 + LEAD);   // requested a whole step ahead of its use
#pragma unroll
            for (int j = 0; j < SF - 1; ++j) t[j] = t[j + 1];
#pragma unroll
            for (int j = 0; j < LEAD; ++j) { xs[j] = xs[j + 1]; gp[j] = gp[j + 1]; }
            const int f = c + LEAD;
            take(cur, f, t[SF - 1], xs[LEAD], gp[LEAD], f < c1);
            T sum = VV::splat(0.f);
#pragma unroll
            for (int j = 0; j < SF; ++j) sum = vadd(sum, t[j]);
            const T v = vadd(vmul(xs[0], sum), gp[0]);
            const size_t i = c * mapStride + col;
            VV::st(target + i, sT == 0.f ? vscale(v, sO) : blend(VV::ld(target + i), v, sT, sO));
        }
    }
}

inline int grid_items(size_t items) {
    const size_t want = (items + TPB - 1) / TPB;
    const size_t cap = (size_t)num_cus() * 16;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

inline bool use_vec(int N, std::initializer_list<const void*> ptrs) {
    if (N % 4) return false;
    for (const void* p : ptrs)
        if (!aligned16(p)) return false;
    return true;
}

}  // namespace
}  // namespace mc

using namespace mc;

#define MC_POOLARGS_OK(who)                                                                                      \
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && subsX > 0 && strideX > 0 && outputsX > 0,      \
               who ": non-positive dimension");                                                                  \
    MC_REQUIRE((size_t)numFilters * imgSize * imgSize * numImages < (1ull << 31), who ": tensor exceeds 32-bit indexing")

extern "C" {

int mc_conv_local_pool(const float* images, float* target, int numImages, int numFilters, int imgSize, int subsX,
                       int startX, int strideX, int outputsX, int pooler, mcStream_t s) {
    MC_REQUIRE(images && target, "mc_conv_local_pool: null pointer");
    MC_POOLARGS_OK("mc_conv_local_pool");
    MC_REQUIRE(pooler == MC_POOL_MAX || pooler == MC_POOL_AVG, "mc_conv_local_pool: unknown pooler %d", pooler);
    const bool v = use_vec(numImages, {images, target});
    if (v && numFilters <= 65535 && subsX * subsX >= 32 && (long)outputsX * outputsX * numFilters <= 8L * num_cus()) {
        int lx = 0;
        while ((1 << lx) < numImages / 4 && lx < 5) ++lx;
        dim3 g2(outputsX * outputsX, numFilters, 1);
        if (pooler == MC_POOL_MAX) hipLaunchKernelGGL(k_pool_bigwin<true>, g2, dim3(256), 0, S(s), images, target, numImages, imgSize, subsX, startX, strideX, outputsX, lx);
        else hipLaunchKernelGGL(k_pool_bigwin<false>, g2, dim3(256), 0, S(s), images, target, numImages, imgSize, subsX, startX, strideX, outputsX, lx);
        return launch_status("mc_conv_local_pool");
    }
    if (v && numFilters <= 65535 && (subsX == 2 || subsX == 3)) {
        dim3 g2;
        const Map2D mp = make_map2d(numImages, outputsX, outputsX, numFilters, g2);
#define MC_PF(MAXP, SZ) hipLaunchKernelGGL((k_pool_fix<MAXP, SZ>), g2, dim3(256), 0, S(s), images, target, numImages, imgSize, startX, strideX, outputsX, mp)
        if (pooler == MC_POOL_MAX) { if (subsX == 2) MC_PF(true, 2); else MC_PF(true, 3); }
        else { if (subsX == 2) MC_PF(false, 2); else MC_PF(false, 3); }
#undef MC_PF
        return launch_status("mc_conv_local_pool");
    }
    if (v && numFilters <= 65535) {
        dim3 g2;
        const Map2D mp = make_map2d(numImages, outputsX, outputsX, numFilters, g2);
        if (pooler == MC_POOL_MAX) hipLaunchKernelGGL(k_pool_2d<true>, g2, dim3(256), 0, S(s), images, target, numImages, imgSize, subsX, startX, strideX, outputsX, mp);
        else hipLaunchKernelGGL(k_pool_2d<false>, g2, dim3(256), 0, S(s), images, target, numImages, imgSize, subsX, startX, strideX, outputsX, mp);
        return launch_status("mc_conv_local_pool");
    }
    const size_t items = (size_t)numFilters * outputsX * outputsX * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
#define L(V, MAXP) hipLaunchKernelGGL((k_pool<V, MAXP>), g, b, 0, S(s), images, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX)
    if (pooler == MC_POOL_MAX) { if (v) L(4, true); else L(1, true); }
    else { if (v) L(4, false); else L(1, false); }
#undef L
    return launch_status("mc_conv_local_pool");
}

static int max_undo_entry(const char* who, bool relu, const float* images, const float* maxGrads, const float* maxActs, float* target,
                          int numImages, int numFilters, int imgSize, int subsX, int startX, int strideX, int outputsX,
                          float scaleTargets, float scaleOutput, mcStream_t s, const unsigned char* keep = nullptr, int gate = 0) {
    MC_REQUIRE(images && maxGrads && maxActs && target, "%s: null pointer", who);
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && subsX > 0 && strideX > 0 && outputsX > 0, "%s: non-positive dimension", who);
    MC_REQUIRE((size_t)numFilters * imgSize * imgSize * numImages < (1ull << 31), "%s: tensor exceeds 32-bit indexing", who);
    const bool v = use_vec(numImages, {images, maxGrads, maxActs, target});
    static const bool nos2 = getenv("MC_MAXUNDO_NOS2") != nullptr;
    // k_max_undo_s2 holds 2 x 2 windows per 2 x 2 pixel block.  Blocks start on even pixels, windows on startX + 2 o: a size-4
    // window with an odd start overlaps a block three ways per axis (starts x0 - 3, x0 - 1, x0 + 1) and takes the general kernel.
    const bool s2fits = subsX <= 3 || (subsX == 4 && (startX & 1) == 0);
    if (v && numFilters <= 65535 && strideX == 2 && s2fits && !nos2) {
        dim3 g2;
        const Map2D mp = make_map2d(numImages, (imgSize + 1) / 2, (imgSize + 1) / 2, numFilters, g2);
        if (relu) hipLaunchKernelGGL(k_max_undo_s2<true>, g2, dim3(256), 0, S(s), images, maxGrads, maxActs, target, numImages, imgSize, subsX, startX, outputsX, scaleTargets, scaleOutput, mp, keep, gate);
        else hipLaunchKernelGGL(k_max_undo_s2<false>, g2, dim3(256), 0, S(s), images, maxGrads, maxActs, target, numImages, imgSize, subsX, startX, outputsX, scaleTargets, scaleOutput, mp, keep, gate);
        return launch_status(who);
    }
    if (v && numFilters <= 65535) {
        dim3 g2;
        const Map2D mp = make_map2d(numImages, imgSize, imgSize, numFilters, g2);
        if (relu) hipLaunchKernelGGL(k_max_undo_2d<true>, g2, dim3(256), 0, S(s), images, maxGrads, maxActs, target, numImages, imgSize, subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, mp, keep, gate);
        else hipLaunchKernelGGL(k_max_undo_2d<false>, g2, dim3(256), 0, S(s), images, maxGrads, maxActs, target, numImages, imgSize, subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, mp, keep, gate);
        return launch_status(who);
    }
    const size_t items = (size_t)numFilters * imgSize * imgSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
#define MC_MU(V, R) hipLaunchKernelGGL((k_max_undo<V, R>), g, b, 0, S(s), images, maxGrads, maxActs, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, keep, gate)
    if (v && relu) MC_MU(4, true);
    else if (v) MC_MU(4, false);
    else if (relu) MC_MU(1, true);
    else MC_MU(1, false);
#undef MC_MU
    return launch_status(who);
}

int mc_conv_local_max_undo(const float* images, const float* maxGrads, const float* maxActs, float* target, int numImages,
                           int numFilters, int imgSize, int subsX, int startX, int strideX, int outputsX,
                           float scaleTargets, float scaleOutput, mcStream_t s) {
    return max_undo_entry("mc_conv_local_max_undo", false, images, maxGrads, maxActs, target, numImages, numFilters, imgSize, subsX,
                          startX, strideX, outputsX, scaleTargets, scaleOutput, s);
}

int mc_conv_local_max_undo_relu_grad(const float* images, const float* maxGrads, const float* maxActs, float* target,
                                     int numImages, int numFilters, int imgSize, int subsX, int startX, int strideX,
                                     int outputsX, float scaleTargets, float scaleOutput, mcStream_t s) {
    return max_undo_entry("mc_conv_local_max_undo_relu_grad", true, images, maxGrads, maxActs, target, numImages, numFilters, imgSize,
                          subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, s);
}

int mc_conv_local_max_undo_dropout(const float* images, const float* maxGrads, const float* maxActs, const unsigned char* keep, float* target,
                                   int numImages, int numFilters, int imgSize, int subsX, int startX, int strideX, int outputsX,
                                   float scaleTargets, float scaleOutput, int reluGrad, mcStream_t s) {
    MC_REQUIRE(keep, "mc_conv_local_max_undo_dropout: null keep bytes");
    return max_undo_entry("mc_conv_local_max_undo_dropout", reluGrad != 0, images, maxGrads, maxActs, target, numImages, numFilters, imgSize,
                          subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, s, keep);
}

int mc_conv_local_max_undo_relu_out(const float* images, const float* maxGrads, const float* maxActs, float* target, int numImages,
                                    int numFilters, int imgSize, int subsX, int startX, int strideX, int outputsX, float scaleTargets,
                                    float scaleOutput, int reluGrad, mcStream_t s) {
    return max_undo_entry("mc_conv_local_max_undo_relu_out", reluGrad != 0, images, maxGrads, maxActs, target, numImages, numFilters, imgSize,
                          subsX, startX, strideX, outputsX, scaleTargets, scaleOutput, s, nullptr, 1);
}

int mc_conv_local_pool_relu(const float* images, float* target, float* reluTarget, int numImages, int numFilters, int imgSize, int subsX,
                            int startX, int strideX, int outputsX, int pooler, mcStream_t s) {
    MC_REQUIRE(images && target && reluTarget, "mc_conv_local_pool_relu: null pointer");
    MC_POOLARGS_OK("mc_conv_local_pool_relu");
    MC_REQUIRE(pooler == MC_POOL_MAX || pooler == MC_POOL_AVG, "mc_conv_local_pool_relu: unknown pooler %d", pooler);
    if (use_vec(numImages, {images, target, reluTarget}) && numFilters <= 65535 && (subsX == 2 || subsX == 3)) {
        dim3 g2;
        Map2D mp = make_map2d(numImages, outputsX, outputsX, numFilters, g2);
        mp.relu = 1;
        const DropArgs dp{reluTarget, nullptr, 0.f, 0, 0};
#define MC_PFR(MAXP, SZ) hipLaunchKernelGGL((k_pool_fix<MAXP, SZ>), g2, dim3(256), 0, S(s), images, target, numImages, imgSize, startX, strideX, outputsX, mp, dp)
        if (pooler == MC_POOL_MAX) { if (subsX == 2) MC_PFR(true, 2); else MC_PFR(true, 3); }
        else { if (subsX == 2) MC_PFR(false, 2); else MC_PFR(false, 3); }
#undef MC_PFR
        return launch_status("mc_conv_local_pool_relu");
    }
    const int rc = mc_conv_local_pool(images, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX, pooler, s);
    if (rc) return rc;
    return mc_neuron_activate(MC_N_RELU, 0.f, 0.f, target, reluTarget, (size_t)numFilters * outputsX * outputsX * numImages, s);
}

size_t mc_dropout_keep_bytes(size_t n) { return (n + 3) / 4; }

int mc_conv_local_pool_dropout(const float* images, float* target, float* dropTarget, unsigned char* keep, int numImages, int numFilters,
                               int imgSize, int subsX, int startX, int strideX, int outputsX, int pooler, float prob,
                               unsigned long long seed, unsigned long long offset, mcStream_t s) {
    MC_REQUIRE(images && target && dropTarget && keep, "mc_conv_local_pool_dropout: null pointer");
    MC_POOLARGS_OK("mc_conv_local_pool_dropout");
    MC_REQUIRE(pooler == MC_POOL_MAX || pooler == MC_POOL_AVG, "mc_conv_local_pool_dropout: unknown pooler %d", pooler);
    MC_REQUIRE(prob > 0.f && prob < 1.f, "mc_conv_local_pool_dropout: prob must be in (0,1), got %g", (double)prob);
    MC_REQUIRE(offset % 4 == 0, "mc_conv_local_pool_dropout: offset must be a multiple of 4");
    MC_REQUIRE(dropTarget != target, "mc_conv_local_pool_dropout: the pooled and the dropped-out tensor must be two buffers (the max-undo routes by the pooled one)");
    const size_t n = (size_t)numFilters * outputsX * outputsX * numImages;
    static const bool nofuse = getenv("MC_POOL_DROPOUT_TWO_PASS") != nullptr;
    if (!nofuse && use_vec(numImages, {images, target, dropTarget}) && numFilters <= 65535 && (subsX == 2 || subsX == 3)) {
        dim3 g2;
        const Map2D mp = make_map2d(numImages, outputsX, outputsX, numFilters, g2);
        const DropArgs dp{dropTarget, keep, prob, seed, offset};
#define MC_PFD(MAXP, SZ) hipLaunchKernelGGL((k_pool_fix<MAXP, SZ, true>), g2, dim3(256), 0, S(s), images, target, numImages, imgSize, startX, strideX, outputsX, mp, dp)
        if (pooler == MC_POOL_MAX) { if (subsX == 2) MC_PFD(true, 2); else MC_PFD(true, 3); }
        else { if (subsX == 2) MC_PFD(false, 2); else MC_PFD(false, 3); }
#undef MC_PFD
        return launch_status("mc_conv_local_pool_dropout");
    }
    const int rc = mc_conv_local_pool(images, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX, pooler, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dropout_keep, dim3(grid_items((n + 3) / 4)), dim3(TPB), 0, S(s), target, dropTarget, keep, n, prob, seed, offset);
    return launch_status("mc_conv_local_pool_dropout");
}

static int avg_undo_entry(const char* who, const float* avgGrads, const float* reluActs, float* target, int numImages, int numFilters, int imgSize,
                          int subsX, int startX, int strideX, int outputsX, float scaleTargets, float scaleOutput, mcStream_t s);

int mc_conv_local_avg_undo_relu_grad(const float* avgGrads, const float* reluActs, float* target, int numImages, int numFilters, int imgSize,
                                     int subsX, int startX, int strideX, int outputsX, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(reluActs, "mc_conv_local_avg_undo_relu_grad: null pointer");
    return avg_undo_entry("mc_conv_local_avg_undo_relu_grad", avgGrads, reluActs, target, numImages, numFilters, imgSize, subsX, startX, strideX,
                          outputsX, 0.f, scaleOutput, s);
}

int mc_conv_local_avg_undo(const float* avgGrads, float* target, int numImages, int numFilters, int imgSize, int subsX,
                           int startX, int strideX, int outputsX, float scaleTargets, float scaleOutput, mcStream_t s) {
    return avg_undo_entry("mc_conv_local_avg_undo", avgGrads, nullptr, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX,
                          scaleTargets, scaleOutput, s);
}

static int avg_undo_entry(const char* who, const float* avgGrads, const float* reluActs, float* target, int numImages, int numFilters, int imgSize,
                          int subsX, int startX, int strideX, int outputsX, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(avgGrads && target, "%s: null pointer", who);
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && subsX > 0 && strideX > 0 && outputsX > 0, "%s: non-positive dimension", who);
    MC_REQUIRE((size_t)numFilters * imgSize * imgSize * numImages < (1ull << 31), "%s: tensor exceeds 32-bit indexing", who);
    const bool v = reluActs ? use_vec(numImages, {avgGrads, target, reluActs}) : use_vec(numImages, {avgGrads, target});
    const size_t items = (size_t)numFilters * imgSize * imgSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_avg_undo<4>, g, b, 0, S(s), avgGrads, reluActs, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_avg_undo<1>, g, b, 0, S(s), avgGrads, reluActs, target, numImages, numFilters, imgSize, subsX, startX, strideX, outputsX, scaleTargets, scaleOutput);
    return launch_status(who);
}

int mc_conv_contrast_norm(const float* images, const float* meanDiffs, float* denoms, float* target, int numImages,
                          int numFilters, int imgSize, int sizeX, float addScale, float powScale, mcStream_t s) {
    MC_REQUIRE(images && meanDiffs && denoms && target, "mc_conv_contrast_norm: null pointer");
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && sizeX > 0, "mc_conv_contrast_norm: non-positive dimension");
    const bool v = use_vec(numImages, {images, meanDiffs, denoms, target});
    const size_t items = (size_t)numFilters * imgSize * imgSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_cnorm<4>, g, b, 0, S(s), images, meanDiffs, denoms, target, numImages, numFilters, imgSize, sizeX, addScale, powScale);
    else hipLaunchKernelGGL(k_cnorm<1>, g, b, 0, S(s), images, meanDiffs, denoms, target, numImages, numFilters, imgSize, sizeX, addScale, powScale);
    return launch_status("mc_conv_contrast_norm");
}

int mc_conv_response_norm(const float* images, float* denoms, float* target, int numImages, int numFilters, int imgSize,
                          int sizeX, float addScale, float powScale, mcStream_t s) {
    // convResponseNorm passes the images as meanDiffs (conv_util.cu:2074-2076)
    return mc_conv_contrast_norm(images, images, denoms, target, numImages, numFilters, imgSize, sizeX, addScale, powScale, s);
}

static int undo_prelims(float* acts, const float* denoms, const float* outGrads, size_t n, float scale, bool v, hipStream_t st) {
    dim3 g(grid_items(n / (v ? 4 : 1))), b(TPB);
    if (v) hipLaunchKernelGGL(k_rnorm_undo_prelims<4>, g, b, 0, st, acts, denoms, outGrads, n, scale);
    else hipLaunchKernelGGL(k_rnorm_undo_prelims<1>, g, b, 0, st, acts, denoms, outGrads, n, scale);
    return launch_status("rnorm_undo_prelims");
}

int mc_conv_response_norm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts, float* target,
                               int numImages, int numFilters, int imgSize, int sizeX, float addScale, float powScale,
                               float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(outGrads && denoms && inputs && acts && target, "mc_conv_response_norm_undo: null pointer");
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && sizeX > 0, "mc_conv_response_norm_undo: non-positive dimension");
    const bool v = use_vec(numImages, {outGrads, denoms, inputs, acts, target});
    const size_t n = (size_t)numFilters * imgSize * imgSize * numImages;
    int rc = undo_prelims(acts, denoms, outGrads, n, -2 * addScale * powScale, v, S(s));
    if (rc) return rc;
    dim3 g(grid_items(n / (v ? 4 : 1))), b(TPB);
    if (v) hipLaunchKernelGGL(k_rnorm_undo<4>, g, b, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize, sizeX, powScale, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_rnorm_undo<1>, g, b, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize, sizeX, powScale, scaleTargets, scaleOutput);
    return launch_status("mc_conv_response_norm_undo");
}

int mc_conv_response_norm_cross_map(const float* images, float* denoms, float* target, int numImages, int numFilters,
                                    int imgSize, int sizeF, float addScale, float powScale, int blocked, float k,
                                    mcStream_t s) {
    MC_REQUIRE(images && denoms && target, "mc_conv_response_norm_cross_map: null pointer");
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && sizeF > 0 && sizeF <= numFilters, "mc_conv_response_norm_cross_map: bad dimension");
    const bool v = use_vec(numImages, {images, denoms, target});
    const size_t items = (size_t)numFilters * imgSize * imgSize * (numImages / (v ? 4 : 1));
    static const bool nowalk = getenv("MC_CMRNORM_NOWALK") != nullptr;
    if (!blocked && sizeF == 5 && numFilters >= 16 && !nowalk) {   // the centred 5-map window of the shipped configs
        const int segLen = 16, nSeg = ceil_div(numFilters, segLen);
        dim3 gw(grid_items((size_t)nSeg * imgSize * imgSize * (numImages / (v ? 4 : 1)))), bw(TPB);
        if (v) hipLaunchKernelGGL((k_cmrnorm_walk<4, 5>), gw, bw, 0, S(s), images, denoms, target, numImages, numFilters, imgSize * imgSize, segLen, addScale, powScale, k);
        else hipLaunchKernelGGL((k_cmrnorm_walk<1, 5>), gw, bw, 0, S(s), images, denoms, target, numImages, numFilters, imgSize * imgSize, segLen, addScale, powScale, k);
        return launch_status("mc_conv_response_norm_cross_map");
    }
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_cmrnorm<4>, g, b, 0, S(s), images, denoms, target, numImages, numFilters, imgSize, sizeF, addScale, powScale, blocked, k);
    else hipLaunchKernelGGL(k_cmrnorm<1>, g, b, 0, S(s), images, denoms, target, numImages, numFilters, imgSize, sizeF, addScale, powScale, blocked, k);
    return launch_status("mc_conv_response_norm_cross_map");
}

int mc_conv_response_norm_cross_map_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                                         float* target, int numImages, int numFilters, int imgSize, int sizeF,
                                         float addScale, float powScale, int blocked, float scaleTargets,
                                         float scaleOutput, mcStream_t s) {
    MC_REQUIRE(outGrads && denoms && inputs && acts && target, "mc_conv_response_norm_cross_map_undo: null pointer");
    MC_REQUIRE(numImages > 0 && numFilters > 0 && imgSize > 0 && sizeF > 0 && sizeF <= numFilters, "mc_conv_response_norm_cross_map_undo: bad dimension");
    const bool v = use_vec(numImages, {outGrads, denoms, inputs, acts, target});
    const size_t n = (size_t)numFilters * imgSize * imgSize * numImages;
    static const bool nowalk = getenv("MC_CMRNORM_NOWALK") != nullptr;
    if (!blocked && sizeF == 5 && numFilters >= 16 && !nowalk) {   // the centred 5-map window of the shipped configs: one pass
        const int segLen = 16, nSeg = ceil_div(numFilters, segLen);
        dim3 gw(grid_items((size_t)nSeg * imgSize * imgSize * (numImages / (v ? 4 : 1)))), bw(TPB);
        if (v) hipLaunchKernelGGL((k_cmrnorm_undo_walk<4, 5>), gw, bw, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize * imgSize, segLen, -2 * addScale * powScale, powScale, scaleTargets, scaleOutput);
        else hipLaunchKernelGGL((k_cmrnorm_undo_walk<1, 5>), gw, bw, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize * imgSize, segLen, -2 * addScale * powScale, powScale, scaleTargets, scaleOutput);
        return launch_status("mc_conv_response_norm_cross_map_undo");
    }
    int rc = undo_prelims(acts, denoms, outGrads, n, -2 * addScale * powScale, v, S(s));
    if (rc) return rc;
    dim3 g(grid_items(n / (v ? 4 : 1))), b(TPB);
    if (v) hipLaunchKernelGGL(k_cmrnorm_undo<4>, g, b, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize, sizeF, powScale, blocked, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_cmrnorm_undo<1>, g, b, 0, S(s), outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSize, sizeF, powScale, blocked, scaleTargets, scaleOutput);
    return launch_status("mc_conv_response_norm_cross_map_undo");
}

}  // extern "C"
