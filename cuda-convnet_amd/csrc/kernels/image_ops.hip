// image_ops.hip -- the image-preprocessing and element-wise-max layers' kernels for gfx950:
// bed of nails (+undo), separable Gaussian blur, bilinear resize, RGB->YUV, RGB->L*a*b*, element-wise max (+grad).
// All HBM-bound.  Tensors are (channels, pixels, numImages) with the image index innermost, so one lane owns 4
// consecutive images (one 16-byte access) of one output item and a wave covers 256 images of it.
// Semantics follow Kernel/src/conv/conv_util.cu (kRGBToYUV :172, kRGBToLAB :237, kResizeBilinear :290,
// kGaussianBlur :358, kBedOfNails :463) and Kernel/src/layers/layer_kernels.cu (kEltwiseMaxGrad :155).
#include "mc_common.h"

namespace mc {
namespace {

constexpr int TPB = 256;

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

// lane-wise helpers over float / float4
template <class F> __device__ __forceinline__ float4 map1(float4 a, F f) { return make_float4(f(a.x), f(a.y), f(a.z), f(a.w)); }
template <class F> __device__ __forceinline__ float map1(float a, F f) { return f(a); }
template <class F> __device__ __forceinline__ float4 map2(float4 a, float4 b, F f) { return make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w)); }
template <class F> __device__ __forceinline__ float map2(float a, float b, F f) { return f(a, b); }
template <class F> __device__ __forceinline__ float4 map3(float4 a, float4 b, float4 c, F f) {
    return make_float4(f(a.x, b.x, c.x), f(a.y, b.y, c.y), f(a.z, b.z, c.z), f(a.w, b.w, c.w));
}
template <class F> __device__ __forceinline__ float map3(float a, float b, float c, F f) { return f(a, b, c); }

template <class T> __device__ __forceinline__ T vfma(float s, T a, T c) {
    return map2(a, c, [s](float x, float y) { return fmaf(s, x, y); });
}
template <class T> __device__ __forceinline__ T vscale(T a, float s) {
    return map1(a, [s](float x) { return x * s; });
}
// scaleTargets * old + scaleOutput * v (old is read only when scaleTargets != 0)
template <int V> __device__ __forceinline__ void store_blend(float* t, typename Vec<V>::T v, float sT, float sO) {
    typedef Vec<V> VV;
    if (sT == 0.f) VV::st(t, vscale(v, sO));
    else VV::st(t, map2(VV::ld(t), v, [sT, sO](float o, float x) { return sT * o + sO * x; }));
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

// ------------------------------------------------------------------ bed of nails (kBedOfNails, conv_util.cu:463-537)
// forward: out[c, oy, ox, n] = sT*out + sO*img[c, start + oy*stride, start + ox*stride, n]
template <int V>
__global__ void k_bed_of_nails(const float* __restrict__ imgs, float* target, int N, int C, int imgSize, int startX, int strideX,
                               int outputsX, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V, numOut = outputsX * outputsX;
    const size_t total = (size_t)C * numOut * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int c = row / numOut, o = row - c * numOut;
        const int oy = o / outputsX, ox = o - oy * outputsX;
        const int px = (startX + oy * strideX) * imgSize + startX + ox * strideX;
        store_blend<V>(target + (size_t)row * N + n, VV::ld(imgs + ((size_t)c * imgSize * imgSize + px) * N + n), sT, sO);
    }
}

// undo: the gradient of a nail goes back to its pixel.  scaleTargets == 0: every other pixel of the target
// becomes 0 (the reference zeroes the matrix first, conv_util.cu:563-565); otherwise ONLY the nail pixels are
// touched (t = sT*t + sO*g) and the rest keeps its value unscaled, as the reference kernel does.
template <int V>
__global__ void k_bed_of_nails_undo(const float* __restrict__ actsGrad, float* target, int N, int C, int imgSize, int startX,
                                    int strideX, int outputsX, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V, imgPix = imgSize * imgSize, numOut = outputsX * outputsX;
    const size_t total = (size_t)C * imgPix * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int c = row / imgPix, px = row - c * imgPix;
        const int y = px / imgSize, x = px - y * imgSize;
        const int dy = y - startX, dx = x - startX;
        const int oy = dy / strideX, ox = dx / strideX;
        const bool nail = dy >= 0 && dx >= 0 && oy * strideX == dy && ox * strideX == dx && oy < outputsX && ox < outputsX;
        float* t = target + (size_t)row * N + n;
        if (nail) store_blend<V>(t, VV::ld(actsGrad + ((size_t)c * numOut + oy * outputsX + ox) * N + n), sT, sO);
        else if (sT == 0.f) VV::st(t, VV::splat(0.f));
    }
}

// ------------------------------------------------------------------ Gaussian blur, one direction (kGaussianBlur, conv_util.cu:358-443)
// out[x] = sum_{d=-R..R} in[x + d] * f[R + d], zero outside the line, terms added in ascending x + d.
// One lane walks one line (image row when horiz, column otherwise) of 4 images with the 2R+1 inputs of the
// window in registers, so every input is read once and -- as in the reference -- target may alias imgs: the
// lane has loaded in[0 .. x+R] before it writes out[x], and no other lane touches its line.
template <int V, int R>
__global__ void k_gaussian_blur(const float* imgs, const float* __restrict__ filter, float* target, int N, int C, int imgSize,
                                int horiz, float sT, float sO) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    constexpr int FW = 2 * R + 1;
    float f[FW];
#pragma unroll
    for (int j = 0; j < FW; ++j) f[j] = filter[j];
    const int nChunks = N / V;
    const size_t total = (size_t)C * imgSize * nChunks;
    const size_t step = horiz ? (size_t)N : (size_t)N * imgSize;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int line = (int)(item / nChunks), n = (int)(item - (size_t)line * nChunks) * V;
        const int c = line / imgSize, r = line - c * imgSize;
        const size_t base = (size_t)c * imgSize * imgSize * N + (horiz ? (size_t)r * imgSize * N : (size_t)r * N) + n;
        const float* in = imgs + base;
        float* out = target + base;
        T win[FW];   // before the shift of step x: win[k] = in[x - 1 - R + k]
#pragma unroll
        for (int k = 0; k < FW; ++k) win[k] = VV::splat(0.f);
#pragma unroll
        for (int j = 0; j < R; ++j) win[R + 1 + j] = VV::ld(in + j * step);
        for (int x = 0; x < imgSize; ++x) {
#pragma unroll
            for (int k = 0; k + 1 < FW; ++k) win[k] = win[k + 1];
            win[FW - 1] = x + R < imgSize ? VV::ld(in + (size_t)(x + R) * step) : VV::splat(0.f);
            T acc = VV::splat(0.f);
#pragma unroll
            for (int k = 0; k < FW; ++k) acc = vfma(f[k], win[k], acc);
            store_blend<V>(out + (size_t)x * step, acc, sT, sO);
        }
    }
}

// ------------------------------------------------------------------ bilinear resize (kResizeBilinear, conv_util.cu:290-341)
template <int V>
__global__ void k_resize_bilinear(const float* __restrict__ imgs, float* target, int N, int C, int imgSize, int tgtSize,
                                  float scale, float centerScale) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    const int nChunks = N / V, tgtPix = tgtSize * tgtSize;
    const size_t total = (size_t)C * tgtPix * nChunks;
    const float hi = (float)imgSize - 1.01f;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int c = row / tgtPix, px = row - c * tgtPix;
        const int pxY = px / tgtSize, pxX = px - pxY * tgtSize;
        const float srcX = fmaxf(0.f, fminf(hi, fmaf((float)pxX, scale, centerScale)));
        const float srcY = fmaxf(0.f, fminf(hi, fmaf((float)pxY, scale, centerScale)));
        const float u = floorf(srcX + 1.f) - srcX;
        const float w = srcY - floorf(srcY);
        const int p0 = (int)floorf(srcY) * imgSize + (int)floorf(srcX);   // top-left; srcX,Y <= imgSize - 1.01 keeps +1 inside
        const float* s = imgs + ((size_t)c * imgSize * imgSize + p0) * N + n;
        const T v0 = VV::ld(s), v1 = VV::ld(s + N), v2 = VV::ld(s + (size_t)imgSize * N), v3 = VV::ld(s + (size_t)(imgSize + 1) * N);
        const T c0 = map2(v0, v1, [u](float a, float b) { return fmaf(u, a - b, b); });
        const T c1 = map2(v2, v3, [u](float a, float b) { return fmaf(u, a - b, b); });
        VV::st(target + (size_t)row * N + n, map2(c0, c1, [w](float a, float b) { return fmaf(w, b - a, a); }));
    }
}

// ------------------------------------------------------------------ colour transforms (kRGBToYUV :172-194, kRGBToLAB :237-268)
__device__ __forceinline__ float labf(float x) {   // conv_util.cu:196-201
    return x > 0.0088564517f ? powf(x, 0.3333f) : 7.787037f * x + 0.13793103f;
}

template <int V, int MODE>   // MODE 0: YUV, 1: LAB, 2: LAB centred
__global__ void k_rgb_transform(const float* __restrict__ imgs, float* target, int N, int imgPix) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    const int nChunks = N / V;
    const size_t total = (size_t)imgPix * nChunks, chan = (size_t)imgPix * N;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(item / nChunks), n = (int)(item - (size_t)px * nChunks) * V;
        const size_t o = (size_t)px * N + n;
        const T R = VV::ld(imgs + o), G = VV::ld(imgs + chan + o), B = VV::ld(imgs + 2 * chan + o);
        T o0, o1, o2;
        if (MODE == 0) {
            o0 = map3(R, G, B, [](float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; });
            o1 = map3(R, G, B, [](float r, float g, float b) { return -0.09991f * r + -0.33609f * g + 0.436f * b; });
            o2 = map3(R, G, B, [](float r, float g, float b) { return 0.615f * r + -0.55861f * g + -0.05639f * b; });
        } else {
            const T lx = map3(R, G, B, [](float r, float g, float b) { return labf(0.49f * r + 0.31f * g + 0.2f * b); });
            const T ly = map3(R, G, B, [](float r, float g, float b) { return labf(0.17697f * r + 0.8124f * g + 0.01063f * b); });
            const T lz = map3(R, G, B, [](float r, float g, float b) { return labf(0.01f * g + 0.99f * b); });
            o0 = map1(ly, [](float y) { return 116.0f * y - 16.0f - (MODE == 2 ? 50.0f : 0.f); });
            o1 = map2(lx, ly, [](float x, float y) { return 500.0f * (x - y); });
            o2 = map2(ly, lz, [](float y, float z) { return 200.0f * (y - z); });
        }
        VV::st(target + o, o0);
        VV::st(target + chan + o, o1);
        VV::st(target + 2 * chan + o, o2);
    }
}

// ------------------------------------------------------------------ element-wise max (+ gradient routing, layer_kernels.cu:155-164)
template <int V>
__global__ void k_eltwise_max(const float* a, const float* b, float* out, size_t n) {
    typedef Vec<V> VV;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V)
        VV::st(out + i, map2(VV::ld(a + i), VV::ld(b + i), [](float x, float y) { return fmaxf(x, y); }));
}

template <int V, bool ADD>
__global__ void k_eltwise_max_grad(const float* __restrict__ actGrad, const float* __restrict__ input, const float* __restrict__ output,
                                   float* target, size_t n) {
    typedef Vec<V> VV;
    typedef typename VV::T T;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
        T g = map3(VV::ld(actGrad + i), VV::ld(input + i), VV::ld(output + i), [](float gr, float in, float o) { return gr * (o == in); });
        if (ADD) g = map2(VV::ld(target + i), g, [](float t, float x) { return t + x; });
        VV::st(target + i, g);
    }
}


// ------------------------------------------------------------------ group sparsity over same-label cases (plugin cost.gsinlabel)
// PluginsSrc/src/extrautils.cu: CalculateSqrtSumSquare :45-107 + MergeTemp :116-123, kCalculateGradient :167-199.
// S[ch, j] = sqrt(count_j * sum_{n: label_n = value_j} sum_p acts[ch, p, n]^2); cost = sum S; the gradient of case n
// is count * acts / (S[ch, label_n] + 2.5e-7).  Instead of the reference's sorted histogram of label values
// (thrust, :4-31) each case finds its group by comparing labels, so S is kept PER CASE (sqrts[ch, n]).
__global__ void k_gs_pixel_sumsq(const float* __restrict__ acts, float* q, int N, int channels, int P) {
    const int total = channels * N;
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        const int ch = item / N, n = item - ch * N;
        const float* a = acts + (size_t)ch * P * N + n;
        float s = 0.f;
        for (int p = 0; p < P; ++p) {
            const float v = a[(size_t)p * N];
            s = fmaf(v, v, s);
        }
        q[item] = s;
    }
}

// sqrts[ch, n] = S of n's label group; counts[n] = size of the group; costTerms[ch, n] = S for the first case of each
// group and 0 for the others, so that sum(costTerms) is the cost with every group counted once.
__global__ void k_gs_groups(const float* __restrict__ labels, const float* __restrict__ q, float* sqrts, float* counts, float* costTerms,
                            int N, int channels) {
    const int total = channels * N;
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        const int ch = item / N, n = item - ch * N;
        const float mine = labels[n];
        float t = 0.f;
        int c = 0;
        bool first = true;
        for (int m = 0; m < N; ++m)
            if (labels[m] == mine) {
                t += q[ch * N + m];
                ++c;
                first = first && m >= n;
            }
        const float sq = sqrtf((float)c * t);
        sqrts[item] = sq;
        costTerms[item] = first ? sq : 0.f;
        if (ch == 0) counts[n] = (float)c;
    }
}

template <int V>
__global__ void k_gs_grad(const float* __restrict__ acts, const float* __restrict__ sqrts, const float* __restrict__ counts, float* target,
                          int N, int channels, int P, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const size_t total = (size_t)channels * P * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int ch = row / P;
        const typename VV::T g = map3(VV::ld(acts + (size_t)row * N + n), VV::ld(sqrts + (size_t)ch * N + n), VV::ld(counts + n),
                                      [](float a, float s, float c) { return c * a / (s + 2.5e-7f); });
        store_blend<V>(target + (size_t)row * N + n, g, sT, sO);
    }
}


// ------------------------------------------------------------------ channel gather / scatter (random sparse connectivity)
// The reference's *Sparse conv kernels (filter_acts.cu:259-413, img_acts.cu:266-470, weight_acts.cu:256-409) read
// channel colorIndices[g*filterChannels + c] where the dense ones read channel g*filterChannels + c.  Here the
// connected channels are gathered into a dense (groups*filterChannels)-channel tensor, the dense grouped
// contraction runs on it (MFMA engine, unchanged), and the backward-data result is summed back per real channel.
template <int V>
__global__ void k_gather_channels(const float* __restrict__ src, const int* __restrict__ idx, float* dst, int N, int K, int P, int numSrc) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const size_t total = (size_t)K * P * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int k = row / P, p = row - k * P;
        const int c = idx[k];
        VV::st(dst + (size_t)row * N + n, (unsigned)c < (unsigned)numSrc ? VV::ld(src + ((size_t)c * P + p) * N + n) : VV::splat(0.f));
    }
}

// dst[c] = sT*dst[c] + sO * sum_{o < O} src[inv[c*O + o]], terms added in ascending o
template <int V>
__global__ void k_scatter_channels(const float* __restrict__ src, const int* __restrict__ inv, float* dst, int N, int C, int O, int P,
                                   int numSrc, float sT, float sO) {
    typedef Vec<V> VV;
    const int nChunks = N / V;
    const size_t total = (size_t)C * P * nChunks;
    for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(item / nChunks), n = (int)(item - (size_t)row * nChunks) * V;
        const int c = row / P, p = row - c * P;
        typename VV::T acc = VV::splat(0.f);
        for (int o = 0; o < O; ++o) {
            const int k = inv[c * O + o];
            if ((unsigned)k < (unsigned)numSrc)
                acc = map2(acc, VV::ld(src + ((size_t)k * P + p) * N + n), [](float a, float b) { return a + b; });
        }
        store_blend<V>(dst + (size_t)row * N + n, acc, sT, sO);
    }
}

}  // namespace
}  // namespace mc

using namespace mc;

#define MC_IMG_DIMS_OK(who, chans, pixelsIn)                                                                              \
    MC_REQUIRE(numImages > 0 && (chans) > 0 && imgSize > 0, "%s: non-positive dimension", who);                          \
    MC_REQUIRE((size_t)(chans) * (pixelsIn) * numImages < (1ull << 31), "%s: tensor exceeds 32-bit indexing", who)

extern "C" {

static int nails_ok(const char* who, int imgSize, int startX, int strideX, int* outputsX) {
    MC_REQUIRE(strideX > 1, "%s: stride %d must be greater than 1 (conv_util.cu:552)", who, strideX);
    *outputsX = (imgSize + strideX - 1) / strideX;   // DIVUP(imgSize, strideX), conv_util.cu:554
    MC_REQUIRE(startX >= 0 && startX + (*outputsX - 1) * strideX < imgSize, "%s: nail (%d + %d*%d) outside the %d-pixel image", who,
               startX, *outputsX - 1, strideX, imgSize);
    return MC_OK;
}

int mc_conv_bed_of_nails(const float* images, float* target, int numImages, int numChannels, int imgSize, int startX,
                         int strideX, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(images && target, "mc_conv_bed_of_nails: null pointer");
    MC_IMG_DIMS_OK("mc_conv_bed_of_nails", numChannels, imgSize * imgSize);
    int outputsX;
    if (int rc = nails_ok("mc_conv_bed_of_nails", imgSize, startX, strideX, &outputsX)) return rc;
    const bool v = use_vec(numImages, {images, target});
    const size_t items = (size_t)numChannels * outputsX * outputsX * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_bed_of_nails<4>, g, b, 0, S(s), images, target, numImages, numChannels, imgSize, startX, strideX, outputsX, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_bed_of_nails<1>, g, b, 0, S(s), images, target, numImages, numChannels, imgSize, startX, strideX, outputsX, scaleTargets, scaleOutput);
    return launch_status("mc_conv_bed_of_nails");
}

int mc_conv_bed_of_nails_undo(const float* actsGrad, float* target, int numImages, int numChannels, int imgSize, int startX,
                              int strideX, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(actsGrad && target, "mc_conv_bed_of_nails_undo: null pointer");
    MC_IMG_DIMS_OK("mc_conv_bed_of_nails_undo", numChannels, imgSize * imgSize);
    int outputsX;
    if (int rc = nails_ok("mc_conv_bed_of_nails_undo", imgSize, startX, strideX, &outputsX)) return rc;
    const bool v = use_vec(numImages, {actsGrad, target});
    const size_t items = (size_t)numChannels * imgSize * imgSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_bed_of_nails_undo<4>, g, b, 0, S(s), actsGrad, target, numImages, numChannels, imgSize, startX, strideX, outputsX, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_bed_of_nails_undo<1>, g, b, 0, S(s), actsGrad, target, numImages, numChannels, imgSize, startX, strideX, outputsX, scaleTargets, scaleOutput);
    return launch_status("mc_conv_bed_of_nails_undo");
}

int mc_conv_gaussian_blur(const float* images, const float* filter, float* target, int numImages, int numChannels, int imgSize,
                          int radius, int horiz, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(images && filter && target, "mc_conv_gaussian_blur: null pointer");
    MC_IMG_DIMS_OK("mc_conv_gaussian_blur", numChannels, imgSize * imgSize);
    MC_REQUIRE(radius >= 1 && radius <= 4, "mc_conv_gaussian_blur: radius %d not in 1..4 (conv_util.cu:692)", radius);
    MC_REQUIRE(imgSize >= 2 * radius + 1, "mc_conv_gaussian_blur: image size %d smaller than the %d-tap filter (conv_util.cu:693)", imgSize, 2 * radius + 1);
    const bool v = use_vec(numImages, {images, target});
    const size_t items = (size_t)numChannels * imgSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
#define MC_GB(V, R) hipLaunchKernelGGL((k_gaussian_blur<V, R>), g, b, 0, S(s), images, filter, target, numImages, numChannels, imgSize, horiz, scaleTargets, scaleOutput)
#define MC_GBV(R) do { if (v) MC_GB(4, R); else MC_GB(1, R); } while (0)
    switch (radius) {
        case 1: MC_GBV(1); break;
        case 2: MC_GBV(2); break;
        case 3: MC_GBV(3); break;
        default: MC_GBV(4); break;
    }
#undef MC_GBV
#undef MC_GB
    return launch_status("mc_conv_gaussian_blur");
}

int mc_conv_resize_bilinear(const float* images, float* target, int numImages, int numChannels, int imgSize, int tgtSize,
                            float scale, mcStream_t s) {
    MC_REQUIRE(images && target, "mc_conv_resize_bilinear: null pointer");
    MC_REQUIRE(images != target, "mc_conv_resize_bilinear: target must not alias images");
    MC_IMG_DIMS_OK("mc_conv_resize_bilinear", numChannels, imgSize * imgSize);
    MC_REQUIRE(imgSize >= 2 && tgtSize > 0, "mc_conv_resize_bilinear: image size %d (>= 2) / target size %d (> 0)", imgSize, tgtSize);
    MC_REQUIRE((size_t)numChannels * tgtSize * tgtSize * numImages < (1ull << 31), "mc_conv_resize_bilinear: target exceeds 32-bit indexing");
    const double centerScale = imgSize * 0.5 - tgtSize * 0.5 * scale;   // conv_util.cu:2439-2441
    const bool v = use_vec(numImages, {images, target});
    const size_t items = (size_t)numChannels * tgtSize * tgtSize * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
    if (v) hipLaunchKernelGGL(k_resize_bilinear<4>, g, b, 0, S(s), images, target, numImages, numChannels, imgSize, tgtSize, scale, (float)centerScale);
    else hipLaunchKernelGGL(k_resize_bilinear<1>, g, b, 0, S(s), images, target, numImages, numChannels, imgSize, tgtSize, scale, (float)centerScale);
    return launch_status("mc_conv_resize_bilinear");
}

static int rgb_entry(const char* who, int mode, const float* images, float* target, int numImages, int imgPixels, mcStream_t s) {
    MC_REQUIRE(images && target, "%s: null pointer", who);
    MC_REQUIRE(numImages > 0 && imgPixels > 0, "%s: non-positive dimension", who);
    MC_REQUIRE((size_t)3 * imgPixels * numImages < (1ull << 31), "%s: tensor exceeds 32-bit indexing", who);
    const bool v = use_vec(numImages, {images, target});
    const size_t items = (size_t)imgPixels * (numImages / (v ? 4 : 1));
    dim3 g(grid_items(items)), b(TPB);
#define MC_RGB(V, M) hipLaunchKernelGGL((k_rgb_transform<V, M>), g, b, 0, S(s), images, target, numImages, imgPixels)
    if (mode == 0) { if (v) MC_RGB(4, 0); else MC_RGB(1, 0); }
    else if (mode == 1) { if (v) MC_RGB(4, 1); else MC_RGB(1, 1); }
    else { if (v) MC_RGB(4, 2); else MC_RGB(1, 2); }
#undef MC_RGB
    return launch_status(who);
}

int mc_conv_rgb_to_yuv(const float* images, float* target, int numImages, int imgPixels, mcStream_t s) {
    return rgb_entry("mc_conv_rgb_to_yuv", 0, images, target, numImages, imgPixels, s);
}

int mc_conv_rgb_to_lab(const float* images, float* target, int numImages, int imgPixels, int center, mcStream_t s) {
    return rgb_entry("mc_conv_rgb_to_lab", center ? 2 : 1, images, target, numImages, imgPixels, s);
}

int mc_eltwise_max(const float* a, const float* b, float* out, size_t n, mcStream_t s) {
    MC_REQUIRE(a && b && out, "mc_eltwise_max: null pointer");
    if (n == 0) return MC_OK;
    const bool v = n % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(out);
    dim3 g(grid_items(n / (v ? 4 : 1))), blk(TPB);
    if (v) hipLaunchKernelGGL(k_eltwise_max<4>, g, blk, 0, S(s), a, b, out, n);
    else hipLaunchKernelGGL(k_eltwise_max<1>, g, blk, 0, S(s), a, b, out, n);
    return launch_status("mc_eltwise_max");
}

int mc_compute_eltwise_max_grad(const float* actGrad, const float* input, const float* output, float* target, int add, size_t n,
                                mcStream_t s) {
    MC_REQUIRE(actGrad && input && output && target, "mc_compute_eltwise_max_grad: null pointer");
    if (n == 0) return MC_OK;
    const bool v = n % 4 == 0 && aligned16(actGrad) && aligned16(input) && aligned16(output) && aligned16(target);
    dim3 g(grid_items(n / (v ? 4 : 1))), blk(TPB);
#define MC_EMG(V, A) hipLaunchKernelGGL((k_eltwise_max_grad<V, A>), g, blk, 0, S(s), actGrad, input, output, target, n)
    if (add) { if (v) MC_EMG(4, true); else MC_EMG(1, true); }
    else { if (v) MC_EMG(4, false); else MC_EMG(1, false); }
#undef MC_EMG
    return launch_status("mc_compute_eltwise_max_grad");
}

int mc_group_sparsity_in_label_cost(const float* labels, const float* acts, float* pixelSumSq, float* sqrts, float* counts,
                                    float* costTerms, int numCases, int channels, int imagePixels, mcStream_t s) {
    MC_REQUIRE(labels && acts && pixelSumSq && sqrts && counts && costTerms, "mc_group_sparsity_in_label_cost: null pointer");
    MC_REQUIRE(numCases > 0 && channels > 0 && imagePixels > 0, "mc_group_sparsity_in_label_cost: non-positive dimension");
    MC_REQUIRE((size_t)channels * imagePixels * numCases < (1ull << 31), "mc_group_sparsity_in_label_cost: tensor exceeds 32-bit indexing");
    dim3 g(grid_items((size_t)channels * numCases)), b(TPB);
    hipLaunchKernelGGL(k_gs_pixel_sumsq, g, b, 0, S(s), acts, pixelSumSq, numCases, channels, imagePixels);
    hipLaunchKernelGGL(k_gs_groups, g, b, 0, S(s), labels, pixelSumSq, sqrts, counts, costTerms, numCases, channels);
    return launch_status("mc_group_sparsity_in_label_cost");
}

int mc_group_sparsity_in_label_grad(const float* acts, const float* sqrts, const float* counts, float* target, int numCases,
                                    int channels, int imagePixels, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(acts && sqrts && counts && target, "mc_group_sparsity_in_label_grad: null pointer");
    MC_REQUIRE(numCases > 0 && channels > 0 && imagePixels > 0, "mc_group_sparsity_in_label_grad: non-positive dimension");
    MC_REQUIRE((size_t)channels * imagePixels * numCases < (1ull << 31), "mc_group_sparsity_in_label_grad: tensor exceeds 32-bit indexing");
    const bool v = use_vec(numCases, {acts, sqrts, counts, target});
    dim3 g(grid_items((size_t)channels * imagePixels * (numCases / (v ? 4 : 1)))), b(TPB);
    if (v) hipLaunchKernelGGL(k_gs_grad<4>, g, b, 0, S(s), acts, sqrts, counts, target, numCases, channels, imagePixels, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_gs_grad<1>, g, b, 0, S(s), acts, sqrts, counts, target, numCases, channels, imagePixels, scaleTargets, scaleOutput);
    return launch_status("mc_group_sparsity_in_label_grad");
}

int mc_conv_gather_channels(const float* images, const int* channelIndices, float* target, int numImages, int numImgColors,
                            int numIndices, int imgPixels, mcStream_t s) {
    MC_REQUIRE(images && channelIndices && target, "mc_conv_gather_channels: null pointer");
    MC_REQUIRE(images != target, "mc_conv_gather_channels: target must not alias images");
    MC_REQUIRE(numImages > 0 && numImgColors > 0 && numIndices > 0 && imgPixels > 0, "mc_conv_gather_channels: non-positive dimension");
    MC_REQUIRE((size_t)numImgColors * imgPixels * numImages < (1ull << 31) && (size_t)numIndices * imgPixels * numImages < (1ull << 31),
               "mc_conv_gather_channels: tensor exceeds 32-bit indexing");
    const bool v = use_vec(numImages, {images, target});
    dim3 g(grid_items((size_t)numIndices * imgPixels * (numImages / (v ? 4 : 1)))), b(TPB);
    if (v) hipLaunchKernelGGL(k_gather_channels<4>, g, b, 0, S(s), images, channelIndices, target, numImages, numIndices, imgPixels, numImgColors);
    else hipLaunchKernelGGL(k_gather_channels<1>, g, b, 0, S(s), images, channelIndices, target, numImages, numIndices, imgPixels, numImgColors);
    return launch_status("mc_conv_gather_channels");
}

int mc_conv_scatter_channels(const float* gathered, const int* inverseIndices, float* target, int numImages, int numImgColors,
                             int overSample, int imgPixels, float scaleTargets, float scaleOutput, mcStream_t s) {
    MC_REQUIRE(gathered && inverseIndices && target, "mc_conv_scatter_channels: null pointer");
    MC_REQUIRE(gathered != target, "mc_conv_scatter_channels: target must not alias the gathered tensor");
    MC_REQUIRE(numImages > 0 && numImgColors > 0 && overSample > 0 && imgPixels > 0, "mc_conv_scatter_channels: non-positive dimension");
    MC_REQUIRE((size_t)numImgColors * overSample * imgPixels * numImages < (1ull << 31), "mc_conv_scatter_channels: tensor exceeds 32-bit indexing");
    const bool v = use_vec(numImages, {gathered, target});
    dim3 g(grid_items((size_t)numImgColors * imgPixels * (numImages / (v ? 4 : 1)))), b(TPB);
    if (v) hipLaunchKernelGGL(k_scatter_channels<4>, g, b, 0, S(s), gathered, inverseIndices, target, numImages, numImgColors, overSample, imgPixels, numImgColors * overSample, scaleTargets, scaleOutput);
    else hipLaunchKernelGGL(k_scatter_channels<1>, g, b, 0, S(s), gathered, inverseIndices, target, numImages, numImgColors, overSample, imgPixels, numImgColors * overSample, scaleTargets, scaleOutput);
    return launch_status("mc_conv_scatter_channels");
}

}  // extern "C"
