// elementwise.hip -- HBM-bound kernels of the conv-net step for gfx950: neurons, dropout,
// Philox RNG, bias add / reductions, softmax + logreg cost, SGD update.
// All are coalesced over the innermost (image) axis, 16 bytes per lane where alignment allows,
// grid-stride with at most 8 workgroups per CU.
#include "mc_common.h"
#include "philox.h"

namespace mc {
namespace {

constexpr int TPB = 256;

inline int grid_for(size_t work_items) {
    const size_t want = (work_items + TPB - 1) / TPB;
    const size_t cap = (size_t)num_cus() * 8;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

// ---------------------------------------------------------------- generic map kernels
// out[i] = f(a[i])   /  out[i] = f(a[i], b[i])  / out[i] = f(a[i], b[i], out[i])
template <class F>
__global__ void k_map1(const float* a, float* out, size_t n, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = n / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (size_t v = i; v < n4; v += stride) {
        float4 x = a4[v];
        x.x = f(x.x); x.y = f(x.y); x.z = f(x.z); x.w = f(x.w);
        o4[v] = x;
    }
    for (size_t v = n4 * 4 + i; v < n; v += stride) out[v] = f(a[v]);
}
template <class F>
__global__ void k_map1_scalar(const float* a, float* out, size_t n, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) out[v] = f(a[v]);
}
template <class F>
__global__ void k_map2(const float* a, const float* b, float* out, size_t n, bool add, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = n / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (size_t v = i; v < n4; v += stride) {
        const float4 x = a4[v], y = b4[v];
        float4 r = make_float4(f(x.x, y.x), f(x.y, y.y), f(x.z, y.z), f(x.w, y.w));
        if (add) {
            const float4 t = o4[v];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        o4[v] = r;
    }
    for (size_t v = n4 * 4 + i; v < n; v += stride) out[v] = (add ? out[v] : 0.f) + f(a[v], b[v]);
}
template <class F>
__global__ void k_map2_scalar(const float* a, const float* b, float* out, size_t n, bool add, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride)
        out[v] = (add ? out[v] : 0.f) + f(a[v], b[v]);
}

template <class F>
int map1(const float* a, float* out, size_t n, F f, hipStream_t st) {
    if (n == 0) return MC_OK;
    if (aligned16(a) && aligned16(out))
        hipLaunchKernelGGL(k_map1<F>, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, st, a, out, n, f);
    else
        hipLaunchKernelGGL(k_map1_scalar<F>, dim3(grid_for(n)), dim3(TPB), 0, st, a, out, n, f);
    return launch_status("map1");
}
template <class F>
int map2(const float* a, const float* b, float* out, size_t n, bool add, F f, hipStream_t st) {
    if (n == 0) return MC_OK;
    if (aligned16(a) && aligned16(b) && aligned16(out))
        hipLaunchKernelGGL(k_map2<F>, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, st, a, b, out, n, add, f);
    else
        hipLaunchKernelGGL(k_map2_scalar<F>, dim3(grid_for(n)), dim3(TPB), 0, st, a, b, out, n, add, f);
    return launch_status("map2");
}

// ---------------------------------------------------------------- neuron functors
// Forward: Kernel/include/layers/neuron.cuh (operator classes), gradient: *GradientOperator.
struct FIdent { __device__ float operator()(float x) const { return x; } };
struct FRelu { __device__ float operator()(float x) const { return x < 0.f ? 0.f : x; } };           // :166-171
struct FBRelu { float a; __device__ float operator()(float x) const { return x < 0.f ? 0.f : (x > a ? a : x); } };  // :207-216
struct FAbs { __device__ float operator()(float x) const { return fabsf(x); } };
struct FLogistic { __device__ float operator()(float x) const { return 1.0f / (1.0f + expf(-x)); } };
struct FTanh {  // :289-298  a * (2 / (1 + exp(-2 b x)) - 1)
    float a, n2b;
    __device__ float operator()(float x) const { return a * (2.0f / (1.0f + expf(x * n2b)) - 1.0f); }
};
struct FSoftRelu { __device__ float operator()(float x) const { return x > 4.0f ? x : logf(1.0f + expf(x)); } };  // :338-345
struct FSquare { __device__ float operator()(float x) const { return x * x; } };
struct FSqrt { __device__ float operator()(float x) const { return sqrtf(x); } };
struct FLinear { float a, b; __device__ float operator()(float x) const { return a * x + b; } };     // :449-456
struct FScale { float a; __device__ float operator()(float x) const { return a * x; } };

// gradient functors: (actsGrad, actOrInput) -> input gradient
struct GIdent { __device__ float operator()(float g, float) const { return g; } };
struct GRelu { __device__ float operator()(float g, float y) const { return g * (y > 0.0f); } };      // :173-178
struct GBRelu { float a; __device__ float operator()(float g, float y) const { return g * (y > 0.0f) * (y < a); } };  // :218-227
struct GAbs { __device__ float operator()(float g, float x) const { return g * (x > 0.0f ? 1.0f : -1.0f); } };        // :255-260
struct GLogistic { __device__ float operator()(float g, float y) const { return g * y * (1.0f - y); } };              // :134-139
struct GTanh {  // :300-310
    float a, n4ab;
    __device__ float operator()(float g, float y) const {
        const float t = (1.0f - y / a) / 2.0f;
        return g * n4ab * (t * (t - 1.0f));
    }
};
struct GSoftRelu {  // :347-356 (the reference feeds the OUTPUT here; reproduced)
    __device__ float operator()(float g, float u) const {
        if (u > 4.0f) return g;
        const float f = expf(u);
        return g * (f / (1.0f + f));
    }
};
struct GSquare { __device__ float operator()(float g, float x) const { return g * 2.0f * x; } };     // :384-389
struct GSqrt { __device__ float operator()(float g, float y) const { return g / (2.0f * y); } };     // :416-421
struct GLinear { float a; __device__ float operator()(float g, float) const { return g * a; } };     // :440-442
// dropout: PluginsSrc/include/extralayers.cuh:81-96
struct FDropout { float p; __device__ float operator()(float src, float u) const { return u < p ? 0.f : src; } };

// ---------------------------------------------------------------- Philox4x32-10 (philox.h)
__global__ void k_rand_uniform(float* out, size_t n, unsigned long long seed, unsigned long long offset) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nblk = (n + 3) / 4;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        const unsigned long long ctr = offset / 4 + b;  // one Philox block = 4 outputs
        unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = (unsigned)(offset & 3), c3 = 0x6d633535u;
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float v[4] = {u01(c0), u01(c1), u01(c2), u01(c3)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b * 4 + u < n) out[b * 4 + u] = v[u];
    }
}

// Dropout whose uniforms are never stored: lane v handles elements 4v..4v+3 = one Philox block of the stream
// k_rand_uniform would write for (seed, offset), so the result equals mc_rand_uniform + mc_dropout_apply bit for bit
// and the backward pass regenerates the same mask from the same (seed, offset).
__global__ void k_dropout_counter(const float* in, float* out, float p, bool add, size_t n, unsigned long long seed, unsigned long long offset,
                                  bool vec4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nblk = (n + 3) / 4;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        const unsigned long long ctr = offset / 4 + b;
        unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = (unsigned)(offset & 3), c3 = 0x6d633535u;
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float u[4] = {u01(c0), u01(c1), u01(c2), u01(c3)};
        if (vec4 && b * 4 + 3 < n) {
            const float4 x = reinterpret_cast<const float4*>(in)[b];
            float4 r = make_float4(u[0] < p ? 0.f : x.x, u[1] < p ? 0.f : x.y, u[2] < p ? 0.f : x.z, u[3] < p ? 0.f : x.w);
            if (add) {
                const float4 t = reinterpret_cast<float4*>(out)[b];
                r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
            }
            reinterpret_cast<float4*>(out)[b] = r;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (b * 4 + e < n) out[b * 4 + e] = (add ? out[b * 4 + e] : 0.f) + (u[e] < p ? 0.f : in[b * 4 + e]);
        }
    }
}

// ---------------------------------------------------------------- bias add / reductions
// acts(rows, cols): acts[r][c] += s*vec[r]
__global__ void k_add_row_vector(float* acts, const float* __restrict__ vec, int rows, int cols, float s, bool vec4) {
    const size_t total = (size_t)rows * cols;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (vec4) {
        float4* a4 = reinterpret_cast<float4*>(acts);
        const int c4 = cols / 4;
        for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total / 4; v += stride) {
            const float b = s * vec[v / c4];
            float4 x = a4[v];
            x.x += b; x.y += b; x.z += b; x.w += b;
            a4[v] = x;
        }
    } else {
        for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += stride) acts[v] += s * vec[v / cols];
    }
}

// out[r] = sT*out[r] + sO*sum_c v[r][c]; one workgroup of 1024 threads per (row, segment);
// fixed order: per-thread strided partials -> wave shuffle tree -> LDS tree.
constexpr int RED_T = 1024;
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (w == 0) {
        r = lane < (int)(blockDim.x >> 6) ? sh[lane] : 0.f;
        r = wave_sum(r);
    }
    return r;  // valid in thread 0
}
__global__ void __launch_bounds__(RED_T)
k_sum_rows(const float* __restrict__ v, float* out, int rows, int cols, float sT, float sO, bool vec4) {
    __shared__ float sh[RED_T / 64];
    const int r = blockIdx.x;
    const float* row = v + (size_t)r * cols;
    float acc = 0.f;
    if (vec4) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int c = threadIdx.x; c < cols / 4; c += RED_T) {
            const float4 x = r4[c];
            acc += (x.x + x.y) + (x.z + x.w);
        }
    } else {
        for (int c = threadIdx.x; c < cols; c += RED_T) acc += row[c];
    }
    const float tot = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) out[r] = (sT == 0.f ? 0.f : sT * out[r]) + sO * tot;
}

// out[c] = sT*out[c] + sO*sum_r v[r][c]
__global__ void k_sum_cols(const float* __restrict__ v, float* out, int rows, int cols, float sT, float sO) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < (size_t)cols; c += stride) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) acc += v[(size_t)r * cols + c];
        out[c] = (sT == 0.f ? 0.f : sT * out[c]) + sO * acc;
    }
}
// A second, independent sum (same rows, overwriting) can ride in the same launch: blocks [nb1, gridDim.x) serve it.  The
// split-K weight gradients reduce their slabs and the bias gradient's partial sums with one launch this way.
struct SumSeg2 {
    const float4* v;
    float4* out;
    int cols4;
    float sO;
    int nb1;   // blocks of the first sum; 0 = no second sum
};
__global__ void k_sum_cols4(const float4* __restrict__ v, float4* out, int rows, int cols4, float sT, float sO, SumSeg2 two) {
    int blk = blockIdx.x, nblk = two.nb1 ? two.nb1 : gridDim.x;
    if (two.nb1 && blk >= two.nb1) {
        v = two.v; out = two.out; cols4 = two.cols4; sT = 0.f; sO = two.sO;
        blk -= two.nb1; nblk = gridDim.x - two.nb1;
    }
    const size_t stride = (size_t)nblk * blockDim.x;
    for (size_t c = (size_t)blk * blockDim.x + threadIdx.x; c < (size_t)cols4; c += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < rows; ++r) {
            const float4 x = v[(size_t)r * cols4 + c];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sT != 0.f) {
            o = out[c];
            o.x *= sT; o.y *= sT; o.z *= sT; o.w *= sT;
        }
        o.x += sO * acc.x; o.y += sO * acc.y; o.z += sO * acc.z; o.w += sO * acc.w;
        out[c] = o;
    }
}

__global__ void k_accumulate_f64(double* dst, const double* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] += src[i];
}

__global__ void __launch_bounds__(RED_T) k_sum_all(const float* __restrict__ v, size_t n, double* out) {
    __shared__ double shd[RED_T];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += RED_T) acc += (double)v[i];
    shd[threadIdx.x] = acc;
    __syncthreads();
    for (int s = RED_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) shd[threadIdx.x] += shd[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = shd[0];
}

// ---------------------------------------------------------------- axpby / update
struct FAxpby { float sy, sx; __device__ float operator()(float x, float y) const { return sy * y + sx * x; } };

// One launch for every tensor of the net: workgroup b updates span b of the arenas.  Spans never
// straddle tensors, start 16-byte aligned, and carry their tensor's hyper-parameters.
__global__ void k_weights_update_spans(float* w, float* inc, const float* __restrict__ grad, const mcUpdateSpan* __restrict__ spans) {
    const mcUpdateSpan sp = spans[blockIdx.x];
    float* W = w + sp.start;
    float* I = inc + sp.start;
    const float* G = grad + sp.start;
    const unsigned n4 = sp.count / 4;
    for (unsigned v = threadIdx.x; v < n4; v += blockDim.x) {
        float4 ww = reinterpret_cast<float4*>(W)[v], ii = reinterpret_cast<float4*>(I)[v];
        const float4 gg = reinterpret_cast<const float4*>(G)[v];
        ii.x = sp.mom * ii.x + gg.x; ii.y = sp.mom * ii.y + gg.y; ii.z = sp.mom * ii.z + gg.z; ii.w = sp.mom * ii.w + gg.w;
        if (sp.wceps != 0.f) { ii.x += sp.wceps * ww.x; ii.y += sp.wceps * ww.y; ii.z += sp.wceps * ww.z; ii.w += sp.wceps * ww.w; }
        ww.x += ii.x; ww.y += ii.y; ww.z += ii.z; ww.w += ii.w;
        reinterpret_cast<float4*>(I)[v] = ii;
        reinterpret_cast<float4*>(W)[v] = ww;
    }
    for (unsigned v = n4 * 4 + threadIdx.x; v < sp.count; v += blockDim.x) {
        float ii = sp.mom * I[v] + G[v];
        if (sp.wceps != 0.f) ii += sp.wceps * W[v];
        I[v] = ii;
        W[v] += ii;
    }
}

__global__ void k_weights_update(float* w, float* inc, const float* __restrict__ grad, size_t n, float mom, float wceps, bool vec4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec4) {
        float4* w4 = reinterpret_cast<float4*>(w);
        float4* i4 = reinterpret_cast<float4*>(inc);
        const float4* g4 = reinterpret_cast<const float4*>(grad);
        for (size_t v = i; v < n / 4; v += stride) {
            float4 W = w4[v], I = i4[v];
            const float4 G = g4[v];
            // Weights::update (weights.cuh:155-162): inc = mom*inc + grad; inc += (-wc*eps)*W; W += inc
            I.x = mom * I.x + G.x; I.y = mom * I.y + G.y; I.z = mom * I.z + G.z; I.w = mom * I.w + G.w;
            if (wceps != 0.f) { I.x += wceps * W.x; I.y += wceps * W.y; I.z += wceps * W.z; I.w += wceps * W.w; }
            W.x += I.x; W.y += I.y; W.z += I.z; W.w += I.w;
            i4[v] = I;
            w4[v] = W;
        }
        for (size_t v = (n / 4) * 4 + i; v < n; v += stride) {
            float I = mom * inc[v] + grad[v];
            if (wceps != 0.f) I += wceps * w[v];
            inc[v] = I;
            w[v] += I;
        }
    } else {
        for (size_t v = i; v < n; v += stride) {
            float I = mom * inc[v] + grad[v];
            if (wceps != 0.f) I += wceps * w[v];
            inc[v] = I;
            w[v] += I;
        }
    }
}

// ---------------------------------------------------------------- softmax / logreg (one thread per case)
// probs (numOut, numCases) row-major; coalesced across cases.
__global__ void k_softmax(const float* __restrict__ in, float* out, int numCases, int numOut) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= numCases) return;
    float mx = in[n];
    for (int j = 1; j < numOut; ++j) mx = fmaxf(mx, in[(size_t)j * numCases + n]);
    float sum = 0.f;
    for (int j = 0; j < numOut; ++j) {
        const float e = expf(in[(size_t)j * numCases + n] - mx);
        out[(size_t)j * numCases + n] = e;
        sum += e;
    }
    for (int j = 0; j < numOut; ++j) out[(size_t)j * numCases + n] = out[(size_t)j * numCases + n] / sum;
}
// ---- wide classifiers (>= 64 classes) -------------------------------------------------------------------
// One lane walking all classes of a case is right for ten classes and 385 us for a thousand.  Here a workgroup takes four
// neighbouring cases (one float4 per class row) and its 256 lanes take the classes 256 apart; maxima and sums are combined
// in a fixed tree, so results are deterministic (the order of the additions differs from the one-lane kernels': the two
// families agree to rounding, not bit for bit, which is why a given class count always takes the same family).
constexpr int WIDE_T = 256;
__device__ __forceinline__ float4 wide_reduce(float4 v, float4* sh, bool isMax) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = WIDE_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float4 a = sh[threadIdx.x], b = sh[threadIdx.x + s];
            sh[threadIdx.x] = isMax ? make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w))
                                    : make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    const float4 r = sh[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(WIDE_T) k_softmax_wide(const float* in, float* out, int numCases, int numOut) {
    __shared__ float4 sh[WIDE_T];
    const size_t n = (size_t)blockIdx.x * 4;
    const float4* src = reinterpret_cast<const float4*>(in + n);
    const size_t rowStride = (size_t)numCases / 4;
    float4 mx = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 x = src[j * rowStride];
        mx = make_float4(fmaxf(mx.x, x.x), fmaxf(mx.y, x.y), fmaxf(mx.z, x.z), fmaxf(mx.w, x.w));
    }
    mx = wide_reduce(mx, sh, true);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 x = src[j * rowStride];
        sum.x += expf(x.x - mx.x); sum.y += expf(x.y - mx.y); sum.z += expf(x.z - mx.z); sum.w += expf(x.w - mx.w);
    }
    sum = wide_reduce(sum, sh, false);
    float4* dst = reinterpret_cast<float4*>(out + n);   // (out may be in: every element is read and written by one lane, reads first)
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 x = src[j * rowStride];
        dst[j * rowStride] = make_float4(expf(x.x - mx.x) / sum.x, expf(x.y - mx.y) / sum.y, expf(x.z - mx.z) / sum.z, expf(x.w - mx.w) / sum.w);
    }
}

// per-case part of kLogregCost for four cases per workgroup: max and its multiplicity are order-free, so the values are the
// one-lane kernel's bit for bit
__global__ void __launch_bounds__(WIDE_T) k_logreg_cost_wide(const float* __restrict__ labels, const float* __restrict__ probs,
                                                             float* labelLogProbs, float* correctProbs, int numCases, int numOut) {
    __shared__ float4 sh[WIDE_T];
    const size_t n = (size_t)blockIdx.x * 4;
    const float4* src = reinterpret_cast<const float4*>(probs + n);
    const size_t rowStride = (size_t)numCases / 4;
    float4 mx = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 x = src[j * rowStride];
        mx = make_float4(fmaxf(mx.x, x.x), fmaxf(mx.y, x.y), fmaxf(mx.z, x.z), fmaxf(mx.w, x.w));
    }
    mx = wide_reduce(mx, sh, true);
    float4 cnt = make_float4(0.f, 0.f, 0.f, 0.f);   // counts up to numOut: exact in float
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 x = src[j * rowStride];
        cnt.x += x.x == mx.x; cnt.y += x.y == mx.y; cnt.z += x.z == mx.z; cnt.w += x.w == mx.w;
    }
    cnt = wide_reduce(cnt, sh, false);
    if (threadIdx.x < 4) {
        const int e = threadIdx.x;
        const float maxp = e == 0 ? mx.x : e == 1 ? mx.y : e == 2 ? mx.z : mx.w;
        const float c = e == 0 ? cnt.x : e == 1 ? cnt.y : e == 2 ? cnt.z : cnt.w;
        const int label = (int)labels[n + e];
        const float lp = probs[(size_t)label * numCases + n + e];
        labelLogProbs[n + e] = logf(lp);
        correctProbs[n + e] = lp == maxp ? 1.0f / c : 0.f;
    }
}

// kSoftmaxGrad for wide classifiers: target = y * (g - sum_j g_j y_j), the dot product added in the fixed tree
__global__ void __launch_bounds__(WIDE_T) k_softmax_grad_wide(const float* __restrict__ y, const float* __restrict__ g, float* target, bool add,
                                                              int numCases, int numOut) {
    __shared__ float4 sh[WIDE_T];
    const size_t n = (size_t)blockIdx.x * 4;
    const float4* y4 = reinterpret_cast<const float4*>(y + n);
    const float4* g4 = reinterpret_cast<const float4*>(g + n);
    float4* t4 = reinterpret_cast<float4*>(target + n);
    const size_t rowStride = (size_t)numCases / 4;
    float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 a = g4[j * rowStride], b = y4[j * rowStride];
        dot.x += a.x * b.x; dot.y += a.y * b.y; dot.z += a.z * b.z; dot.w += a.w * b.w;
    }
    dot = wide_reduce(dot, sh, false);
    for (int j = threadIdx.x; j < numOut; j += WIDE_T) {
        const float4 a = g4[j * rowStride], b = y4[j * rowStride];
        float4 v = make_float4(b.x * (a.x - dot.x), b.y * (a.y - dot.y), b.z * (a.z - dot.z), b.w * (a.w - dot.w));
        if (add) {
            const float4 t = t4[j * rowStride];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        t4[j * rowStride] = v;
    }
}

// the two sums of k_logreg_cost_sums over per-case values that are already there, added in that kernel's order
__global__ void __launch_bounds__(RED_T) k_logreg_pair_sums(const float* __restrict__ labelLogProbs, const float* __restrict__ correctProbs,
                                                            double* sums, int numCases) {
    __shared__ double shl[RED_T], shc[RED_T];
    double accL = 0.0, accC = 0.0;
    for (int n = threadIdx.x; n < numCases; n += RED_T) {
        accL += (double)labelLogProbs[n];
        accC += (double)correctProbs[n];
    }
    shl[threadIdx.x] = accL;
    shc[threadIdx.x] = accC;
    __syncthreads();
    for (int s = RED_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            shl[threadIdx.x] += shl[threadIdx.x + s];
            shc[threadIdx.x] += shc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = shl[0];
        sums[1] = shc[0];
    }
}

// kLogregCost (layer_kernels.cu:41-74)
__global__ void k_logreg_cost(const float* __restrict__ labels, const float* __restrict__ probs, float* labelLogProbs,
                              float* correctProbs, int numCases, int numOut) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= numCases) return;
    const int label = (int)labels[n];
    float maxp = probs[n];
    for (int j = 1; j < numOut; ++j) maxp = fmaxf(maxp, probs[(size_t)j * numCases + n]);
    const float lp = probs[(size_t)label * numCases + n];
    labelLogProbs[n] = logf(lp);
    if (lp != maxp) {
        correctProbs[n] = 0.f;
    } else {
        int numMax = 0;
        for (int j = 0; j < numOut; ++j) numMax += probs[(size_t)j * numCases + n] == maxp;
        correctProbs[n] = 1.0f / (float)numMax;
    }
}
// k_logreg_cost followed by the two sums of LogregCostLayer::fpropActs (layer.cu:993-1005) in one workgroup: lane t takes
// cases t, t + RED_T, ... and adds them in that order, then the same tree as k_sum_all -- the sums equal
// mc_compute_logreg_cost + 2 x mc_sum_all bit for bit, without the two dependent launches.
__global__ void __launch_bounds__(RED_T) k_logreg_cost_sums(const float* __restrict__ labels, const float* __restrict__ probs,
                                                            float* labelLogProbs, float* correctProbs, double* sums, int numCases, int numOut) {
    __shared__ double shl[RED_T], shc[RED_T];
    double accL = 0.0, accC = 0.0;
    for (int n = threadIdx.x; n < numCases; n += RED_T) {
        const int label = (int)labels[n];
        float maxp = probs[n];
        for (int j = 1; j < numOut; ++j) maxp = fmaxf(maxp, probs[(size_t)j * numCases + n]);
        const float lp = probs[(size_t)label * numCases + n];
        const float ll = logf(lp);
        float cp = 0.f;
        if (lp == maxp) {
            int numMax = 0;
            for (int j = 0; j < numOut; ++j) numMax += probs[(size_t)j * numCases + n] == maxp;
            cp = 1.0f / (float)numMax;
        }
        labelLogProbs[n] = ll;
        correctProbs[n] = cp;
        accL += (double)ll;
        accC += (double)cp;
    }
    shl[threadIdx.x] = accL;
    shc[threadIdx.x] = accC;
    __syncthreads();
    for (int s = RED_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            shl[threadIdx.x] += shl[threadIdx.x + s];
            shc[threadIdx.x] += shc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = shl[0];
        sums[1] = shc[0];
    }
}
// The tail of a classifier in one launch: k_softmax, k_logreg_cost_sums and (optionally) k_logreg_softmax_grad for the same
// cases, each formula exactly as in those kernels, so every output equals theirs bit for bit.  One workgroup: lane t takes
// cases t, t + RED_T, ... (three dependent launches of a few microseconds each are all latency).
__global__ void __launch_bounds__(RED_T) k_softmax_logreg(const float* __restrict__ labels, const float* __restrict__ in, float* probs,
                                                          float* labelLogProbs, float* correctProbs, double* sums, float* grad, bool add,
                                                          float coeff, int numCases, int numOut) {
    __shared__ double shl[RED_T], shc[RED_T];
    double accL = 0.0, accC = 0.0;
    for (int n = threadIdx.x; n < numCases; n += RED_T) {
        // Every pass reads the inputs afresh (they stay in cache) and recomputes exp(x - max) instead of reading back what the
        // pass before stored: the values are the same bits, and the loads of a pass are independent of each other, so a lane
        // has several in flight instead of one store -> load round trip per class.
        const float* col = in + n;
        const int label = (int)labels[n];
        float mx, sum = 0.f, maxp = 0.f, lp = 0.f, cp = 0.f;
        if (numOut <= 16) {   // the ten-class nets: one pass of loads, the rest from registers (same operations in the same order)
            float x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = j < numOut ? col[(size_t)j * numCases] : 0.f;
            mx = x[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mx = j < numOut ? fmaxf(mx, x[j]) : mx;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                x[j] = expf(x[j] - mx);
                sum = j < numOut ? sum + x[j] : sum;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                x[j] = x[j] / sum;
                if (j < numOut) {
                    const size_t idx = (size_t)j * numCases + n;
                    probs[idx] = x[j];
                    maxp = j == 0 ? x[j] : fmaxf(maxp, x[j]);
                    lp = j == label ? x[j] : lp;
                    if (grad) {
                        const float v = coeff * ((label == j) - x[j]);
                        grad[idx] = add ? grad[idx] + v : v;
                    }
                }
            }
            if (lp == maxp) {
                int numMax = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) numMax += (j < numOut) & (x[j] == maxp);
                cp = 1.0f / (float)numMax;
            }
        } else {
            mx = col[0];
#pragma unroll 8
            for (int j = 1; j < numOut; ++j) mx = fmaxf(mx, col[(size_t)j * numCases]);
#pragma unroll 8
            for (int j = 0; j < numOut; ++j) sum += expf(col[(size_t)j * numCases] - mx);
#pragma unroll 4
            for (int j = 0; j < numOut; ++j) {
                const size_t idx = (size_t)j * numCases + n;
                const float p = expf(col[(size_t)j * numCases] - mx) / sum;
                probs[idx] = p;
                maxp = j == 0 ? p : fmaxf(maxp, p);
                lp = j == label ? p : lp;
                if (grad) {
                    const float v = coeff * ((label == j) - p);
                    grad[idx] = add ? grad[idx] + v : v;
                }
            }
            if (lp == maxp) {
                int numMax = 0;
#pragma unroll 8
                for (int j = 0; j < numOut; ++j) numMax += expf(col[(size_t)j * numCases] - mx) / sum == maxp;
                cp = 1.0f / (float)numMax;
            }
        }
        const float ll = logf(lp);
        labelLogProbs[n] = ll;
        correctProbs[n] = cp;
        accL += (double)ll;
        accC += (double)cp;
    }
    shl[threadIdx.x] = accL;
    shc[threadIdx.x] = accC;
    __syncthreads();
    for (int s = RED_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            shl[threadIdx.x] += shl[threadIdx.x + s];
            shc[threadIdx.x] += shc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = shl[0];
        sums[1] = shc[0];
    }
}
// kLogregCostGrad (layer_kernels.cu:84-101): dE/dy_j = coeff * (label == j) / y_j
__global__ void k_logreg_grad(const float* __restrict__ labels, const float* __restrict__ probs, float* target, bool add,
                              float coeff, int numCases, int numOut) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (n >= numCases) return;
    const size_t idx = (size_t)j * numCases + n;
    const int label = (int)labels[n];
    const float v = coeff * (label == j ? 1.0f / probs[idx] : 0.f);
    target[idx] = add ? target[idx] + v : v;
}
// kSoftmaxGrad (layer_kernels.cu:109-128): dE/dx_j = y_j * (g_j - sum_k g_k y_k)
__global__ void k_softmax_grad(const float* __restrict__ y, const float* __restrict__ g, float* target, bool add,
                               int numCases, int numOut) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= numCases) return;
    float dot = 0.f;
    for (int j = 0; j < numOut; ++j) dot += g[(size_t)j * numCases + n] * y[(size_t)j * numCases + n];
    for (int j = 0; j < numOut; ++j) {
        const size_t idx = (size_t)j * numCases + n;
        const float v = y[idx] * (g[idx] - dot);
        target[idx] = add ? target[idx] + v : v;
    }
}
// kLogregSoftmaxGrad (layer_kernels.cu:137-152): dE/dx_j = coeff * ((label == j) - y_j)
__global__ void k_logreg_softmax_grad(const float* __restrict__ labels, const float* __restrict__ probs, float* target,
                                      bool add, float coeff, int numCases, int numOut) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (n >= numCases) return;
    const size_t idx = (size_t)j * numCases + n;
    const int label = (int)labels[n];
    const float v = coeff * ((label == j) - probs[idx]);
    target[idx] = add ? target[idx] + v : v;
}

}  // namespace
}  // namespace mc

using namespace mc;

// classifiers of MC_WIDE_CLASSES (64) classes or more take the class-parallel kernels (whole float4 groups of cases only)
static inline bool wide_classifier(int numCases, int numOut) { return numOut >= 64 && numCases % 4 == 0; }

extern "C" {

int mc_neuron_activate(int type, float a, float b, const float* in, float* out, size_t n, mcStream_t s) {
    MC_REQUIRE(in && out, "mc_neuron_activate: null pointer");
    hipStream_t st = S(s);
    switch (type) {
        case MC_N_IDENT:
            if (in == out) return MC_OK;
            return map1(in, out, n, FIdent(), st);
        case MC_N_RELU: return map1(in, out, n, FRelu(), st);
        case MC_N_BRELU: return map1(in, out, n, FBRelu{a}, st);
        case MC_N_ABS: return map1(in, out, n, FAbs(), st);
        case MC_N_LOGISTIC: return map1(in, out, n, FLogistic(), st);
        case MC_N_TANH: return map1(in, out, n, FTanh{a, -2.0f * b}, st);
        case MC_N_SOFTRELU: return map1(in, out, n, FSoftRelu(), st);
        case MC_N_SQUARE: return map1(in, out, n, FSquare(), st);
        case MC_N_SQRT: return map1(in, out, n, FSqrt(), st);
        case MC_N_LINEAR: return map1(in, out, n, FLinear{a, b}, st);
    }
    set_error("mc_neuron_activate: unknown neuron type %d", type);
    return MC_EINVAL;
}

int mc_neuron_input_grad(int type, float a, float b, const float* g, const float* y, float* target, int add, size_t n,
                         mcStream_t s) {
    MC_REQUIRE(g && target, "mc_neuron_input_grad: null pointer");
    hipStream_t st = S(s);
    const float* yy = y ? y : g;  // ident / linear ignore it
    const bool ad = add != 0;
    switch (type) {
        case MC_N_IDENT:
            if (!ad && g == target) return MC_OK;
            return map2(g, yy, target, n, ad, GIdent(), st);
        case MC_N_RELU: MC_REQUIRE(y, "relu gradient needs the activations"); return map2(g, y, target, n, ad, GRelu(), st);
        case MC_N_BRELU: MC_REQUIRE(y, "brelu gradient needs the activations"); return map2(g, y, target, n, ad, GBRelu{a}, st);
        case MC_N_ABS: MC_REQUIRE(y, "abs gradient needs the inputs"); return map2(g, y, target, n, ad, GAbs(), st);
        case MC_N_LOGISTIC: MC_REQUIRE(y, "logistic gradient needs the activations"); return map2(g, y, target, n, ad, GLogistic(), st);
        case MC_N_TANH: MC_REQUIRE(y, "tanh gradient needs the activations"); return map2(g, y, target, n, ad, GTanh{a, -4.0f * a * b}, st);
        case MC_N_SOFTRELU: MC_REQUIRE(y, "softrelu gradient needs the activations"); return map2(g, y, target, n, ad, GSoftRelu(), st);
        case MC_N_SQUARE: MC_REQUIRE(y, "square gradient needs the inputs"); return map2(g, y, target, n, ad, GSquare(), st);
        case MC_N_SQRT: MC_REQUIRE(y, "sqrt gradient needs the activations"); return map2(g, y, target, n, ad, GSqrt(), st);
        case MC_N_LINEAR: return map2(g, yy, target, n, ad, GLinear{a}, st);
    }
    set_error("mc_neuron_input_grad: unknown neuron type %d", type);
    return MC_EINVAL;
}

int mc_dropout_apply(const float* in, const float* mask, float* out, float prob, int add, size_t n, mcStream_t s) {
    MC_REQUIRE(in && mask && out, "mc_dropout_apply: null pointer");
    MC_REQUIRE(prob > 0.f && prob < 1.f, "mc_dropout_apply: prob must be in (0,1), got %g", (double)prob);
    return map2(in, mask, out, n, add != 0, FDropout{prob}, S(s));
}

int mc_dropout_counter(const float* in, float* out, float prob, int add, size_t n, unsigned long long seed, unsigned long long offset,
                       mcStream_t s) {
    MC_REQUIRE(in && out, "mc_dropout_counter: null pointer");
    MC_REQUIRE(prob > 0.f && prob < 1.f, "mc_dropout_counter: prob must be in (0,1), got %g", (double)prob);
    MC_REQUIRE(offset % 4 == 0, "mc_dropout_counter: offset must be a multiple of 4");
    if (n == 0) return MC_OK;
    hipLaunchKernelGGL(k_dropout_counter, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, S(s), in, out, prob, add != 0, n, seed, offset,
                       aligned16(in) && aligned16(out));
    return launch_status("mc_dropout_counter");
}

int mc_rand_uniform(float* out, size_t n, unsigned long long seed, unsigned long long offset, mcStream_t s) {
    MC_REQUIRE(out, "mc_rand_uniform: null pointer");
    MC_REQUIRE(offset % 4 == 0, "mc_rand_uniform: offset must be a multiple of 4");
    if (n == 0) return MC_OK;
    hipLaunchKernelGGL(k_rand_uniform, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, S(s), out, n, seed, offset);
    return launch_status("mc_rand_uniform");
}

int mc_add_row_vector(float* acts, const float* vec, int rows, int cols, float scaleVec, mcStream_t s) {
    MC_REQUIRE(acts && vec && rows > 0 && cols > 0, "mc_add_row_vector: bad arguments");
    const bool v4 = cols % 4 == 0 && aligned16(acts);
    const size_t items = (size_t)rows * cols / (v4 ? 4 : 1);
    hipLaunchKernelGGL(k_add_row_vector, dim3(grid_for(items)), dim3(TPB), 0, S(s), acts, vec, rows, cols, scaleVec, v4);
    return launch_status("mc_add_row_vector");
}

int mc_sum_rows(const float* v, float* out, int rows, int cols, float sT, float sO, mcStream_t s) {
    MC_REQUIRE(v && out && rows > 0 && cols > 0, "mc_sum_rows: bad arguments");
    const bool v4 = cols % 4 == 0 && aligned16(v);
    hipLaunchKernelGGL(k_sum_rows, dim3(rows), dim3(RED_T), 0, S(s), v, out, rows, cols, sT, sO, v4);
    return launch_status("mc_sum_rows");
}

// rows >> 1: workgroup = 16 float4 columns x 16 row groups; each thread adds rows rg, rg+16, ...
// in order, then the 16 partials of a column are added in a fixed tree -> run-to-run deterministic.
__global__ void __launch_bounds__(256) k_sum_cols4_tall(const float4* __restrict__ v, float4* out, int rows, int cols4, float sT, float sO,
                                                          SumSeg2 two) {
    __shared__ float4 sh[16][17];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    int blk = blockIdx.x;
    if (two.nb1 && blk >= two.nb1) {
        v = two.v; out = two.out; cols4 = two.cols4; sT = 0.f; sO = two.sO;
        blk -= two.nb1;
    }
    const int c = blk * 16 + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols4) {
        // sixteen rows' loads in flight per thread, added in row order (the order of the plain loop: same bits); with one load at
        // a time the 16 dependent round trips of a 256-slab split-K reduction were the whole kernel
        int r = rg;
        for (; r + 15 * 16 < rows; r += 16 * 16) {
            float4 x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = v[(size_t)(r + 16 * u) * cols4 + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        for (; r < rows; r += 16) {
            const float4 x = v[(size_t)r * cols4 + c];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
    }
    sh[rg][cg] = acc;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (rg < half) {
            const float4 a = sh[rg][cg], b = sh[rg + half][cg];
            sh[rg][cg] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (rg == 0 && c < cols4) {
        const float4 t = sh[0][cg];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sT != 0.f) {
            o = out[c];
            o.x *= sT; o.y *= sT; o.z *= sT; o.w *= sT;
        }
        o.x += sO * t.x; o.y += sO * t.y; o.z += sO * t.z; o.w += sO * t.w;
        out[c] = o;
    }
}

// out1 = sT1 * out1 + sO1 * column sums of v1 (rows x cols1) and out2 = sO2 * column sums of v2 (rows x cols2), one launch
// when both are float4-shaped, else two.
// tallMin: from how many rows on the sum is split over 16 row groups (the callers keep the summation order of the kernel their
// unfused variant uses, so fused and unfused weight gradients agree bit for bit)
int mc__sum_cols_pair(const float* v1, float* out1, int cols1, float sT1, float sO1, const float* v2, float* out2, int cols2, float sO2, int rows,
                      int tallMin, hipStream_t st) {
    if (cols1 % 4 || cols2 % 4 || !aligned16(v1) || !aligned16(out1) || !aligned16(v2) || !aligned16(out2)) {
        const int rc = mc_sum_cols(v1, out1, rows, cols1, sT1, sO1, reinterpret_cast<mcStream_t>(st));
        return rc ? rc : mc_sum_cols(v2, out2, rows, cols2, 0.f, sO2, reinterpret_cast<mcStream_t>(st));
    }
    SumSeg2 two{reinterpret_cast<const float4*>(v2), reinterpret_cast<float4*>(out2), cols2 / 4, sO2, 0};
    if (rows >= tallMin) {
        two.nb1 = ceil_div(cols1 / 4, 16);
        hipLaunchKernelGGL(k_sum_cols4_tall, dim3(two.nb1 + ceil_div(cols2 / 4, 16)), dim3(256), 0, st, reinterpret_cast<const float4*>(v1),
                           reinterpret_cast<float4*>(out1), rows, cols1 / 4, sT1, sO1, two);
    } else {
        two.nb1 = grid_for(cols1 / 4);
        hipLaunchKernelGGL(k_sum_cols4, dim3(two.nb1 + grid_for(cols2 / 4)), dim3(TPB), 0, st, reinterpret_cast<const float4*>(v1),
                           reinterpret_cast<float4*>(out1), rows, cols1 / 4, sT1, sO1, two);
    }
    return launch_status("mc_sum_cols(pair)");
}

int mc_sum_cols(const float* v, float* out, int rows, int cols, float sT, float sO, mcStream_t s) {
    MC_REQUIRE(v && out && rows > 0 && cols > 0, "mc_sum_cols: bad arguments");
    const SumSeg2 none{nullptr, nullptr, 0, 0.f, 0};
    if (cols % 4 == 0 && aligned16(v) && aligned16(out) && rows >= 32)
        hipLaunchKernelGGL(k_sum_cols4_tall, dim3(ceil_div(cols / 4, 16)), dim3(256), 0, S(s), reinterpret_cast<const float4*>(v),
                           reinterpret_cast<float4*>(out), rows, cols / 4, sT, sO, none);
    else if (cols % 4 == 0 && aligned16(v) && aligned16(out))
        hipLaunchKernelGGL(k_sum_cols4, dim3(grid_for(cols / 4)), dim3(TPB), 0, S(s), reinterpret_cast<const float4*>(v),
                           reinterpret_cast<float4*>(out), rows, cols / 4, sT, sO, none);
    else
        hipLaunchKernelGGL(k_sum_cols, dim3(grid_for(cols)), dim3(TPB), 0, S(s), v, out, rows, cols, sT, sO);
    return launch_status("mc_sum_cols");
}

int mc_axpby(float* y, const float* x, float sy, float sx, size_t n, mcStream_t s) {
    MC_REQUIRE(y && x, "mc_axpby: null pointer");
    return map2(x, y, y, n, false, FAxpby{sy, sx}, S(s));
}

int mc_scale(float* y, const float* x, float sx, size_t n, mcStream_t s) {
    MC_REQUIRE(y && x, "mc_scale: null pointer");
    return map1(x, y, n, FScale{sx}, S(s));
}

int mc_accumulate_f64(double* dst, const double* src, int n, mcStream_t s) {
    MC_REQUIRE(dst && src && n > 0 && n <= 1024, "mc_accumulate_f64: bad arguments");
    hipLaunchKernelGGL(k_accumulate_f64, dim3(1), dim3(64), 0, S(s), dst, src, n);
    return launch_status("mc_accumulate_f64");
}

int mc_sum_all(const float* v, size_t n, double* out, mcStream_t s) {
    MC_REQUIRE(v && out, "mc_sum_all: null pointer");
    hipLaunchKernelGGL(k_sum_all, dim3(1), dim3(RED_T), 0, S(s), v, n, out);
    return launch_status("mc_sum_all");
}

int mc_weights_update(float* w, float* inc, const float* grad, size_t n, float mom, float eps, float wc, mcStream_t s) {
    MC_REQUIRE(w && inc && grad, "mc_weights_update: null pointer");
    if (n == 0) return MC_OK;
    const bool v4 = aligned16(w) && aligned16(inc) && aligned16(grad);
    const float wceps = wc > 0.f ? -wc * eps : 0.f;
    hipLaunchKernelGGL(k_weights_update, dim3(grid_for(v4 ? (n + 3) / 4 : n)), dim3(TPB), 0, S(s), w, inc, grad, n, mom,
                       wceps, v4);
    return launch_status("mc_weights_update");
}

int mc_weights_update_spans(float* weights, float* inc, const float* grad, const mcUpdateSpan* spansDevice, int numSpans, mcStream_t s) {
    MC_REQUIRE(weights && inc && grad && spansDevice, "mc_weights_update_spans: null pointer");
    MC_REQUIRE(aligned16(weights) && aligned16(inc) && aligned16(grad), "mc_weights_update_spans: arenas must be 16-byte aligned");
    if (numSpans <= 0) return MC_OK;
    hipLaunchKernelGGL(k_weights_update_spans, dim3(numSpans), dim3(TPB), 0, S(s), weights, inc, grad, spansDevice);
    return launch_status("mc_weights_update_spans");
}

int mc_softmax(const float* input, float* target, int numCases, int numOut, mcStream_t s) {
    MC_REQUIRE(input && target && numCases > 0 && numOut > 0, "mc_softmax: bad arguments");
    if (wide_classifier(numCases, numOut) && aligned16(input) && aligned16(target)) {
        hipLaunchKernelGGL(k_softmax_wide, dim3(numCases / 4), dim3(WIDE_T), 0, S(s), input, target, numCases, numOut);
        return launch_status("mc_softmax");
    }
    hipLaunchKernelGGL(k_softmax, dim3(ceil_div(numCases, 64)), dim3(64), 0, S(s), input, target, numCases, numOut);
    return launch_status("mc_softmax");
}

int mc_compute_logreg_cost(const float* labels, const float* probs, float* labelLogProbs, float* correctProbs,
                           int numCases, int numOut, mcStream_t s) {
    MC_REQUIRE(labels && probs && labelLogProbs && correctProbs && numCases > 0 && numOut > 0, "mc_compute_logreg_cost: bad arguments");
    hipLaunchKernelGGL(k_logreg_cost, dim3(ceil_div(numCases, 64)), dim3(64), 0, S(s), labels, probs, labelLogProbs,
                       correctProbs, numCases, numOut);
    return launch_status("mc_compute_logreg_cost");
}

int mc_compute_logreg_cost_sums(const float* labels, const float* probs, float* labelLogProbs, float* correctProbs, double* sums,
                                int numCases, int numOut, mcStream_t s) {
    MC_REQUIRE(labels && probs && labelLogProbs && correctProbs && sums && numCases > 0 && numOut > 0, "mc_compute_logreg_cost_sums: bad arguments");
    if (wide_classifier(numCases, numOut) && aligned16(probs)) {
        hipLaunchKernelGGL(k_logreg_cost_wide, dim3(numCases / 4), dim3(WIDE_T), 0, S(s), labels, probs, labelLogProbs, correctProbs, numCases, numOut);
        hipLaunchKernelGGL(k_logreg_pair_sums, dim3(1), dim3(RED_T), 0, S(s), labelLogProbs, correctProbs, sums, numCases);
        return launch_status("mc_compute_logreg_cost_sums");
    }
    hipLaunchKernelGGL(k_logreg_cost_sums, dim3(1), dim3(RED_T), 0, S(s), labels, probs, labelLogProbs, correctProbs, sums, numCases, numOut);
    return launch_status("mc_compute_logreg_cost_sums");
}

int mc_softmax_logreg(const float* labels, const float* input, float* probs, float* labelLogProbs, float* correctProbs, double* sums,
                      float* gradTarget, int add, float coeff, int numCases, int numOut, mcStream_t s) {
    MC_REQUIRE(labels && input && probs && labelLogProbs && correctProbs && sums && numCases > 0 && numOut > 0, "mc_softmax_logreg: bad arguments");
    MC_REQUIRE(probs != input, "mc_softmax_logreg: probs must not alias the input");
    if (wide_classifier(numCases, numOut) && aligned16(input) && aligned16(probs)) {   // the class-parallel family, launch by launch
        hipLaunchKernelGGL(k_softmax_wide, dim3(numCases / 4), dim3(WIDE_T), 0, S(s), input, probs, numCases, numOut);
        hipLaunchKernelGGL(k_logreg_cost_wide, dim3(numCases / 4), dim3(WIDE_T), 0, S(s), labels, probs, labelLogProbs, correctProbs, numCases, numOut);
        hipLaunchKernelGGL(k_logreg_pair_sums, dim3(1), dim3(RED_T), 0, S(s), labelLogProbs, correctProbs, sums, numCases);
        if (gradTarget)
            hipLaunchKernelGGL(k_logreg_softmax_grad, dim3(ceil_div(numCases, 64), numOut), dim3(64), 0, S(s), labels, probs, gradTarget, add != 0, coeff,
                               numCases, numOut);
        return launch_status("mc_softmax_logreg");
    }
    hipLaunchKernelGGL(k_softmax_logreg, dim3(1), dim3(RED_T), 0, S(s), labels, input, probs, labelLogProbs, correctProbs, sums, gradTarget,
                       add != 0, coeff, numCases, numOut);
    return launch_status("mc_softmax_logreg");
}

int mc_compute_logreg_grad(const float* labels, const float* probs, float* target, int add, float coeff, int numCases,
                           int numOut, mcStream_t s) {
    MC_REQUIRE(labels && probs && target && numCases > 0 && numOut > 0, "mc_compute_logreg_grad: bad arguments");
    hipLaunchKernelGGL(k_logreg_grad, dim3(ceil_div(numCases, 64), numOut), dim3(64), 0, S(s), labels, probs, target,
                       add != 0, coeff, numCases, numOut);
    return launch_status("mc_compute_logreg_grad");
}

int mc_compute_softmax_grad(const float* acts, const float* actsGrad, float* target, int add, int numCases, int numOut,
                            mcStream_t s) {
    MC_REQUIRE(acts && actsGrad && target && numCases > 0 && numOut > 0, "mc_compute_softmax_grad: bad arguments");
    if (wide_classifier(numCases, numOut) && aligned16(acts) && aligned16(actsGrad) && aligned16(target) && target != actsGrad) {
        hipLaunchKernelGGL(k_softmax_grad_wide, dim3(numCases / 4), dim3(WIDE_T), 0, S(s), acts, actsGrad, target, add != 0, numCases, numOut);
        return launch_status("mc_compute_softmax_grad");
    }
    hipLaunchKernelGGL(k_softmax_grad, dim3(ceil_div(numCases, 64)), dim3(64), 0, S(s), acts, actsGrad, target, add != 0,
                       numCases, numOut);
    return launch_status("mc_compute_softmax_grad");
}

int mc_compute_logreg_softmax_grad(const float* labels, const float* probs, float* target, int add, float coeff,
                                   int numCases, int numOut, mcStream_t s) {
    MC_REQUIRE(labels && probs && target && numCases > 0 && numOut > 0, "mc_compute_logreg_softmax_grad: bad arguments");
    hipLaunchKernelGGL(k_logreg_softmax_grad, dim3(ceil_div(numCases, 64), numOut), dim3(64), 0, S(s), labels, probs,
                       target, add != 0, coeff, numCases, numOut);
    return launch_status("mc_compute_logreg_softmax_grad");
}

}  // extern "C"
