// conv_band.hip -- small-filter convolution forward (stride 1, shared filters, one group) with the input BAND of an output
// block staged in LDS once and every filter tap formed from it by address, instead of an im2col tile gathered per K step
// (igemm_core.h, PatchNatAddr).  Replaces filterActs (Kernel/src/conv/filter_acts.cu:2206-2211) for the shapes the engine is
// gather-bound on: few filters (F = 32 / 64: every gathered element feeds only F / 32 matrix instructions) or few channels
// (C = 3: K = 75, five K steps per tile).
//
// Tile: all F filters x (2 x 2 output pixels x 32 images) = 128 columns.  In the batch-innermost (C, H, W, N) layout the
// coalescing unit is one pixel x 32 images = 128 bytes, so the band of a 5x5 filter is 6 x 6 pixels x CB channels x 128 B
// (9 pixels fetched per output pixel against 25 for the gather).  Wave w of the four owns pixel w of the block: its 32 columns
// are the 32 images, so the B fragment of tap (c, dy, dx) is band[c][py + dy][px + dx][image] -- one ds_read_b32 whose address
// is a lane constant + a compile-time offset.  The filters of the channel chunk sit beside the band as [k][F]; a workgroup
// whose layer has a single chunk (C == CB: first layers) keeps them for its whole life.
//
// Bit-identical to the engine: the same v_mfma_f32_32x32x2_f32, fed the same pairs (k, k + 4) of each group of 8 k in the same
// order (igemm_core.h load_slab: k = 8 k8 + 4 h + e), zero for k beyond K, the same epilogue expression.  So the choice between
// the two is a timing matter only (api.hip times both once per shape).
#include "mc_common.h"

namespace mc {
namespace {

struct BandArgs {
    const float* images;
    const float* filters;
    float* targets;
    const float* biases;   // nullable
    unsigned* bits;        // nullable (BITS instantiations): relu mask in the engine's accumulator layout
    int N, C, H, W, F, pad, MY, MX;
    int tilesX, imgBlocks, nTiles, nChunks;
    int ldc;               // MY * MX * N
    float alpha;
    int relu;
};

template <int FS, int CB>
struct BandGeo {
    static_assert(CB * FS * FS % 8 == 0 || CB * FS * FS < 8 * 16, "a channel chunk that is not the whole K must hold whole groups of 8 k");
    static constexpr int BY = FS + 1, BX = FS + 1;              // band of a 2 x 2 pixel block
    static constexpr int KC = CB * FS * FS;                     // k of one channel chunk
    static constexpr int KP = (KC + 7) / 8 * 8;                 // padded to whole groups of 8
    static constexpr int BAND_FLOATS = CB * BY * BX * 32;
    // LDS float offset of tap k inside the band (k beyond the chunk: its last tap -- finite data under a zero filter row)
    static constexpr int off(int k) {
        const int kk = k < KC ? k : KC - 1;
        const int c = kk / (FS * FS), t = kk % (FS * FS);
        return ((c * BY + t / FS) * BX + t % FS) * 32;
    }
    // the distinct values of off(k + 4) - off(k) over the k the low lane half takes (k % 8 < 4): a handful
    struct Deltas { int n; int v[6]; };
    static constexpr Deltas deltas() {
        Deltas d{0, {0, 0, 0, 0, 0, 0}};
        for (int k8 = 0; k8 < KP / 8; ++k8)
            for (int e = 0; e < 4; ++e) {
                const int dl = off(k8 * 8 + e + 4) - off(k8 * 8 + e);
                bool seen = false;
                for (int i = 0; i < d.n; ++i) seen = seen || d.v[i] == dl;
                if (!seen && d.n < 6) d.v[d.n++] = dl;
                else if (!seen) d.n = 99;
            }
        return d;
    }
};

template <int FS, int CB, int MT, bool ARES, bool BITS>
__global__ void __launch_bounds__(256) k_conv_band(BandArgs a) {
    typedef BandGeo<FS, CB> G;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int FP = MT * 32;                                  // filters of the tile (F == FP: checked on the host)
    constexpr int A_FLOATS = G::KP * FP;
    constexpr int STAGE = (ARES ? 0 : A_FLOATS) + G::BAND_FLOATS;
    constexpr auto DL = G::deltas();
    static_assert(DL.n <= 6, "too many distinct tap distances for the lane-half base registers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [bias FP][A (resident form)][stage 0][stage 1]; stage = [A chunk (chunked form)][band]
    float* biasL = smem;
    float* aRes = smem + FP;
    float* stage0 = aRes + (ARES ? A_FLOATS : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int py = wave >> 1, px = wave & 1;

    // ---- staging (all 256 threads): band items are (channel, band row, band column) x 8 float4 = 128 bytes of 32 images
    constexpr int BAND_V4 = G::BAND_FLOATS / 4, BAND_R = (BAND_V4 + 255) / 256;
    constexpr int A_V4 = G::KC * FP / 4, A_R = ARES ? 0 : (A_V4 + 255) / 256;
    float4 bandR[BAND_R], aR[A_R > 0 ? A_R : 1];
    auto tile_origin = [&](int tile, int& y0, int& x0, int& n0) {
        const int ib = tile % a.imgBlocks, pb = tile / a.imgBlocks;
        n0 = ib * 32;
        x0 = (pb % a.tilesX) * 2;
        y0 = (pb / a.tilesX) * 2;
    };
    auto fetch = [&](int tile, int chunk) {
        int y0, x0, n0;
        tile_origin(tile, y0, x0, n0);
        const float* img = a.images + (size_t)chunk * CB * a.H * a.W * a.N + n0;
#pragma unroll
        for (int r = 0; r < BAND_R; ++r) {
            const int i = tid + 256 * r, item = i >> 3, q = i & 7;
            const int c = item / (G::BY * G::BX), rem = item - c * (G::BY * G::BX);
            const int y = y0 - a.pad + rem / G::BX, x = x0 - a.pad + rem % G::BX;
            const bool ok = (BAND_V4 % 256 == 0 || i < BAND_V4) && y >= 0 && y < a.H && x >= 0 && x < a.W;
            bandR[r] = ok ? *reinterpret_cast<const float4*>(img + ((size_t)(c * a.H + y) * a.W + x) * a.N + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (!ARES) {
            const float4* w = reinterpret_cast<const float4*>(a.filters + (size_t)chunk * G::KC * FP);
#pragma unroll
            for (int r = 0; r < A_R; ++r) {
                const int i = tid + 256 * r;
                aR[r] = (A_V4 % 256 == 0 || i < A_V4) ? w[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto commit = [&](float* st) {
        float4* bd = reinterpret_cast<float4*>(st + (ARES ? 0 : A_FLOATS));
#pragma unroll
        for (int r = 0; r < BAND_R; ++r) {
            const int i = tid + 256 * r;
            if (BAND_V4 % 256 == 0 || i < BAND_V4) bd[i] = bandR[r];
        }
        if constexpr (!ARES) {
            float4* ad = reinterpret_cast<float4*>(st);
#pragma unroll
            for (int r = 0; r < A_R; ++r) {
                const int i = tid + 256 * r;
                if (A_V4 % 256 == 0 || i < A_V4) ad[i] = aR[r];
            }
        }
    };

    // ---- once per workgroup: bias column, zero filter rows beyond the chunk's k, the resident filters
    for (int i = tid; i < FP; i += 256) biasL[i] = a.biases ? a.biases[i] : 0.f;
    if constexpr (G::KP > G::KC) {
        for (int i = tid; i < (G::KP - G::KC) * FP; i += 256) {
            if (ARES) aRes[G::KC * FP + i] = 0.f;
            else { stage0[G::KC * FP + i] = 0.f; stage0[STAGE + G::KC * FP + i] = 0.f; }
        }
    }
    if constexpr (ARES) {
        const float4* w = reinterpret_cast<const float4*>(a.filters);
        for (int i = tid; i < A_V4; i += 256) reinterpret_cast<float4*>(aRes)[i] = w[i];
    }

    // ---- fragment addressing: lane bases + compile-time offsets
    const int aBase = h * 4 * FP + l31;                              // filters [k][FP]: rows k (h = 0) and k + 4 (h = 1)
    int bBase[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) bBase[i] = (py * G::BX + px) * 32 + l31 + (i < DL.n ? h * DL.v[i] : 0);

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    constexpr int SLABS = G::KP / 2, PF = 2;
    auto compute = [&](const float* A, const float* B) {
        float af[3][MT], bf[3];
        auto load_slab = [&](int s, int set) {
            const int k = (s >> 2) * 8 + (s & 3);                    // the low half's k; the high half takes k + 4
            const int ol = G::off(k), dl = G::off(k + 4) - ol;
            int di = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) di = (i < DL.n && DL.v[i] == dl) ? i : di;
            bf[set] = B[bBase[di] + ol];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[set][m] = A[aBase + k * FP + m * 32];
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) load_slab(s, s % 3);
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
            if (s + PF < SLABS) load_slab(s + PF, (s + PF) % 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % 3][m], bf[s % 3], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // C/D map of the 32x32 MFMA: column = lane & 31 (image), row = (r & 3) + 8 (r >> 2) + 4 h.  Same expression as the engine's store.
    auto store_tile = [&](int tile) {
        int y0, x0, n0;
        tile_origin(tile, y0, x0, n0);
        const int y = y0 + py, x = x0 + px;
        const bool pok = y < a.MY && x < a.MX;
        const size_t col = (size_t)(y * a.MX + x) * a.N + n0 + l31;
        const bool relu = a.relu != 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int rl0 = m * 32 + 4 * h;
            unsigned pbits = 0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 bq = *reinterpret_cast<const float4*>(&biasL[rl0 + 8 * q4]);
                const float bia[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q4 + e, f = rl0 + 8 * q4 + e;
                    float v = a.alpha * acc[m][r] + bia[e];
                    v = (relu & (v < 0.f)) ? 0.f : v;
                    if constexpr (BITS) pbits |= (v > 0.f) ? (1u << r) : 0u;
                    if (pok) a.targets[(size_t)f * a.ldc + col] = v;
                }
            }
            if constexpr (BITS) {
                if (pok) reinterpret_cast<unsigned short*>(a.bits)[((size_t)m * a.ldc + col) * 2 + h] = (unsigned short)pbits;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
    };

    // ---- the pipeline: stage s = (tile, chunk); while stage s is multiplied out of one LDS buffer the loads of stage s + 1 are
    // in flight into registers, which go to the other buffer after the multiply; one barrier per stage
    int tile = blockIdx.x, chunk = 0, buf = 0;
    if (tile < a.nTiles) {
        fetch(tile, 0);
        commit(stage0);
    }
    __syncthreads();
    while (tile < a.nTiles) {
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == a.nChunks) { nchunk = 0; ntile += gridDim.x; }
        const bool more = ntile < a.nTiles;
        if (more) fetch(ntile, nchunk);
        const float* st = stage0 + buf * STAGE;
        compute(ARES ? aRes : st, st + (ARES ? 0 : A_FLOATS));
        if (chunk == a.nChunks - 1) store_tile(tile);
        if (more) commit(stage0 + (buf ^ 1) * STAGE);
        __syncthreads();
        buf ^= 1;
        tile = ntile;
        chunk = nchunk;
    }
}

template <int FS, int CB, int MT, bool ARES, bool BITS>
int launch_band(const BandArgs& a, hipStream_t st) {
    typedef BandGeo<FS, CB> G;
    constexpr int FP = MT * 32;
    constexpr size_t lds = (size_t)(FP + (ARES ? G::KP * FP + 2 * G::BAND_FLOATS : 2 * (G::KP * FP + G::BAND_FLOATS))) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = k_conv_band<FS, CB, MT, ARES, BITS>;
    static unsigned long long attrDone = 0;
    static int perCU[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!((attrDone >> dev) & 1ull)) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int nb = 0;
        perCU[dev] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) == hipSuccess && nb > 0) ? nb : 1;
        attrDone |= 1ull << dev;
    }
    long workers = (long)perCU[dev] * num_cus();
    if (workers > a.nTiles) workers = a.nTiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)workers), dim3(256), lds, st, a);
    return launch_status("k_conv_band");
}

}  // namespace

// Which band instantiation serves this shape, or 0.  (Codes: FS * 1000 + CB * 10 + MT.)
int conv_band_kind(int N, int C, int H, int W, int F, int fs, int pad, int stride, int MY, int MX, int groups) {
    if (stride != 1 || groups != 1 || fs != 5 || N % 32 != 0 || pad < 0 || H <= 0 || W <= 0 || MY <= 0 || MX <= 0) return 0;
    if ((long)C * H * W * N >= (1L << 30) || (long)F * MY * MX * N >= (1L << 30)) return 0;
    const long tiles = (long)ceil_div(MY, 2) * ceil_div(MX, 2) * (N / 32);
    if (tiles < num_cus() / 2) return 0;   // too few blocks to fill the device: the engine splits K there
    if (C == 3 && (F == 32 || F == 64 || F == 192)) return 5000 + 30 + F / 32;
    if (F == 32 && C % 8 == 0) return 5000 + 80 + 1;
    // (64 filters over many channels: a chunk must be a multiple of 8 channels -- 25 taps x 8 -- for its groups of 8 k to be the
    // engine's, and two stages of 200 x 64 filters + the band do not fit 160 KB)
    return 0;
}

int conv_band_launch(const float* images, const float* filters, float* targets, const float* biases, int neuron, int N, int C, int H, int W,
                     int F, int fs, int pad, int MY, int MX, float scaleOutput, unsigned* reluBits, hipStream_t st) {
    const int kind = conv_band_kind(N, C, H, W, F, fs, pad, 1, MY, MX, 1);
    BandArgs a{images, filters, targets, biases, reluBits, N, C, H, W, F, pad, MY, MX, ceil_div(MX, 2), N / 32,
               ceil_div(MY, 2) * ceil_div(MX, 2) * (N / 32), 1, MY * MX * N, scaleOutput, neuron == MC_NEURON_RELU ? 1 : 0};
    switch (kind) {
        case 5031: return reluBits ? launch_band<5, 3, 1, true, true>(a, st) : launch_band<5, 3, 1, true, false>(a, st);
        case 5032: return reluBits ? launch_band<5, 3, 2, true, true>(a, st) : launch_band<5, 3, 2, true, false>(a, st);
        case 5036: return reluBits ? launch_band<5, 3, 6, true, true>(a, st) : launch_band<5, 3, 6, true, false>(a, st);
        case 5081: a.nChunks = C / 8; return reluBits ? launch_band<5, 8, 1, false, true>(a, st) : launch_band<5, 8, 1, false, false>(a, st);
    }
    set_error("conv_band_launch: no band kernel for this shape");
    return MC_EINVAL;
}

}  // namespace mc
