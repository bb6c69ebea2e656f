#pragma once
// igemm_core.h -- fp32-MFMA implicit-GEMM engine for gfx950 (MI355X): the kernel template, its operand loaders and the
// host-side launch / tile-choice helpers.  Included by igemm_{fwd,bwd,wgrad,gemm}.hip, one translation unit per family of
// instantiations (they compile in parallel).
//
// One kernel template computes C(i,j) = sum_k A(i,k) * B(k,j) with both operands supplied
// by "loaders" that know how to gather a (BK x ROWS) tile from the reference's
// batch-innermost tensors.  The contractions of the conv-net step are all instances:
//
//   conv forward   (Kernel/src/conv/filter_acts.cu:58-413)  A = filters  [k][f]      B = im2col patches [k][(m,n)]
//   conv bwd-data  (Kernel/src/conv/img_acts.cu:63-658)     A = filters^T [c][(p,f)]  B = hidActs gather [(p,f)][(px,n)]
//   conv bwd-wts   (Kernel/src/conv/weight_acts.cu:58-409)  A = patches  [(c,p)][(m,n)]  B = hidActs [f][(m,n)]
//   dense          (Kernel/src/nvmatrix/nvmatrix.cu:189-239) plain matrices, 4 transpose combos
//
// Design notes (MI355X-first, not a translation of the reference's register tiling):
//   * wave64, v_mfma_f32_32x32x2_f32: each wave owns an (MT x NT) grid of 32x32 accumulators.
//     A/B fragments are ONE f32 VGPR per lane: lane l supplies A[i = l&31][k = l>>5].
//   * Two LDS layouts, chosen by which axis is contiguous in HBM:
//       NAT  [BK][ROWS]       operand is contiguous along its row (i or j) axis -> ds_read_b32,
//                              32 consecutive floats per half-wave: conflict free.
//       KC   [BK/4][ROWS+1][4] operand is contiguous along k -> one ds_read_b128 feeds 4 MFMAs.
//     Any k permutation is legal as long as A and B agree, so inside each group of 8 k's
//     MFMA e (0..3) of lane-half h consumes k = 4h + e; that makes both layouts line up.
//   * Wave specialisation: a workgroup is 8 waves = 4 consumer waves (LDS fragment reads + MFMA only,
//     one per SIMD) and 4 producer waves (tile gather: address arithmetic, buffer loads with
//     out-of-range -> 0 predication, LDS stores).  The matrix pipe and the vector/memory pipes are
//     separate, so the producers' work runs under the consumers' MFMAs instead of in front of them.
//     Producers keep two register sets: tile t+2's loads are issued before tile t+1 is written to
//     the idle LDS buffer, so a load has up to two K steps to land.  One barrier per K step.
//   * blockIdx.z = (k-chunk q, group g): k-chunks give deterministic split-K (each chunk writes
//     its own slab; the reference's partialSum layout is exactly that), groups are independent
//     GEMMs on disjoint channel/filter ranges.
#include <stdlib.h>
#include <string.h>
#include <map>
#include <type_traits>
#include <mutex>
#include <atomic>
#include "mc_common.h"

#ifdef MC_STAMP   // diagnostic build only (tools/build_variant.sh stamp -DMC_STAMP): where do the K steps spend their cycles?
static __device__ unsigned long long g_stamp[8192 * 8];   // one array per translation unit (no relocatable device code)
#define MC_STAMP_API(tag)                                                                                      \
    extern "C" int mc_debug_read_stamps_##tag(unsigned long long* out, int n) {                                \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), (size_t)n * 8, 0, hipMemcpyDeviceToHost);    \
    }                                                                                                          \
    extern "C" int mc_debug_clear_stamps_##tag(void) {                                                         \
        static unsigned long long zeros[8192 * 8];                                                             \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice);    \
    }
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#define MC_STAMP_API(tag)
#endif

namespace mc {
extern thread_local bool g_tuneInhibit;   // defined in igemm_fwd.hip; see set_tune_inhibit
extern thread_local char* g_describeBuf;  // when set, launch_cfg describes the launch it would make here instead of making it
extern thread_local int g_describeLen;
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int THREADS = 256;
#ifndef MC_NS
#define MC_NS 3    // tile register sets per producer thread (tiles in flight per workgroup)
#endif
#ifndef MC_BK_KC
#define MC_BK_KC 32   // k extent of one LDS tile when both operands are k-contiguous (weight gradients)
// (measured, tools/build_variant.sh bk64 -DMC_BK_KC=64: 256-byte instead of 128-byte runs per row and K step do not help --
// NIN cccp1 weight gradient 117 -> 151 us, cccp2 70 -> 90, cccp3 40 -> 56: twice the fragment registers and LDS per tile cost
// more than the longer DRAM bursts gain)
#endif
#ifndef MC_BK
#define MC_BK 16   // k extent of one LDS tile for the contractions whose operands are not both k-contiguous
#endif

struct ConvG {
    int N, H, W;      // numImages, imgSizeY, imgSizeX
    int fc;           // filter channels (per group)
    int fs, fp;       // filterSize, filterSize^2
    int pad, stride;  // paddingStart (<= 0), moduleStride
    int MY, MX, M;    // modules
    int F, fpg;       // numFilters, filters per group
    int local;        // unshared filters (leading module index)
    int kcUniform;    // weight gradient: every K tile lies inside one module (numImages % BK == 0, chunks start on a tile)
    float inv_fp, inv_fs, inv_fpg, inv_N, inv_MX, inv_W, inv_stride;
};

struct Dims {
    int Mg, Ng, Kg;   // GEMM extents (per group)
    int kChunk;       // k extent of one chunk (multiple of 4 when a KC loader is vectorised)
    int groups;
    int xcdSwizzle;   // set by launch_cfg when column tiles and column workers are both multiples of 8
    int colTiles;     // column tiles of the product (set by launch_cfg); gridDim.x = column workers <= colTiles
    int batches;      // independent products sharing one launch (unshared layers: one per module / pixel); >= 1
    int colB;         // column offset of batch b in the B operand and in C: b * colB
    int ktabN;        // entries of the B loader's k table in LDS (set by launch_cfg; 0 = the loader decodes k arithmetically)
    int kStep;        // distance between consecutive K tiles of one chunk (set by launch_cfg): BK, or chunks * BK when the chunks interleave
    int stagger;      // experiment (MC_IGEMM_STAGGER=n): the second half of the grid's workgroups starts n x 8128 cycles late
};

struct Epi {
    float* C;
    int ldc;
    int rowG, colG;   // per-group row / column offset
    long chunkStride; // per k-chunk offset (slabs)
    float alpha, beta;
    const float* bias;  // per output row (after group offset), nullable
    int act;            // MC_NEURON_*
    const float* mask;  // nullable: outputs whose mask element is not > 0 are written as 0 (relu gradient)
    int ldmask;
    unsigned cBytes;    // extent of one chunk's output in bytes: range of the store descriptor
    unsigned mBytes;    // extent of the mask in bytes
    // relu masks as bits, one per output element, in the layout of the MFMA accumulators (mc_kernels.h, "relu bit masks"):
    // a 32-bit word per (32-row block, column), word (i >> 5) * ldc + j; bit 16 * h + r <-> row (i & ~31) + (r & 3) + 8 * (r >> 2) + 4 * h,
    // i.e. each lane's 16 accumulators of a 32x32 block are one 16-bit half word: no cross-lane traffic to write or to read.
    unsigned* bitsOut;        // nullable: written by the epilogue, bit = (output > 0)
    const unsigned* bitsIn;   // EPI_BITS kernels: outputs whose bit is clear are written as 0 (relu gradient)
    unsigned wBytes;          // extent of the bit matrix in bytes
    // Sum of one operand over k, a by-product of kernels whose A or B loader is tagged SUM (weight gradients: the operand is
    // the layer's output gradient, its sum over k the bias gradient): chunk q writes sumAlpha * sum_k op(k, x) to
    // sumOut[q * sumStride + x + g * (rowG | colG)].  Nullable.  32x32 tiles: each consumer lane adds the k values it feeds to the MFMAs
    // anyway, in k order (tile-independent among them).  16x16x4 tiles: the producer waves add the chunks they stage (Loader::add_sums),
    // another fixed order: with sums asked for, the two families are chosen between by rule, never by timing.
    float* sumOut;
    long sumStride;
    float sumAlpha;
};
// Epilogue flavours (a template parameter: as run-time branches inside one kernel the variants that load cost every
// instantiation 40-80 registers, i.e. a resident workgroup per CU):
//   EPI_PLAIN  alpha * acc (+ bias) (relu) (+ bit-mask output); no loads.  The hot path of every product.
//   EPI_SLOW   anything: scaleTargets != 0, float masks.  One element at a time, few registers, slow (200+ cycles per
//              store instruction): accumulating calls and odd shapes only.
//   EPI_BITS   EPI_PLAIN with the relu-gradient mask read as bits (dense products only).
//   EPI_PLAIN_BITSOUT  EPI_PLAIN that also writes the bit mask (output > 0) of what it stores (relu forward products whose
//              gradient mask a dense backward product reads).
enum { EPI_PLAIN = 0, EPI_SLOW = 1, EPI_BITS = 2, EPI_PLAIN_BITSOUT = 3 };
// (batch b of a batched launch writes columns [b * d.colB, b * d.colB + d.Ng) of C)

// exact floor(a / d) for 0 <= a < 2^22 using a precomputed float reciprocal
__device__ __forceinline__ int fdiv(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// ---------------------------------------------------------------------------------
// Element address functions.  Each returns the element offset of operand element
// (k, x) and whether it is a real element (false -> contributes 0).
// ---------------------------------------------------------------------------------
struct MatP {
    const float* p;
    int ld;      // row stride of the stored matrix
    int R;       // extent of the non-k axis (per group)
    int gStride; // offset added per group
    unsigned bytes;  // extent of the whole operand (buffer descriptor range: loads beyond it return 0)
    int bStride; // offset added per batch (unshared layers: one filter bank per module)
};

// stored (K, R): element (k, x) at k*ld + x
struct MatNatAddr {
    static const char* tag() { return "MatNatAddr"; }
    typedef MatP P;
    static constexpr bool KC = false;
    __device__ static __forceinline__ bool addr(const P& p, const ConvG&, int g, int b, int k, int x, int& off) {
        off = g * p.gStride + b * p.bStride + k * p.ld + x;
        return x < p.R;
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};
// stored (R, K): element (k, x) at x*ld + k
struct MatKcAddr {
    static const char* tag() { return "MatKcAddr"; }
    typedef MatP P;
    static constexpr bool KC = true;
    __device__ static __forceinline__ bool addr(const P& p, const ConvG&, int g, int b, int k, int x, int& off) {
        off = g * p.gStride + b * p.bStride + x * p.ld + k;
        return x < p.R;
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};

// MatKcAddr whose sum over k the kernel also delivers (Epi::sumOut)
struct MatKcSumAddr : MatKcAddr {
    static const char* tag() { return "MatKcSumAddr"; }
    static constexpr bool SUM = true;
};
template <class T, class = void>
struct has_ktab : std::false_type {};
template <class T>
struct has_ktab<T, std::void_t<decltype(T::KTAB)>> : std::true_type {};
constexpr int KTAB_MAX = 4096;   // k extents up to this are tabulated (32 KB of LDS); longer ones keep the arithmetic decode
template <class T, class = void>
struct has_sum : std::false_type {};
template <class T>
struct has_sum<T, std::void_t<decltype(T::SUM)>> : std::true_type {};

struct ConvP {
    const float* p;
    int R;  // extent of the non-k axis
    unsigned bytes;  // extent of the whole tensor (buffer descriptor range)
};

// im2col patch element: filter tap kc = (c, fy, fx), column (m, n)
__device__ __forceinline__ bool patch_addr(const ConvG& cg, int g, int kc, int m, int n, int& off) {
    const int c = fdiv(kc, cg.inv_fp);
    const int p = kc - c * cg.fp;
    const int fy = fdiv(p, cg.inv_fs);
    const int fx = p - fy * cg.fs;
    const int my = fdiv(m, cg.inv_MX);
    const int mx = m - my * cg.MX;
    const int y = cg.pad + my * cg.stride + fy;
    const int x = cg.pad + mx * cg.stride + fx;
    off = (((g * cg.fc + c) * cg.H + y) * cg.W + x) * cg.N + n;
    return (unsigned)y < (unsigned)cg.H && (unsigned)x < (unsigned)cg.W;
}

// forward B operand: k = filter tap, x = column (m*N + n); NAT (n contiguous)
// K tables.  The gather address of these operands separates into a column part and a k part,
//   offset(k, x) = col(x).base + entry(k).x,   valid iff (unsigned)(col.cy + dy) < limY and (unsigned)(col.cx + dx) < limX
// with (dy, dx) packed in entry(k).y.  The column part is loop invariant in the K loop; the k part -- two float-reciprocal
// divisions, three multiplies per element when done arithmetically, ~30 instructions per 16-byte load, which is what bounds
// the producer waves of products with few rows (MC_STAMP: 820 of the 1100 cycles of a K step of layers-80sec conv2) -- is
// tabulated in LDS once per workgroup (8 bytes per k) and costs one ds_read_b64.
struct KCol { int base, cy, cx; };   // a column outside the operand gets cy = KCOL_DEAD: its range check fails for every k
constexpr int KCOL_DEAD = -(1 << 30);
struct PatchNatAddr {
    static const char* tag() { return "PatchNatAddr"; }
    typedef ConvP P;
    static constexpr bool KC = false;
    static constexpr bool KTAB = true;
    __device__ static __forceinline__ int2 ktab_entry(const ConvG& cg, int k) {
        const int c = fdiv(k, cg.inv_fp), p = k - c * cg.fp;
        const int fy = fdiv(p, cg.inv_fs), fx = p - fy * cg.fs;
        return make_int2(((c * cg.H + fy) * cg.W + fx) * cg.N, (fy & 0xFFFF) | (fx << 16));
    }
    __device__ static __forceinline__ KCol ktab_col(const P& p, const ConvG& cg, int g, int x) {
        const int m = x / cg.N, n = x - m * cg.N;
        const int my = fdiv(m, cg.inv_MX), mx = m - my * cg.MX;
        KCol c;
        c.cy = cg.pad + my * cg.stride;
        c.cx = cg.pad + mx * cg.stride;
        c.base = ((g * cg.fc * cg.H + c.cy) * cg.W + c.cx) * cg.N + n;
        if (x >= p.R) c.cy = KCOL_DEAD;
        return c;
    }
    __device__ static __forceinline__ int ktab_limY(const ConvG& cg) { return cg.H; }
    __device__ static __forceinline__ int ktab_limX(const ConvG& cg) { return cg.W; }
    __device__ static __forceinline__ bool addr(const P& p, const ConvG& cg, int g, int b, int k, int x, int& off) {
        const int m = x / cg.N;  // column decode is loop invariant in the K loop: exact integer division
        const int n = x - m * cg.N;
        return patch_addr(cg, g, k, m, n, off) & (x < p.R);
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};
// weight-gradient A operand: x = filter tap (row), k = (m*N + n); KC (n contiguous)
struct PatchKcAddr {
    static const char* tag() { return "PatchKcAddr"; }
    typedef ConvP P;
    static constexpr bool KC = true;
    // When a K tile (k = module * N + image) never straddles two modules, everything that depends on the module
    // is the same for the whole tile: computed once from the tile's first k (workgroup-uniform), it leaves the
    // per-element work at two adds and two range checks.  The row's own (channel, tap) decode does not depend
    // on k at all and is hoisted out of the K loop by the compiler.
    struct Tile { int y0, x0, base; };
    __device__ static __forceinline__ Tile tile(const ConvG& cg, int k0) {
        const int m = fdiv(k0, cg.inv_N);
        const int my = fdiv(m, cg.inv_MX), mx = m - my * cg.MX;
        Tile t;
        t.y0 = cg.pad + my * cg.stride;
        t.x0 = cg.pad + mx * cg.stride;
        t.base = (t.y0 * cg.W + t.x0) * cg.N + (k0 - m * cg.N);
        return t;
    }
    __device__ static __forceinline__ bool addr_in_tile(const P& p, const ConvG& cg, int g, const Tile& t, int dk, int x, int& off) {
        const int xx = x < p.R ? x : 0;
        const int c = fdiv(xx, cg.inv_fp);
        const int pp = xx - c * cg.fp;
        const int fy = fdiv(pp, cg.inv_fs);
        const int fx = pp - fy * cg.fs;
        off = (((g * cg.fc + c) * cg.H + fy) * cg.W + fx) * cg.N + t.base + dk;
        return ((unsigned)(t.y0 + fy) < (unsigned)cg.H) & ((unsigned)(t.x0 + fx) < (unsigned)cg.W) & (x < p.R);
    }
    __device__ static __forceinline__ bool addr(const P& p, const ConvG& cg, int g, int b, int k, int x, int& off) {
        const int m = fdiv(k, cg.inv_N);
        const int n = k - m * cg.N;
        const int xx = x < p.R ? x : 0;
        return patch_addr(cg, g, xx, m, n, off) & (x < p.R);
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};
// bwd-data B operand: k = (p, fl) (filter pixel, filter within group), x = (pixel*N + n); NAT
struct HidNatAddr {
    static const char* tag() { return "HidNatAddr"; }
    typedef ConvP P;
    static constexpr bool KC = false;
    __device__ static __forceinline__ bool addr(const P& p, const ConvG& cg, int g, int b, int k, int x, int& off) {
        const int px = x / cg.N;  // loop invariant in the K loop
        const int n = x - px * cg.N;
        const int y = px / cg.W;
        const int xx = px - y * cg.W;
        const int fpix = fdiv(k, cg.inv_fpg);
        const int fl = k - fpix * cg.fpg;
        const int fy = fdiv(fpix, cg.inv_fs);
        const int fx = fpix - fy * cg.fs;
        int ty = y - cg.pad - fy, tx = xx - cg.pad - fx;
        bool ok = (ty >= 0) & (tx >= 0) & (x < p.R);
        if (cg.stride != 1) {   // exact for 0 <= t < 2^22 (fdiv); negative t is already rejected by ok
            const int my = fdiv(max(ty, 0), cg.inv_stride), mx = fdiv(max(tx, 0), cg.inv_stride);
            ok &= (my * cg.stride == ty) & (mx * cg.stride == tx);
            ty = my;
            tx = mx;
        }
        ok &= (ty < cg.MY) & (tx < cg.MX);
        off = ((g * cg.fpg + fl) * cg.M + ty * cg.MX + tx) * cg.N + n;
        return ok;
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};
// the same operand for moduleStride == 1 (every BASELINE layer that back-propagates to its input): no stride
// division at all, 24-bit multiplies, range checks as unsigned compares -- the gather address arithmetic is what
// bounds the producer waves of the backward-data product
struct HidNatS1Addr {
    static const char* tag() { return "HidNatS1Addr"; }
    typedef ConvP P;
    static constexpr bool KC = false;
    static constexpr bool KTAB = true;   // (see PatchNatAddr)
    __device__ static __forceinline__ int2 ktab_entry(const ConvG& cg, int k) {
        const int fpix = fdiv(k, cg.inv_fpg), fl = k - fpix * cg.fpg;
        const int fy = fdiv(fpix, cg.inv_fs), fx = fpix - fy * cg.fs;
        return make_int2((fl * cg.M - fy * cg.MX - fx) * cg.N, (int)(((unsigned)(-fy) & 0xFFFFu) | ((unsigned)(-fx) << 16)));
    }
    __device__ static __forceinline__ KCol ktab_col(const P& p, const ConvG& cg, int g, int x) {
        const int px = x / cg.N, n = x - px * cg.N;
        const int y = px / cg.W, xx = px - y * cg.W;
        KCol c;
        c.cy = y - cg.pad;
        c.cx = xx - cg.pad;
        c.base = (g * cg.fpg * cg.M + c.cy * cg.MX + c.cx) * cg.N + n;
        if (x >= p.R) c.cy = KCOL_DEAD;
        return c;
    }
    __device__ static __forceinline__ int ktab_limY(const ConvG& cg) { return cg.MY; }
    __device__ static __forceinline__ int ktab_limX(const ConvG& cg) { return cg.MX; }
    __device__ static __forceinline__ bool addr(const P& p, const ConvG& cg, int g, int b, int k, int x, int& off) {
        const int px = x / cg.N;  // loop invariant in the K loop
        const int n = x - px * cg.N;
        const int y = px / cg.W;
        const int xx = px - y * cg.W;
        const int fpix = fdiv(k, cg.inv_fpg);
        const int fl = k - __mul24(fpix, cg.fpg);
        const int fy = fdiv(fpix, cg.inv_fs);
        const int fx = fpix - __mul24(fy, cg.fs);
        const int ty = y - cg.pad - fy, tx = xx - cg.pad - fx;
        const bool ok = ((unsigned)ty < (unsigned)cg.MY) & ((unsigned)tx < (unsigned)cg.MX) & (x < p.R);
        off = (__mul24(g * cg.fpg + fl, cg.M) + __mul24(ty, cg.MX) + tx) * cg.N + n;
        return ok;
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};
// bwd-data A operand: x = channel within group (row), k = (p, fl); filters[(x*fp + p)*F + g*fpg + fl]; KC
struct FiltTAddr {
    static const char* tag() { return "FiltTAddr"; }
    typedef ConvP P;
    static constexpr bool KC = true;
    __device__ static __forceinline__ bool addr(const P& p, const ConvG& cg, int g, int b, int k, int x, int& off) {
        const int fpix = fdiv(k, cg.inv_fpg);
        const int fl = k - fpix * cg.fpg;
        if (!cg.local) {
            off = (x * cg.fp + fpix) * cg.F + g * cg.fpg + fl;
            return x < p.R;
        }
        // unshared filters (img_acts.cu conv=false): batch b is the image pixel; tap fpix of that pixel belongs to
        // the module whose window holds the pixel at that tap, and each module has its own filter bank
        const int y = b / cg.W, xx = b - y * cg.W;
        const int fy = fdiv(fpix, cg.inv_fs), fx = fpix - fy * cg.fs;
        int ty = y - cg.pad - fy, tx = xx - cg.pad - fx;
        bool ok = (ty >= 0) & (tx >= 0) & (x < p.R);
        if (cg.stride != 1) {
            const int my = fdiv(max(ty, 0), cg.inv_stride), mx = fdiv(max(tx, 0), cg.inv_stride);
            ok &= (my * cg.stride == ty) & (mx * cg.stride == tx);
            ty = my;
            tx = mx;
        }
        ok &= (ty < cg.MY) & (tx < cg.MX);
        off = (((ty * cg.MX + tx) * cg.fc + x) * cg.fp + fpix) * cg.F + g * cg.fpg + fl;
        return ok;
    }
    __device__ static __forceinline__ const float* base(const P& p) { return p.p; }
};

// ---------------------------------------------------------------------------------
// Tile loader (producer waves): global -> registers (fetch) -> LDS (commit).
//   ADDR::KC == false: LDS [BK][ROWS] floats, chunk = 4 consecutive x at one k
//   ADDR::KC == true : LDS [BK/4][ROWS+1] float4, chunk = 4 consecutive k at one x
// VEC == 4: one 16-byte buffer load per chunk (caller guarantees alignment and that a chunk never
// straddles a validity boundary); VEC == 1: four 4-byte buffer loads.  Predication is branch-free:
// an element that contributes 0 (padding tap, k tail, row/column tail) is loaded at offset
// 0xFFFFFFFF, which the buffer descriptor's range check turns into a 0 without touching memory.
// ---------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFFFu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}

// PERM (k-contiguous operands of the 16x16x4 tiles): the four k values of a 16-byte chunk are stored as (k0, k2 | k1, k3) in even
// chunks of a row and (k1, k3 | k0, k2) in odd ones, so that a consumer lane reads the two values it feeds to a group's two matrix
// instructions with ONE ds_read_b64 and no select (see the MF consumer).
// TR (operands contiguous along their row axis, "NAT"): the tile is TRANSPOSED on its way into LDS, so that the consumers read
// the four k values of a row's k-quad with one ds_read_b128 -- four slabs per read, as for the k-contiguous operands -- instead
// of one ds_read_b32 per slab.  (Timing ablation MC_ABL_NOLDSREAD: without their fragment reads the consumers of NIN's forward
// products finish 8-14 % sooner -- conv2 forward 271 -> 240 us -- so a read instruction saved is matrix-pipe time won.)  A thread
// gathers a UNIT = 4 consecutive k x 4 consecutive x (four 16-byte loads, lanes running along x: coalesced as before) and
// writes four (k0..k3) quads.  LDS float4 index of (k-quad kq, row x): kq * 4 QP + (x & 3) * QP + (x >> 2) -- the rows are
// dealt out over four quarters so that the lanes of a wave, which hold consecutive x-quads, write consecutive float4s (no bank
// conflict: ds_write_b128 is served in groups of 8 contiguous lanes), and the quarter pitch QP = 4 (mod 8) float4s makes the
// consumers' reads conflict-free too: ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32)
// over 16 slots of 16 bytes; lane l31 reads slot (l31 & 3) QP + (l31 >> 2) + const, and with QP = 4 or 12 (mod 16) the sixteen
// (l31 & 3, l31 >> 2) pairs of a group land on sixteen different slots (MI355X_MICROARCH.md, LDS; QP = 2 (mod 8) was measured
// first: SQ_LDS_BANK_CONFLICT 30 % of the LDS cycles, every product slower than with ds_read_b32).
template <class ADDR, int ROWS, int BK, int VEC, bool PERM = false, bool TR = false>
struct Loader {
    typedef typename ADDR::P P;
    typedef ADDR ADDR_T;
    static constexpr int VEC_W = VEC, ROWS_N = ROWS, BK_N = BK;
    static constexpr bool PERM_F = PERM;
    static_assert(!TR || (!ADDR::KC && !PERM && BK % 4 == 0 && ROWS % 32 == 0), "the transposing form is for row-contiguous operands");
    static constexpr bool KC = ADDR::KC;
    static constexpr bool TRANSPOSED = TR;
    static constexpr bool LKC = KC || TR;   // LDS holds k-quads per row (read with b128 / b64), not [k][rows]
    static constexpr int ROWSP = ROWS + 1;
    static constexpr int QP = (ROWS / 4 + 3) / 8 * 8 + 4;   // the least QP >= ROWS / 4 with QP % 8 == 4
    static constexpr int LDS_FLOATS_TR = (BK / 4) * 4 * QP * 4;   // (the pair form's footprint is set below, once HP is known)
    static constexpr int CHUNKS = BK * ROWS / 4;
    static constexpr int UNITS = (BK / 4) * (ROWS / 4);
    static constexpr int NU = (UNITS + THREADS - 1) / THREADS;
#ifdef MC_TR_UNITS   // first form (measured, slower): a thread gathers a 4 k x 4 x unit and writes four transposed 16-byte records
    static constexpr bool TRU = TR;
#else                // second form: the chunks and their threads stay as they are; a chunk (one k, four x) is written as four 4-byte stores
    static constexpr bool TRU = false;
#endif
#ifdef MC_TR_PAIR    // third form: a thread gathers 4 consecutive k x TWO consecutive x (four 8-byte loads: a wave's load is still one
                     // contiguous 512-byte run of a k row) and writes two 16-byte records.  Register ring, threads and store count
                     // are those of the plain form; twice the load instructions and address arithmetic.
    static constexpr bool TRP = TR;
#else
    static constexpr bool TRP = false;
#endif
    static_assert(!(TRU && TRP), "one transposing form at a time");
    // rows dealt over two halves of pitch HP records (even rows, odd rows): lane L of a wave writes rows 2L and 2L + 1 -> consecutive
    // records in each half (8 contiguous lanes per ds_write_b128 group: no conflict); HP = 8 (mod 16) makes the consumers' 16-lane
    // ds_read_b128 groups conflict-free: lane l31 reads record (l31 & 1) HP + (l31 >> 1) + const
    static constexpr int HP = (ROWS / 2 + 7) / 16 * 16 + 8;
    static constexpr int UP = (BK / 4) * (ROWS / 2);
    static constexpr int NUP = (UP + THREADS - 1) / THREADS;
    static constexpr int NCH = TRU ? 4 * NU : TRP ? 2 * NUP : (CHUNKS + THREADS - 1) / THREADS;
    static constexpr int NCOL = TRU ? NU : TRP ? NUP : NCH;   // distinct columns among a thread's chunks (K-table loaders)
    static constexpr int LDS_FLOATS = TRP ? (BK / 4) * 2 * HP * 4 : TR ? LDS_FLOATS_TR : KC ? (BK / 4) * ROWSP * 4 : BK * ROWS;
    // (the chunk registers r[NCH] belong to the caller: a producer thread that serves ONE operand keeps one set for whichever it is)

    // register `it` of thread tid -> (k offset, x offset) of its chunk inside the tile, and whether the chunk exists
    __device__ static __forceinline__ void pos(int it, int tid, int& dk, int& dx, bool& live) {
        if (TRU) {
            const int u = tid + (it >> 2) * THREADS;
            live = UNITS % THREADS == 0 || u < UNITS;
            dk = (u / (ROWS / 4)) * 4 + (it & 3);
            dx = (u % (ROWS / 4)) * 4;
        } else if (TRP) {   // it = 2 * unit: the unit's first k and its x pair
            const int u = tid + (it >> 1) * THREADS;
            live = UP % THREADS == 0 || u < UP;
            dk = (u / (ROWS / 2)) * 4;
            dx = (u % (ROWS / 2)) * 2;
        } else {
            const int c = tid + it * THREADS;
            live = CHUNKS % THREADS == 0 || c < CHUNKS;
            chunk_pos(c, dk, dx);
        }
    }

    // chunk c -> (k offset inside tile, x offset inside tile) of its first element
    __device__ static __forceinline__ void chunk_pos(int c, int& dk, int& dx) {
        if (KC) {
            dk = (c % (BK / 4)) * 4;
            dx = c / (BK / 4);
        } else {
            dk = c / (ROWS / 4);
            dx = (c % (ROWS / 4)) * 4;
        }
    }

    // K-table loaders, vector path: the column part of every chunk this thread gathers, decoded once per tile (two integer
    // divisions each -- left inside fetch() the compiler re-ran them every K step)
    __device__ static __forceinline__ void prep_cols(KCol* cols, const P& p, const ConvG& cg, int g, int x0, int xlim, int tid) {
        if constexpr (has_ktab<ADDR>::value) {
#pragma unroll
            for (int ic = 0; ic < NCOL; ++ic) {
                int dk, dx;
                bool live;
                pos(TRU ? 4 * ic : TRP ? 2 * ic : ic, tid, dk, dx, live);
                const int x = x0 + dx;
                KCol col = ADDR::ktab_col(p, cg, g, x);
                if (!((x < xlim) & live)) col.cy = KCOL_DEAD;
                cols[ic] = col;
            }
        }
    }

    // TAB: the K-table form (a compile-time choice: a run-time branch here would sit inside the producers' K loop, where the
    // compiler's vmcnt bookkeeping must stay exact -- with the branch it waited for vmcnt(3) instead of vmcnt(6) before every
    // commit and drained to vmcnt(0) in the arithmetic path, i.e. one tile in flight instead of three)
    template <bool TAB = false>
    __device__ static __forceinline__ void fetch(float4* r, __amdgpu_buffer_rsrc_t rs, const P& p, const ConvG& cg, int g, int b, int k0, int kend,
                                          int x0, int xlim, int tid, const int2* ktab = nullptr, const KCol* cols = nullptr) {
#ifdef MC_ABL_NOFETCH   // timing-only ablation builds (tools/build_variant.sh); wrong results
        if (k0 > 0) return;
#endif
        if constexpr (TRP) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int iu = 0; iu < NUP; ++iu) {
                int dk, dx;
                bool live;
                pos(2 * iu, tid, dk, dx, live);
                const int x = x0 + dx;
                unsigned off[4];
                if constexpr (has_ktab<ADDR>::value && TAB) {
                    const unsigned limY = (unsigned)ADDR::ktab_limY(cg), limX = (unsigned)ADDR::ktab_limX(cg);
                    int2 ent[4];
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) ent[sub] = ktab[k0 + dk + sub];   // the table is padded to a whole K tile
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) asm volatile("" : "+v"(ent[sub].x), "+v"(ent[sub].y));
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const int dy = (int)(short)(ent[sub].y & 0xFFFF), dxx = ent[sub].y >> 16;
                        const bool ok = ((unsigned)(cols[iu].cy + dy) < limY) & ((unsigned)(cols[iu].cx + dxx) < limX) & (k0 + dk + sub < kend);
                        off[sub] = ok ? (unsigned)(cols[iu].base + ent[sub].x) * 4u : OOB;
                    }
                } else {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        int o;
                        const bool ok = ADDR::addr(p, cg, g, b, k0 + dk + sub, x, o) & (k0 + dk + sub < kend) & (x < xlim) & live;
                        off[sub] = ok ? (unsigned)o * 4u : OOB;
                    }
                }
                const u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(rs, off[0], 0, 0), v1 = __builtin_amdgcn_raw_buffer_load_b64(rs, off[1], 0, 0);
                const u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(rs, off[2], 0, 0), v3 = __builtin_amdgcn_raw_buffer_load_b64(rs, off[3], 0, 0);
                r[2 * iu] = make_float4(__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v1.x), __uint_as_float(v1.y));
                r[2 * iu + 1] = make_float4(__uint_as_float(v2.x), __uint_as_float(v2.y), __uint_as_float(v3.x), __uint_as_float(v3.y));
            }
            return;
        }
        if constexpr (has_ktab<ADDR>::value && TAB) {
            {   // the k part of the address comes from the table (see KCol)
                const unsigned limY = (unsigned)ADDR::ktab_limY(cg), limX = (unsigned)ADDR::ktab_limX(cg);
                // all table reads first, as whole 8-byte entries, and pinned: left to itself the compiler sinks the read of the
                // offset half into the "element is valid" branch -- two dependent LDS round trips and an exec-mask dance per chunk
                int2 ent[NCH];
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    int dk, dx;
                    bool live;
                    pos(it, tid, dk, dx, live);
                    ent[it] = ktab[k0 + dk];   // the table is padded to a whole K tile
                }
#pragma unroll
                for (int it = 0; it < NCH; ++it) asm volatile("" : "+v"(ent[it].x), "+v"(ent[it].y));
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    int dk, dx;
                    bool live;
                    pos(it, tid, dk, dx, live);
                    const int k = k0 + dk;
                    const int2 e = ent[it];
                    const int dy = (int)(short)(e.y & 0xFFFF), dxx = e.y >> 16;
                    if (VEC == 4) {   // the column part was decoded once per tile (prep_cols)
                        const KCol& cl = cols[TRU ? it >> 2 : it];
                        const bool ok = ((unsigned)(cl.cy + dy) < limY) & ((unsigned)(cl.cx + dxx) < limX) & (k < kend);
                        unsigned off = (unsigned)(cl.base + e.x) * 4u;
                        off = ok ? off : OOB;
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                        r[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                    } else {
                        const int x = x0 + dx;
                        float ev[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const KCol col = ADDR::ktab_col(p, cg, g, x + u);
                            const bool ok = ((unsigned)(col.cy + dy) < limY) & ((unsigned)(col.cx + dxx) < limX) & (k < kend) & (x + u < xlim) & live;
                            ev[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)(col.base + e.x) * 4u : OOB, 0, 0));
                        }
                        r[it] = make_float4(ev[0], ev[1], ev[2], ev[3]);
                    }
                }
                return;
            }
        }
        if constexpr (std::is_same<ADDR, PatchKcAddr>::value) {
            if (cg.kcUniform) {   // workgroup-uniform branch
                const PatchKcAddr::Tile t = PatchKcAddr::tile(cg, k0);
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const int c = tid + it * THREADS;
                    const bool live = CHUNKS % THREADS == 0 || c < CHUNKS;
                    int dk, dx;
                    chunk_pos(c, dk, dx);
                    const int k = k0 + dk, x = x0 + dx;
                    if (VEC == 4) {
                        int off;
                        const bool ok = PatchKcAddr::addr_in_tile(p, cg, g, t, dk, x, off) & (k < kend) & (x < xlim) & live;
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)off * 4u : OOB, 0, 0);
                        r[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                    } else {
                        float e[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            int off;
                            const bool ok = PatchKcAddr::addr_in_tile(p, cg, g, t, dk + u, x, off) & (k + u < kend) & (x < xlim) & live;
                            e[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)off * 4u : OOB, 0, 0));
                        }
                        r[it] = make_float4(e[0], e[1], e[2], e[3]);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            int dk, dx;
            bool live;
            pos(it, tid, dk, dx, live);
            const int k = k0 + dk, x = x0 + dx;
            if (VEC == 4) {
                int off;
                const bool ok = ADDR::addr(p, cg, g, b, k, x, off) & (k < kend) & (x < xlim) & live;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)off * 4u : OOB, 0, 0);
                r[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ku = KC ? k + u : k, xu = KC ? x : x + u;
                    int off;
                    const bool ok = ADDR::addr(p, cg, g, b, ku, xu, off) & (ku < kend) & (xu < xlim) & live;
                    e[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)off * 4u : OOB, 0, 0));
                }
                r[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }

    // k-sums of a k-contiguous operand (weight gradients: the operand is the layer's output gradient, its sum over k the bias
    // gradient): each producer thread adds the four k values of its chunks, in k order, to one partial per chunk slot; a row's
    // BK / 4 partials sit in consecutive lanes and are combined when the K loop is over (sum_rows).  The order depends on BK and on
    // the K chunks only, not on the tile.  The producers have the time: the consumers' matrix instructions do not overlap with
    // their own vector instructions (tools/census/mfma16_probe.hip: 2 VALU per 16x16x4 MFMA cost 6.8 of its 32 cycles).
    __device__ static __forceinline__ void add_sums(const float4* r, float* ps) {
        static_assert(KC, "k-sums come with k-contiguous operands");
#pragma unroll
        for (int it = 0; it < NCH; ++it) { ps[it] += r[it].x; ps[it] += r[it].y; ps[it] += r[it].z; ps[it] += r[it].w; }
    }
    // out[x0 + row] = alpha * (sum of the row's partials), rows < xlim; every lane of the wave must call
    __device__ static __forceinline__ void sum_rows(const float* ps, float* out, float alpha, int x0, int xlim, int tid) {
        static_assert(BK / 4 == 8, "a row's partials are combined over 8 consecutive lanes");
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            float v = ps[it];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const int c = tid + it * THREADS;
            const int x = x0 + c / (BK / 4);
            if ((CHUNKS % THREADS == 0 || c < CHUNKS) && (c % (BK / 4)) == 0 && x < xlim) out[x] = alpha * v;
        }
    }
    __device__ static __forceinline__ void commit(const float4* r, float* lds, int tid) {
#ifdef MC_ABL_NOCOMMIT
        return;
#endif
#ifdef MC_ABL_NOWRITE   // timing-only ablation: the loads are waited for and their registers consumed, nothing is written to LDS
#pragma unroll
        for (int it = 0; it < NCH; ++it) asm volatile("" ::"v"(r[it].x), "v"(r[it].y), "v"(r[it].z), "v"(r[it].w));
        return;
#endif
        float4* l4 = reinterpret_cast<float4*>(lds);
        if constexpr (TRP) {
#pragma unroll
            for (int iu = 0; iu < NUP; ++iu) {
                int dk, dx;
                bool live;
                pos(2 * iu, tid, dk, dx, live);
                float4 a = r[2 * iu], b = r[2 * iu + 1];   // (k0: x, x+1 | k1: x, x+1), (k2 | k3)
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));   // regroup HERE, not behind the loads
                if (live) {
                    float4* q = l4 + (dk >> 2) * 2 * HP + (dx >> 1);
                    q[0] = make_float4(a.x, a.z, b.x, b.z);
                    q[HP] = make_float4(a.y, a.w, b.y, b.w);
                }
            }
            return;
        }
        if constexpr (TR && !TRU) {
            // chunk (k, x .. x+3) -> component k & 3 of the records of rows x .. x+3 of k-quad k >> 2: four ds_write_b32 a quarter
            // pitch apart (immediate offsets); no register is regrouped, the threads and their loads are those of the plain form
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                int dk, dx;
                bool live;
                pos(it, tid, dk, dx, live);
                if (live) {
                    float* q = lds + (((dk >> 2) * 4 * QP + (dx >> 2)) << 2) + (dk & 3);
                    q[0] = r[it].x;
                    q[4 * QP] = r[it].y;
                    q[8 * QP] = r[it].z;
                    q[12 * QP] = r[it].w;
                }
            }
            return;
        }
        if constexpr (TRU) {
#pragma unroll
            for (int iu = 0; iu < NU; ++iu) {
                int dk, dx;
                bool live;
                pos(4 * iu, tid, dk, dx, live);
                // The regrouping below (four registers of four different loads into one 16-byte store) is register copies.  Left
                // alone the compiler schedules them right behind the loads -- and with them a wait for the loads just issued, every
                // K step: one tile in flight instead of three.  The empty asm "rewrites" the loaded values HERE, so whatever
                // reads them, copies included, stays behind this point.
                float4 a = r[4 * iu], b = r[4 * iu + 1], c = r[4 * iu + 2], d = r[4 * iu + 3];   // k0 .. k3 at x .. x+3
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
                asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
                if (live) {
                    float4* q = l4 + (dk >> 2) * 4 * QP + (dx >> 2);
                    q[0] = make_float4(a.x, b.x, c.x, d.x);
                    q[QP] = make_float4(a.y, b.y, c.y, d.y);
                    q[2 * QP] = make_float4(a.z, b.z, c.z, d.z);
                    q[3 * QP] = make_float4(a.w, b.w, c.w, d.w);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int c = tid + it * THREADS;
            if (CHUNKS % THREADS == 0 || c < CHUNKS) {
                if (KC) {
                    float4 v = r[it];
                    if constexpr (PERM) {   // component-wise selects (as one ternary over the struct the compiler branches, inside the producers' K loop)
                        const bool odd = (c & 1) != 0;   // = tid & 1: THREADS is even
                        v.x = odd ? r[it].y : r[it].x;
                        v.y = odd ? r[it].w : r[it].z;
                        v.z = odd ? r[it].x : r[it].y;
                        v.w = odd ? r[it].z : r[it].w;
                    }
                    l4[(c % (BK / 4)) * ROWSP + c / (BK / 4)] = v;
                } else {
                    l4[c] = r[it];  // [dk][dx/4] == linear chunk order
                }
            }
        }
    }
};

// LDS stores of this wave have landed, then the workgroup barrier.  Deliberately NOT __syncthreads():
// that also waits for vmcnt(0), which would drain the producers' in-flight tile loads at every K step.
__device__ __forceinline__ void lds_done_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------
// The kernel.  512 threads: waves 0-3 are consumers laid out WM x WN, wave tile (32*MT) x (32*NT);
// waves 4-7 are producers (THREADS = 256 of them share the tile gather).
// ---------------------------------------------------------------------------------
// (Asking for more resident workgroups per CU through __launch_bounds__' second argument was measured: the
// register cap spills in the k-contiguous loaders and no shape gets faster.)
// KQ = 2: the four consumer waves are WM x WN = 2 wave tiles x 2 halves of every K step (wave (w, kq) multiplies slabs
// [kq * SLABS / 2, (kq + 1) * SLABS / 2) of each LDS tile); at the end of the tile the kq = 1 waves hand their accumulators
// over through LDS and the kq = 0 waves add and store.  That makes a 96 x 192 workgroup tile possible (two 96 x 96 wave
// tiles), which fits the dense / conv weight gradients with 75-192 rows and 160-192 columns that every 4-tile shape pads by
// 25-40 %.  The sum of an output element is (first halves of all K steps) + (second halves): a fixed order, but not the order
// of the KQ = 1 tiles, so this tile is selected by rule only (a pure function of the shape), never by the timing autotuner.
// MF = 1: the consumer waves multiply with v_mfma_f32_16x16x4_f32 and MT / NT count 16-row / 16-column blocks of a wave tile.
// With 2 x 2 waves that gives workgroup tiles of 96 / 160 / 192 rows by 160 / 192 columns: a dense or first-layer weight gradient
// whose whole output is 75-192 rows by 160-192 columns (NIN's cccp layers and conv1) is ONE tile per K chunk -- no padded matrix
// work (the 32x32 tiles pad those products by 25-40 %) and each operand element is fetched once device-wide.  Both operands
// k-contiguous, plain epilogue.  The k values of an 8-group go to the matrix pipe in the order of the 32x32x2 path
// (0,4,1,5,2,6,3,7: lane group q of MFMA m takes k = 4 (q & 1) + 2 m + (q >> 1)), and the result of either instruction is one
// k-ordered fma chain, so these tiles give the same bits as every other tile for the same K chunks.
// Which row-contiguous operands take the transposing loader: all, except in the 160-row tile -- that one is held at 128
// registers for two resident workgroups per CU and has none to spare for four chunks per producer thread and operand -- and only on the vector path (the scalar path,
// odd batch sizes, would take 175 registers).
#ifndef MC_TR_NAT   // off by default: see the measurements in DESIGN.md (the consumers gain 11 %, the producers lose more)
template <class ADDR> constexpr bool tr_nat(int, int, int) { return false; }
#else
template <class ADDR> constexpr bool tr_nat(int rows, int other, int vec) { return !ADDR::KC && vec == 4 && rows % 32 == 0 && rows != 160 && other != 160; }
#endif
template <class ADDR, int ROWS, int OTHER, int BK, int VEC, bool PERM = false>
using LoaderT = Loader<ADDR, ROWS, BK, VEC, PERM, tr_nat<ADDR>(ROWS, OTHER, VEC)>;

// LDS-DMA loader of operand L (the products n_stages() names; vector path): a chunk goes from memory straight to its place in
// the stage -- buffer_load_dwordx4 ... lds writes lane l's 16 bytes at (wave-uniform base, in M0) + 16 l, which IS the plain
// [k][rows] layout for a wave's 64 consecutive chunks, and a lane whose offset is out of the descriptor's range writes zeros
// (tools/census/lds_dma_probe.hip) -- so the producers hold no tile registers and issue no ds_write.  wv: this wave's index among
// the producers (uniform).  The caller counts the loads (lds_dma_count) for its own s_waitcnt vmcnt.  (A struct with a static
// member rather than a member template of Loader: called from a second kernel instantiation, that one failed overload
// resolution on the host pass with no reason given.)
template <class L, bool TAB>
struct LdsDma {
__device__ static __forceinline__ void run(float* stage, __amdgpu_buffer_rsrc_t rs, const typename L::P& p, const ConvG& cg, int g, int b, int k0, int kend,
                                    int x0, int xlim, int tid, int wv, const int2* ktab, const KCol* cols) {
    typedef typename L::ADDR_T ADDR;
    constexpr int NCH = L::NCH, CHUNKS = L::CHUNKS;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    if constexpr (L::KC) {
        // k-contiguous operand (the small one of a dense forward product: the weights): its stage image [k-quad][row] x 16 bytes is
        // lane-linear in the ROW, so a wave load takes one k-quad of 64 consecutive rows -- 64 separate 16-byte pieces of memory, a
        // pattern only a small, cache-resident operand should be given.  Lanes past the tile's last row are switched off (they
        // would land in the next k-quad's rows); rows past the matrix edge load zeros like every out-of-range lane.
        static_assert(!L::PERM_F && !L::TRANSPOSED && L::VEC_W == 4, "LDS-DMA: plain k-quad layout");
        constexpr int WPR = (L::ROWS_N + 63) / 64, NW = (L::BK_N / 4) * WPR;   // wave loads per k-quad / per tile
        const int lane = tid & 63;
#pragma unroll
        for (int it = 0; it < (NW + 3) / 4; ++it) {
            const int w = wv + 4 * it;
            if (w < NW) {   // wave-uniform
                const int kq = w / WPR, rb = (w % WPR) * 64, row = rb + lane;
                const int k = k0 + 4 * kq, x = x0 + row;
                int o;
                const bool ok = ADDR::addr(p, cg, g, b, k, x, o) & (k < kend) & (x < xlim);
                const unsigned off = ok ? (unsigned)o * 4u : OOB;
                if (L::ROWS_N % 64 == 0 || row < L::ROWS_N)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(stage + (kq * L::ROWSP + rb) * 4), 16, off, 0, 0, 0);
            }
        }
        return;
    }
    static_assert(L::KC || (!L::TRANSPOSED && L::VEC_W == 4 && CHUNKS % 64 == 0), "LDS-DMA: row-contiguous operands, whole waves of chunks");
    unsigned off[NCH];
    if constexpr (has_ktab<ADDR>::value && TAB) {
        const unsigned limY = (unsigned)ADDR::ktab_limY(cg), limX = (unsigned)ADDR::ktab_limX(cg);
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        int2 ent[NCH];
        i32x2 raw[NCH];
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            int dk, dx;
            bool live;
            L::pos(it, tid, dk, dx, live);
            // the table is read behind the compiler's back: it takes an ordinary LDS read after an LDS-DMA load for a possible
            // alias of the load's destination and drains every load in flight first (s_waitcnt vmcnt(0) before each read)
            typedef __attribute__((address_space(3))) const int2* lds_tab;
            const unsigned la = (unsigned)(size_t)(lds_tab)(ktab + k0 + dk);
            asm volatile("ds_read_b64 %0, %1" : "=v"(raw[it]) : "v"(la));
        }
        // every entry passes through a wait of its own (the repeats cost a cycle each): a use of the value cannot be scheduled
        // ahead of the wait it depends on
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[it]));
            ent[it] = make_int2(raw[it].x, raw[it].y);
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            int dk, dx;
            bool live;
            L::pos(it, tid, dk, dx, live);
            const int2 e = ent[it];
            const int dy = (int)(short)(e.y & 0xFFFF), dxx = e.y >> 16;
            const KCol& cl = cols[it];
            const bool ok = ((unsigned)(cl.cy + dy) < limY) & ((unsigned)(cl.cx + dxx) < limX) & (k0 + dk < kend);
            off[it] = ok ? (unsigned)(cl.base + e.x) * 4u : OOB;
        }
    } else {
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            int dk, dx;
            bool live;
            L::pos(it, tid, dk, dx, live);
            const int k = k0 + dk, x = x0 + dx;
            int o;
            const bool ok = ADDR::addr(p, cg, g, b, k, x, o) & (k < kend) & (x < xlim);
            off[it] = ok ? (unsigned)o * 4u : OOB;
        }
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it)
        if (CHUNKS % THREADS == 0 || wv * 64 + it * THREADS < CHUNKS)   // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(stage + (wv * 64 + it * THREADS) * 4), 16, off[it], 0, 0, 0);
}
};
// loads of wave wv per tile of loader L
template <class L> __device__ __forceinline__ int lds_dma_count(int wv) {
    if (L::KC) return ((L::BK_N / 4) * ((L::ROWS_N + 63) / 64) - wv + 3) / 4;
    return (L::CHUNKS - wv * 64 + THREADS - 1) / THREADS;
}

// LDS stages.  Three with the register-staged producers.  The dense products whose operands are both row-contiguous (the
// backward-data product of a dense / cccp layer: vector path, 32x32 tiles, one K chunk owner per tile) have LDS-DMA producers
// instead, which keep their tiles in flight in LDS rather than in registers and so want the stages of the tiles in flight on
// top of the two being read: MC_LDS_DMA stages (4: measured against 5 and 6, which cost residency and gain nothing).
// Measured on the NIN shapes (same box, default build against this one): cccp backward-data 4-7 % faster (89.8 -> 85.8,
// 51.1 -> 48.7, 30.8 -> 28.7 us).  The convolution forward (-DMC_LDS_DMA_CONV, its K table read by hand-written LDS reads) is
// unchanged at 5x5x192 and 9 % slower at 3x3x192 -- the fourth stage costs it residency, and what its producers cost the
// consumers is LDS bandwidth, which the DMA path uses just the same -- so it keeps the register-staged producers.
// -DMC_NO_LDS_DMA: register-staged producers everywhere.
#ifndef MC_LDS_DMA
#define MC_LDS_DMA 4
#endif
#ifdef MC_NO_LDS_DMA
template <class A, class B> constexpr int n_stages(int, int, int) { return 3; }
#else
template <class A, class B> constexpr int n_stages(int vec, int kq, int mf) {
#ifdef MC_LDS_DMA_CONV
    constexpr bool conv = true;
#else
    constexpr bool conv = !has_ktab<B>::value;
#endif
    // (a k-contiguous A operand only as the plain weight matrix of a dense forward product: see LdsDma)
#ifdef MC_NO_LDS_DMA_KC
    constexpr bool aok = !A::KC;
#else
    constexpr bool aok = !A::KC || std::is_same<A, MatKcAddr>::value;
#endif
    return (aok && !B::KC && conv && vec == 4 && kq == 1 && mf == 0) ? MC_LDS_DMA : 3;
}
#endif
// s_waitcnt vmcnt(n) for a run-time (uniform) n; waiting for fewer loads in flight than allowed is always correct
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define MC_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        MC_VM_CASE(0) MC_VM_CASE(1) MC_VM_CASE(2) MC_VM_CASE(3) MC_VM_CASE(4) MC_VM_CASE(5) MC_VM_CASE(6) MC_VM_CASE(7) MC_VM_CASE(8)
        MC_VM_CASE(9) MC_VM_CASE(10) MC_VM_CASE(11) MC_VM_CASE(12) MC_VM_CASE(13) MC_VM_CASE(14) MC_VM_CASE(15)
#undef MC_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
}

template <class AADDR, class BADDR, int MT, int NT, int WM, int WN, int BK, int VEC, int NS, int EPI, int KQ, int MF = 0>
__global__ void __launch_bounds__(2 * THREADS)
#ifndef MC_NO_T51_CAP   // the 160-row tile sits 3-17 registers above 128 = two resident workgroups per CU: cap it (a few spills outside the K loop)
// (not the weight-gradient kernels: their k-contiguous tiles leave LDS for one workgroup per CU, and at 128 registers they spill 300-500)
__attribute__((amdgpu_waves_per_eu((MT == 5 && NT == 1 && !(AADDR::KC && BADDR::KC)) ? 4 : 1)))
#endif
mc_igemm(typename AADDR::P pa, typename BADDR::P pb, ConvG cg, Dims d, Epi ep) {
    static_assert(WM * WN * KQ == 4 && (KQ == 1 || KQ == 2), "4 consumer waves");
    static_assert(KQ == 1 || EPI == EPI_PLAIN, "the K-split tile has the plain epilogue only");
    static_assert(NS >= 2, "at least two tile register sets");
    static_assert(MF == 0 || (KQ == 1 && EPI == EPI_PLAIN && AADDR::KC && BADDR::KC && BK % 8 == 0), "16x16x4 tiles: k-contiguous operands, plain epilogue");
    constexpr int UNIT = MF ? 16 : 32;
    constexpr int BM = UNIT * MT * WM, BN = UNIT * NT * WN;
    typedef LoaderT<AADDR, BM, BN, BK, VEC, MF == 1> LA;
    typedef LoaderT<BADDR, BN, BM, BK, VEC, MF == 1> LB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // three LDS stages, stage s = [A tile | B tile]: during K step t the consumers read stage t % 3 (and,
    // near its end, prefetch the first fragments of stage (t+1) % 3, complete since the previous barrier)
    // while the producers fill stage (t+2) % 3.
    constexpr int STAGE = LA::LDS_FLOATS + LB::LDS_FLOATS;
    constexpr int NSTG = n_stages<AADDR, BADDR>(VEC, KQ, MF);   // 3, or more with LDS-DMA producers (tile u then lives in stage u % NSTG)

    // Column tiles.  A workgroup is a "column worker": worker w = blockIdx.x of gridDim.x takes column tiles
    // w, w + gridDim.x, ... and runs their K loops back to back as ONE pipeline of nTiles * nk steps, so only the
    // first tile pays the pipeline fill (short-K products: K = 75 ... 192 is 5-12 steps per tile).  With
    // gridDim.x == d.colTiles every worker has one tile.  XCD awareness: with d.xcdSwizzle (column tiles and
    // workers both multiples of 8; workgroups are dealt round-robin over the 8 XCDs) logical tile t maps to
    // column block (t % 8) * (colTiles / 8) + t / 8, which keeps neighbouring column blocks (neighbouring
    // modules, overlapping patches) on one XCD's L2; a worker's tiles all share t % 8.
    auto col_block = [&](int t) { return d.xcdSwizzle ? (t & 7) * (d.colTiles >> 3) + (t >> 3) : t; };
    const int nTiles = (d.colTiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1
    // blockIdx.z = (k-chunk q, group g, batch b)
    const int zb = blockIdx.z / d.batches, b = blockIdx.z - zb * d.batches;
    const int q = zb / d.groups, g = zb - q * d.groups;
    const int coff = b * d.colB;                      // first B / C column of this batch
    const int i0 = blockIdx.y * BM;
    const int ilim = d.Mg, jlim = coff + d.Ng;
    // K chunks: contiguous ranges [q * kChunk, (q + 1) * kChunk), or -- interleaved -- chunk q takes K tiles q, q + chunks, ...
    // (d.kStep = chunks * BK), so that the workgroups running side by side read neighbouring 128-byte pieces of every operand row
    const bool kil = d.kStep != BK;
    const int kbeg = kil ? q * BK : q * d.kChunk;
    const int kend = kil ? d.Kg : min(kbeg + d.kChunk, d.Kg);
    const int nkTile = kil ? (d.Kg - kbeg + d.kStep - 1) / d.kStep : (kend - kbeg + BK - 1) / BK;   // K steps per tile, >= 1
    const int nk = nkTile * nTiles;                   // pipeline steps; barriers executed by every wave: 1 + nk

    if (d.stagger) {   // do two co-resident workgroups that start together hurt each other (prologues and epilogues in lockstep)?
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (2 * lin >= gridDim.x * gridDim.y * gridDim.z)
            for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
#ifdef MC_ABL_STAGGER   // timing-only ablation: do the workgroups of a CU hurt each other by running in lockstep?
    if (MC_ABL_STAGGER == 1 ? ((blockIdx.x >> 3) & 1) : (2 * blockIdx.x >= gridDim.x)) {
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
    }
#endif
    // k table of the B loader (d.ktabN entries behind the stages and the bias column): built by all 512 threads, visible to the
    // producers after one workgroup barrier that every wave passes before the roles split
    const int2* ktab = nullptr;
    if constexpr (has_ktab<BADDR>::value) {
        if (d.ktabN) {
            int2* tab = reinterpret_cast<int2*>(smem + NSTG * STAGE + BM);
            for (int k = threadIdx.x; k < d.ktabN; k += 2 * THREADS) tab[k] = BADDR::ktab_entry(cg, k < d.Kg ? k : 0);
            ktab = tab;
            lds_done_barrier();
        }
    }
    if (threadIdx.x >= THREADS) {
        // ------------------------------ producer waves ------------------------------
        // Register ring of NS tile sets: tile u lives in set u % NS and goes to LDS stage u % 3.  Tiles 0
        // and 1 are staged before the first barrier; during K step t tile t+2 (the oldest loads in
        // flight) is written to its stage and its register set is refilled with tile t+2+NS, so NS-1..NS
        // tiles are in flight per workgroup (under load an L2 / Infinity-Cache read takes 1-2 us).
        const int tid = threadIdx.x - THREADS;
#ifndef MC_NO_PRODUCER_PRIO
        // The consumers were dispatched first, so at equal priority they win every vector-issue arbitration
        // and a SIMD with two MFMA waves leaves the producers ~1 issue slot per 30 cycles (measured with
        // MC_STAMP).  Producers are short bursts between long waits: give them the slot when they want it.
#ifndef MC_PRODUCER_PRIO
#define MC_PRODUCER_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(MC_PRODUCER_PRIO);
#endif
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(AADDR::base(pa), pa.bytes), rb = make_rsrc(BADDR::base(pb), pb.bytes);
        // Register sets.  When both operands take the transposing loader and their units fit the 256 producer threads side by
        // side, a thread serves ONE operand -- waves [0, TA / 64) gather A units, the others B units -- and a set is the four
        // chunks of that one unit (serving both would double the ring: 146 registers, one workgroup per CU).  The role is a
        // compile-time property of the producer instance (ROLE 1 = A, 2 = B, 0 = both), chosen once, outside the K loop: as a
        // branch inside it the compiler lost its exact vmcnt bookkeeping and drained the loads in flight every third step.
        constexpr int TA = (LA::UNITS + 63) / 64 * 64;
        constexpr bool SPLIT = LA::TRU && LB::TRU && LA::NU == 1 && LB::NU == 1 && TA + LB::UNITS <= THREADS;
        auto produce = [&](auto persist_, auto tab_, auto role_) {
        constexpr bool PERSIST = decltype(persist_)::value;
        constexpr bool TAB = decltype(tab_)::value;   // the B loader reads the k part of its addresses from the LDS table
        constexpr int ROLE = decltype(role_)::value;
        struct SetBoth { float4 a[LA::NCH]; float4 b[LB::NCH]; };
        struct SetOne { float4 r[4]; };
        typename std::conditional<ROLE != 0, SetOne, SetBoth>::type as[NS];
        // Persistent workers (several column tiles per workgroup): fetches are issued in step order (u = 0, 1, 2, ...),
        // so the (tile, K position) of the next one is a running pair of workgroup-uniform counters rather than a
        // division of u.  One tile per workgroup (PERSIST false): the column origin is a loop invariant and the
        // loaders' per-column decode stays hoisted out of the K loop (conv2 backward-data is bound by this arithmetic).
        int fk = 0, fj0 = coff + col_block(blockIdx.x) * BN, ft = blockIdx.x;
        const int j0c = fj0;
        KCol bcols[LB::NCOL];   // K-table loaders: the column part of this thread's chunks of the current tile
        const int tidB = ROLE == 2 ? tid - TA : tid;
        auto prep = [&](int j0) {
            if (TAB && VEC == 4 && ROLE != 1) LB::prep_cols(bcols, pb, cg, g, j0, jlim, tidB);
        };
        prep(fj0);
        auto fetch_at = [&](auto& set, int k0, int j0) {
            if constexpr (ROLE == 1) {
                LA::fetch(set.r, ra, pa, cg, g, b, k0, kend, i0, ilim, tid);
            } else if constexpr (ROLE == 2) {
                LB::template fetch<TAB>(set.r, rb, pb, cg, g, b, k0, kend, j0, jlim, tidB, ktab, bcols);
            } else {
                LA::fetch(set.a, ra, pa, cg, g, b, k0, kend, i0, ilim, tid);
                LB::template fetch<TAB>(set.b, rb, pb, cg, g, b, k0, kend, j0, jlim, tid, ktab, bcols);
            }
        };
        auto fetch = [&](auto& set, int u) {
            if (PERSIST) {
                fetch_at(set, kbeg + fk * d.kStep, fj0);
                if (++fk == nkTile) {
                    fk = 0;
                    ft += gridDim.x;
                    fj0 = coff + col_block(ft) * BN;
                    prep(fj0);
                }
            } else {
                fetch_at(set, kbeg + u * d.kStep, j0c);
            }
        };
        // (16x16x4 tiles only: the 32x32 tiles keep their sums in the consumers, where a few adds ride on 64-cycle matrix instructions)
        constexpr bool PSA = has_sum<AADDR>::value && MF == 1, PSB = has_sum<BADDR>::value && MF == 1;
        float psa[PSA ? LA::NCH : 1], psb[PSB ? LB::NCH : 1];   // k-sum partials (launches that ask for sums have one tile per workgroup)
#pragma unroll
        for (int it = 0; it < (PSA ? LA::NCH : 1); ++it) psa[it] = 0.f;
#pragma unroll
        for (int it = 0; it < (PSB ? LB::NCH : 1); ++it) psb[it] = 0.f;
        auto commit = [&](const auto& set, int stage) {
            if constexpr (ROLE == 1) {
                LA::commit(set.r, smem + stage * STAGE, tid);
            } else if constexpr (ROLE == 2) {
                LB::commit(set.r, smem + stage * STAGE + LA::LDS_FLOATS, tidB);
            } else {
                LA::commit(set.a, smem + stage * STAGE, tid);
                LB::commit(set.b, smem + stage * STAGE + LA::LDS_FLOATS, tid);
                if constexpr (PSA) LA::add_sums(set.a, psa);
                if constexpr (PSB) LB::add_sums(set.b, psb);
            }
        };
#pragma unroll
        for (int u = 0; u < NS; ++u)
            if (u < nk) fetch(as[u], u);
        // per-row bias of this tile -> LDS (behind the three stages), so the epilogue does not wait on one global load
        // per output row; visible to the consumers through the first barrier
        if (tid < BM) smem[NSTG * STAGE + tid] = (ep.bias && i0 + tid < d.Mg) ? ep.bias[i0 + tid + g * ep.rowG] : 0.f;
        commit(as[0], 0);
        if (NS < nk) fetch(as[0], NS);
        if (1 < nk) {
            commit(as[1 % NS], 1);
            if (NS + 1 < nk) fetch(as[1 % NS], NS + 1);
        }
        lds_done_barrier();   // tiles 0 and 1 visible
        int t = 0, stage = 2; // stage of tile t+2
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, accC = 0, accF = 0, accL = 0, accB = 0;
        (void)p0; (void)p1; (void)p2; (void)p3; (void)p4; (void)accC; (void)accF; (void)accL; (void)accB;
        // steady state: NS K steps per trip, no conditionals inside (keeps the compiler's vmcnt counts exact)
        for (; t + 2 * NS + 2 <= nk; t += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int u = t + j + 2;                 // tile written during K step t+j; set (j+2) % NS
                STAMP(p0);
                commit(as[(j + 2) % NS], stage);
                STAMP(p1);
                fetch(as[(j + 2) % NS], u + NS);
                STAMP(p2);
                stage = stage == 2 ? 0 : stage + 1;
#ifdef MC_STAMP
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                STAMP(p3);
#endif
                lds_done_barrier();   // end of K step t+j
                STAMP(p4);
#ifdef MC_STAMP
                accC += p1 - p0; accF += p2 - p1; accL += p3 - p2; accB += p4 - p3;
#endif
            }
        }
#ifdef MC_STAMP
        if (tid == 0) {
            const int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            if (blk < 8192) { g_stamp[blk * 8 + 4] = accC; g_stamp[blk * 8 + 5] = accF; g_stamp[blk * 8 + 6] = accL; g_stamp[blk * 8 + 7] = accB; }
        }
#endif
        // tail: t is a multiple of NS, so the set indices stay static
        for (; t < nk; t += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int tt = t + j, u = tt + 2;
                if (tt < nk) {
                    if (u < nk) {
                        commit(as[(j + 2) % NS], stage);
                        if (u + NS < nk) fetch(as[(j + 2) % NS], u + NS);
                    }
                    stage = stage == 2 ? 0 : stage + 1;
                    lds_done_barrier();   // end of K step tt
                }
            }
        }
        if constexpr (PSA) {
            if (ep.sumOut && blockIdx.x == 0) LA::sum_rows(psa, ep.sumOut + q * ep.sumStride + g * ep.rowG, ep.sumAlpha, i0, ilim, tid);
        }
        if constexpr (PSB) {
            if (ep.sumOut && blockIdx.y == 0) LB::sum_rows(psb, ep.sumOut + q * ep.sumStride + g * ep.colG, ep.sumAlpha, j0c, jlim, tid);
        }
        };
        auto produce_roles = [&](auto persist_, auto tab_) {
            if constexpr (SPLIT) {
                if (__builtin_amdgcn_readfirstlane(tid >> 6) < TA / 64) produce(persist_, tab_, std::integral_constant<int, 1>{});
                else produce(persist_, tab_, std::integral_constant<int, 2>{});
            } else {
                produce(persist_, tab_, std::integral_constant<int, 0>{});
            }
        };
        // LDS-DMA producers (NSTG > 3): tile u is loaded straight into stage u % NSTG.  NSTG - 1 tiles are issued before the first
        // barrier; during K step t tile t + NSTG - 1 is issued into the stage the consumers left at the previous barrier, and the
        // barrier that ends step t waits until tile t + 2 has landed (the consumers prefetch its first fragments during step
        // t + 1): a wave's loads complete in order, so that is "at most (tiles issued after t + 2) x (loads per tile of this
        // wave) still in flight".  No tile registers, no ds_write, no compiler bookkeeping to keep exact.
        auto produce_dma = [&](auto persist_, auto tab_) {
            constexpr bool PERSIST = decltype(persist_)::value;
            constexpr bool TAB = decltype(tab_)::value;
            if constexpr (NSTG > 3) {
                int fk = 0, fj0 = coff + col_block(blockIdx.x) * BN, ft = blockIdx.x, istage = 0;
                KCol bcols[LB::NCOL];
                if (TAB) LB::prep_cols(bcols, pb, cg, g, fj0, jlim, tid);
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                const int lw = lds_dma_count<LA>(wv) + lds_dma_count<LB>(wv);
                auto issue = [&](auto) {   // (generic: the body is checked only where it is instantiated, on the device side)
                    float* st = smem + istage * STAGE;
                    const int k0 = kbeg + fk * d.kStep;
                    LdsDma<LA, false>::run(st, ra, pa, cg, g, b, k0, kend, i0, ilim, tid, wv, nullptr, nullptr);
                    LdsDma<LB, TAB>::run(st + LA::LDS_FLOATS, rb, pb, cg, g, b, k0, kend, fj0, jlim, tid, wv, ktab, bcols);
                    istage = istage == NSTG - 1 ? 0 : istage + 1;
                    ++fk;
                    if (PERSIST && fk == nkTile) {
                        fk = 0;
                        ft += gridDim.x;
                        fj0 = coff + col_block(ft) * BN;
                        if (TAB) LB::prep_cols(bcols, pb, cg, g, fj0, jlim, tid);
                    }
                };
                const int pre = nk < NSTG - 1 ? nk : NSTG - 1;
                // (the bias column first: an ordinary LDS store after an LDS-DMA load makes the compiler drain the loads in flight)
                if (tid < BM) smem[NSTG * STAGE + tid] = (ep.bias && i0 + tid < d.Mg) ? ep.bias[i0 + tid + g * ep.rowG] : 0.f;
                for (int u = 0; u < pre; ++u) issue(0);
                wait_vm((pre > 2 ? pre - 2 : 0) * lw);
                lds_done_barrier();   // tiles 0 and 1 visible
                for (int t = 0; t < nk; ++t) {
                    if (t + NSTG - 1 < nk) issue(0);
                    const int last = t + NSTG - 1 < nk ? t + NSTG - 1 : nk - 1;   // newest tile issued
                    wait_vm((last > t + 2 ? last - (t + 2) : 0) * lw);
                    lds_done_barrier();   // end of K step t
                }
            }
        };
        const bool onePerTile = (int)gridDim.x == d.colTiles;
        if constexpr (NSTG > 3) {
            if constexpr (has_ktab<BADDR>::value) {
                if (ktab) {
                    if (onePerTile) produce_dma(std::false_type{}, std::true_type{});
                    else produce_dma(std::true_type{}, std::true_type{});
                    return;
                }
            }
            if (onePerTile) produce_dma(std::false_type{}, std::false_type{});
            else produce_dma(std::true_type{}, std::false_type{});
            return;
        } else {
        if constexpr (has_ktab<BADDR>::value) {
            if (ktab) {
                if (onePerTile) produce_roles(std::false_type{}, std::true_type{});
                else produce_roles(std::true_type{}, std::true_type{});
                return;
            }
        }
        if (onePerTile) produce_roles(std::false_type{}, std::false_type{});
        else produce_roles(std::true_type{}, std::false_type{});
        return;
        }
    }

    if constexpr (MF == 1) {
        // ------------------------------ consumer waves, 16x16x4 blocks ------------------------------
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int lq = lane >> 4, l15 = lane & 15;   // lq: lane group (q is this workgroup's K chunk)
        const int wm = wave / WN, wn = wave % WN;
        f32x4 acc[MT][NT];
        // A K step is BK / 8 groups of 8 k's = two 16-byte chunks per row.  MFMA m (0, 1) of a group takes from lane group lq
        // k = 4 (lq & 1) + 2 m + (lq >> 1) -- over the four lane groups k = 0,4,1,5 then 2,6,3,7, the order of the 32x32x2 path --
        // and the permuted chunk layout (Loader, PERM) puts a lane's two values side by side: ONE ds_read_b64 per 16-row block and
        // group, no other vector instruction in the loop.  That matters: a wave's vector and LDS instructions do NOT run in the
        // shadow of its own matrix instructions (tools/census/mfma16_probe.hip), and there is one consumer wave per SIMD.
        // Two fragment sets: group g + 1 (or group 0 of the next stage, complete since the previous barrier) is requested before
        // the MFMAs of group g.
        constexpr int GROUPS = BK / 8;
        static_assert(GROUPS % 2 == 0, "fragment sets alternate by group parity across K steps");
        const int half = (lq >> 1) ^ (lq & 1), par = lq & 1;
        float2 fa[2][MT], fb[2][NT];
        auto load_group = [&](const float* cA, const float* cB, int g_, int set) {
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[set][a] = reinterpret_cast<const float2*>(cA)[((2 * g_ + par) * LA::ROWSP + (wm * MT + a) * 16 + l15) * 2 + half];
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[set][b] = reinterpret_cast<const float2*>(cB)[((2 * g_ + par) * LB::ROWSP + (wn * NT + b) * 16 + l15) * 2 + half];
        };
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ep.C + q * ep.chunkStride, 0, ep.cBytes, 0x00020000);
        lds_done_barrier();   // tiles 0 and 1 visible
        load_group(smem, smem + LA::LDS_FLOATS, 0, 0);
        int stage = 0;
        unsigned long long c0 = 0, c1 = 0, c2 = 0, cStart = 0, accM = 0, accW = 0, s0 = 0, s1 = 0, accS = 0;
        (void)c0; (void)c1; (void)c2; (void)cStart; (void)accM; (void)accW; (void)s0; (void)s1; (void)accS;
        STAMP(cStart);
#ifdef MC_STAMP
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int ct = blockIdx.x; ct < d.colTiles; ct += gridDim.x) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < nkTile; ++t) {
                const int nstage = stage == NSTG - 1 ? 0 : stage + 1;
                const float* cA = smem + stage * STAGE;
                const float* cB = cA + LA::LDS_FLOATS;
                const float* nA = smem + nstage * STAGE;   // on the last K step this stage holds stale data: read, never used
                const float* nB = nA + LA::LDS_FLOATS;
                STAMP(c0);
#pragma unroll
                for (int g_ = 0; g_ < GROUPS; ++g_) {
                    const int set = g_ & 1;
                    if (g_ + 1 < GROUPS) load_group(cA, cB, g_ + 1, set ^ 1);
                    else load_group(nA, nB, 0, set ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][a].x, fb[set][b].x, acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][a].y, fb[set][b].y, acc[a][b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                STAMP(c1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                STAMP(c2);
#ifdef MC_STAMP
                accM += c1 - c0; accW += c2 - c1;
#endif
                stage = nstage;
            }
            STAMP(s0);
            const int j0 = coff + col_block(ct) * BN;
            // store: C/D map of the 16x16 block: col = lane & 15, row = 4 * (lane >> 4) + r.  Branch-free through the chunk's descriptor.
            const unsigned ldcB = (unsigned)ep.ldc * 4u;
            const bool relu = ep.act == MC_NEURON_RELU;
            // the row / column terms below do not change from tile to tile; opaque copies of the lane coordinates keep the compiler
            // from hoisting a register per (block, row) of them out of the tile loop (spills at 256 registers otherwise)
            int lqq = lq, lcc = l15;
            asm volatile("" : "+v"(lqq), "+v"(lcc));
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                __builtin_amdgcn_sched_barrier(0);   // a block row at a time
                const int rl0 = (wm * MT + a) * 16 + 4 * lqq;
                const int iB = i0 + rl0;
                const float4 bq = *reinterpret_cast<const float4*>(&smem[NSTG * STAGE + rl0]);
                const float bia[4] = {bq.x, bq.y, bq.z, bq.w};
                const unsigned rowB = (unsigned)(iB + g * ep.rowG) * ldcB;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int j = j0 + (wn * NT + b) * 16 + lcc;
                    const bool jok = j < jlim;
                    const unsigned colB = (unsigned)(j + g * ep.colG) * 4u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned off = (jok & (iB + r < ilim)) ? rowB + (unsigned)r * ldcB + colB : OOB;
                        float v = ep.alpha * acc[a][b][r] + bia[r];
                        v = (relu & (v < 0.f)) ? 0.f : v;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, off, 0, 0);
                    }
                }
            }
            STAMP(s1);
#ifdef MC_STAMP
            accS += s1 - s0;
#endif
        }
#ifdef MC_STAMP
        if (threadIdx.x == 0) {
            const int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
            if (blk < 8192) { g_stamp[blk * 8 + 0] = c2 - cStart; g_stamp[blk * 8 + 1] = accM; g_stamp[blk * 8 + 2] = ((rt1 - rt0) << 32) | (accW & 0xFFFFFFFFull); g_stamp[blk * 8 + 3] = (unsigned long long)nk | (accS << 32); }
        }
#endif
        return;
    }
    // ------------------------------ consumer waves ------------------------------
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int kq = wave / (WM * WN), w2 = wave - kq * (WM * WN);
    const int wm = w2 / WN, wn = w2 % WN;

    f32x16 acc[MT][NT];

    constexpr bool SUMA = has_sum<AADDR>::value && KQ == 1 && MF == 0, SUMB = has_sum<BADDR>::value && KQ == 1 && MF == 0;
    float rs[MT], cs[NT];   // SUMA / SUMB: this lane's share of the k-sums of its A rows / B columns
    (void)rs; (void)cs;

    // LDS -> register fragments, software-pipelined ACROSS K steps: "slab" s holds the operands of one
    // MFMA k-step (2 k values: lane half h supplies k = kof(s) + h-dependent offset).  The reads of slab
    // s+PF are issued before the MFMAs of slab s; the last PF slabs of a K step prefetch the first PF
    // slabs of the NEXT stage, so the matrix pipe does not drain at the barrier.
#ifndef MC_PF
#define MC_PF 2   // fragment prefetch distance in slabs
#endif
    constexpr int SLABS = BK / 2 / KQ, PF = MC_PF;   // slabs of a K step this wave multiplies (its half when KQ == 2)
    static_assert(SLABS >= 4 + PF && SLABS % 4 == 0, "prefetch into consumed fragment registers");
    float af[SLABS][MT], bf[SLABS][NT];
#ifdef MC_ABL_NOLDSREAD   // timing-only ablation: the consumers read no fragments (what do their LDS reads cost the MFMA stream?)
    float ablv = 1.f;
    asm volatile("" : "+v"(ablv));
#endif
    auto load_slab = [&](const float* cA, const float* cB, int s_) {
        // slab s_ (of this wave's part of the tile) = (k8, e): k = k8*8 + 4*h + e.  NAT operands read one float per slab; KC
        // operands fetch the four slabs of a k8 group with one ds_read_b128 when e == 0.
#ifdef MC_ABL_NOLDSREAD
        for (int a = 0; a < MT; ++a) af[s_][a] = ablv;
        for (int b = 0; b < NT; ++b) bf[s_][b] = ablv;
        return;
#endif
        const int k8 = (s_ >> 2) + kq * (SLABS / 4), e = s_ & 3;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int row = (wm * MT + a) * 32 + l31;
            if (LA::LKC) {
                // float4 index of (k-quad, row): k-contiguous loaders [quad][ROWS + 1], transposing loaders quarters of pitch QP
                const int qi = LA::KC ? (k8 * 2 + h) * LA::ROWSP + row
                             : LA::TRP ? (k8 * 2 + h) * 2 * LA::HP + (l31 & 1) * LA::HP + (wm * MT + a) * 16 + (l31 >> 1)
                                       : (k8 * 2 + h) * 4 * LA::QP + (l31 & 3) * LA::QP + (wm * MT + a) * 8 + (l31 >> 2);
                if (MT >= 5 || KQ == 2) {   // tall / 3x3 wave tiles: two b64 reads per k8 group halve the fragment registers in flight
                    if ((e & 1) == 0) {
                        const float2 v = reinterpret_cast<const float2*>(cA)[qi * 2 + (e >> 1)];
                        af[s_][a] = v.x; af[s_ + 1][a] = v.y;
                    }
                } else if (e == 0) {
                    const float4 v = reinterpret_cast<const float4*>(cA)[qi];
                    af[s_][a] = v.x; af[s_ + 1][a] = v.y; af[s_ + 2][a] = v.z; af[s_ + 3][a] = v.w;
                }
            } else {
#ifdef MC_NO_READ2   // A/B switch: one ds_read_b32 per slab
                af[s_][a] = cA[(k8 * 8 + 4 * h + e) * BM + row];
#else
                // slabs e and e + 1 of a lane half are consecutive k: BM floats apart, one ds_read2_b32 for the pair
                if ((e & 1) == 0) {
                    af[s_][a] = cA[(k8 * 8 + 4 * h + e) * BM + row];
                    af[s_ + 1][a] = cA[(k8 * 8 + 4 * h + e + 1) * BM + row];
                }
#endif
            }
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = (wn * NT + b) * 32 + l31;
            if (LB::LKC) {
                const int qi = LB::KC ? (k8 * 2 + h) * LB::ROWSP + col
                             : LB::TRP ? (k8 * 2 + h) * 2 * LB::HP + (l31 & 1) * LB::HP + (wn * NT + b) * 16 + (l31 >> 1)
                                       : (k8 * 2 + h) * 4 * LB::QP + (l31 & 3) * LB::QP + (wn * NT + b) * 8 + (l31 >> 2);
                if (KQ == 2) {
                    if ((e & 1) == 0) {
                        const float2 v = reinterpret_cast<const float2*>(cB)[qi * 2 + (e >> 1)];
                        bf[s_][b] = v.x; bf[s_ + 1][b] = v.y;
                    }
                } else if (e == 0) {
                    const float4 v = reinterpret_cast<const float4*>(cB)[qi];
                    bf[s_][b] = v.x; bf[s_ + 1][b] = v.y; bf[s_ + 2][b] = v.z; bf[s_ + 3][b] = v.w;
                }
            } else {
#ifdef MC_NO_READ2
                bf[s_][b] = cB[(k8 * 8 + 4 * h + e) * BN + col];
#else
                if ((e & 1) == 0) {
                    bf[s_][b] = cB[(k8 * 8 + 4 * h + e) * BN + col];
                    bf[s_ + 1][b] = cB[(k8 * 8 + 4 * h + e + 1) * BN + col];
                }
#endif
            }
        }
    };

    // Tile store: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Branch-free like the loads: every element goes through a buffer descriptor over this chunk's output
    // (ep.cBytes) and an element outside the product is stored at offset 0xFFFFFFFF, which the range check drops.
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ep.C + q * ep.chunkStride, 0, ep.cBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.mask ? ep.mask : ep.C), 0, ep.mask ? ep.mBytes : 0u, 0x00020000);
    // EPI_PLAIN / EPI_BITS: the store is BATCHED -- per 32x32 block the byte offsets of 8 rows are formed, then 8 stores go
    // out back to back; the bias comes from LDS as four b128 reads per block (a tile without bias reads zeros), relu is a
    // select.  (One element at a time -- LDS bias read, wait, mask load, wait, store -- cost 200-500 cycles per store
    // instruction: MC_STAMP showed 30-50 % of a consumer's cycles in the store of short-K products.)
    // Bit masks: the relu-gradient mask of a block is ONE register per lane, the 16-bit half word of the lane's own 16
    // accumulators, loaded at the top of the tile so that its latency hides under the K loop.  A relu epilogue writes
    // the same layout: one 16-bit store per lane and block.
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(EPI == EPI_BITS ? ep.bitsIn : ep.bitsOut), 0,
                                                                            (EPI == EPI_BITS || EPI == EPI_PLAIN_BITSOUT) ? ep.wBytes : 0u, 0x00020000);
    unsigned mw[MT][NT];   // EPI_BITS: this lane's mask bits of the tile
    (void)mw;
    auto bits_off = [&](int a, int b, int j0) -> unsigned {   // byte offset of this lane's half word of block (a, b)
        const int ib = i0 + (wm * MT + a) * 32, j = j0 + (wn * NT + b) * 32 + l31;
        return ((ib < ilim) & (j < jlim)) ? (unsigned)((ib >> 5) * ep.ldc + j) * 4u + 2u * (unsigned)h : OOB;
    };
    auto load_mask_words = [&](int j0) {
        if constexpr (EPI == EPI_BITS) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) mw[a][b] = __builtin_amdgcn_raw_buffer_load_b16(rw, bits_off(a, b, j0), 0, 0);
        }
    };
    auto store_tile = [&](int j0) {
#ifdef MC_ABL_NOSTORE   // timing-only ablation: keep the accumulators alive, store nothing in range
        if (j0 >= 0) j0 = 0x3FFFFFFF;
#endif
        // the row terms below do not change from tile to tile; an opaque copy of h keeps the compiler from
        // hoisting ~64 registers per 32-row block of them out of the tile loop (measured: spills in the 160-row tiles)
        int hh = h;
        asm volatile("" : "+v"(hh));
        if constexpr (EPI == EPI_SLOW) {
    #pragma unroll
            for (int a = 0; a < MT; ++a) {
    #pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int j = j0 + (wn * NT + b) * 32 + l31;
                    const bool jok = j < jlim;
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (wm * MT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const int i = i0 + rl;
                        const bool ok = jok & (i < ilim);
                        const int row = i + g * ep.rowG;
                        const unsigned off = ok ? (unsigned)(row * ep.ldc + j + g * ep.colG) * 4u : OOB;
                        float v = ep.alpha * acc[a][b][r];
                        if (ep.beta != 0.f) v += ep.beta * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
                        if (ep.bias) v += smem[NSTG * STAGE + rl];
                        if (ep.act == MC_NEURON_RELU) v = v < 0.f ? 0.f : v;
                        if (ep.mask) {
                            const unsigned moff = ok ? (unsigned)(row * ep.ldmask + j + g * ep.colG) * 4u : OOB;
                            if (!(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, moff, 0, 0)) > 0.f)) v = 0.f;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, off, 0, 0);
                    }
                }
            }
        } else {
            const unsigned ldcB = (unsigned)ep.ldc * 4u;
            const bool relu = ep.act == MC_NEURON_RELU;
            constexpr bool WB = EPI == EPI_PLAIN_BITSOUT;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                // this lane's 16 rows of block a: rl0 + (r & 3) + 8 * (r >> 2), taken four at a time (one b128 of bias each;
                // the next group's bias is requested before this group's stores)
                const int rl0 = (wm * MT + a) * 32 + 4 * hh;
                const int iB = i0 + rl0;
                const unsigned rowB = (unsigned)(iB + g * ep.rowG) * ldcB;
                unsigned pb[NT];   // (output > 0) of this lane's 16 elements of block (a, b)
#pragma unroll
                for (int b = 0; b < NT; ++b) pb[b] = 0;
                float4 bq = *reinterpret_cast<const float4*>(&smem[NSTG * STAGE + rl0]);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    __builtin_amdgcn_sched_barrier(0);   // one group at a time: hoisted, the loads and offsets of a whole tile are 40+ live registers
                    float4 bn = bq;
                    if (q4 < 3) bn = *reinterpret_cast<const float4*>(&smem[NSTG * STAGE + rl0 + 8 * (q4 + 1)]);
                    const float bia[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        const int j = j0 + (wn * NT + b) * 32 + l31;
                        const bool jok = j < jlim;
                        const unsigned colB = (unsigned)(j + g * ep.colG) * 4u;
                        unsigned off[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int dr = 8 * q4 + e;
                            off[e] = (jok & (iB + dr < ilim)) ? rowB + (unsigned)dr * ldcB + colB : OOB;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * q4 + e;
                            float v = ep.alpha * acc[a][b][r] + bia[e];
                            v = (relu & (v < 0.f)) ? 0.f : v;
                            if constexpr (EPI == EPI_BITS) v = ((mw[a][b] >> r) & 1u) ? v : 0.f;
                            if constexpr (WB) pb[b] |= (v > 0.f) ? (1u << r) : 0u;   // rows / columns outside the product: bits nobody reads
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, off[e], 0, 0);
                        }
                    }
                    bq = bn;
                }
                if constexpr (WB) {
#pragma unroll
                    for (int b = 0; b < NT; ++b) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)pb[b], rw, bits_off(a, b, j0), 0, 0);
                }
            }
        }
    };

    lds_done_barrier();   // tiles 0 and 1 visible
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) load_slab(smem, smem + LA::LDS_FLOATS, s_);
    int stage = 0;
    unsigned long long c0 = 0, c1 = 0, c2 = 0, cStart = 0, accM = 0, accW = 0, s0 = 0, s1 = 0, accS = 0;
    (void)c0; (void)c1; (void)c2; (void)cStart; (void)accM; (void)accW; (void)s0; (void)s1; (void)accS;
    STAMP(cStart);
#ifdef MC_STAMP
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // one trip per column tile of this worker; the accumulators are (re)defined at the top of the trip, so no
    // accumulator value is live around the outer loop
    for (int ct = blockIdx.x; ct < d.colTiles; ct += gridDim.x) {
    load_mask_words(coff + col_block(ct) * BN);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    if constexpr (SUMA) {
#pragma unroll
        for (int a = 0; a < MT; ++a) rs[a] = 0.f;
    }
    if constexpr (SUMB) {
#pragma unroll
        for (int b = 0; b < NT; ++b) cs[b] = 0.f;
    }
    for (int t = 0; t < nkTile; ++t) {
        const int nstage = stage == NSTG - 1 ? 0 : stage + 1;
        STAMP(c0);
        const float* cA = smem + stage * STAGE;
        const float* cB = cA + LA::LDS_FLOATS;
        const float* nA = smem + nstage * STAGE;   // on the last K step this stage holds stale data: read, never used
        const float* nB = nA + LA::LDS_FLOATS;
#pragma unroll
        for (int s_ = 0; s_ < SLABS; ++s_) {
            if (s_ + PF < SLABS) load_slab(cA, cB, s_ + PF);
            else load_slab(nA, nB, s_ + PF - SLABS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s_][a], bf[s_][b], acc[a][b], 0, 0, 0);
            if constexpr (SUMA) {   // a handful of adds under MT * NT matrix instructions of 64 cycles each
#pragma unroll
                for (int a = 0; a < MT; ++a) rs[a] += af[s_][a];
            }
            if constexpr (SUMB) {
#pragma unroll
                for (int b = 0; b < NT; ++b) cs[b] += bf[s_][b];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // end of K step t.  No LDS wait here: every read of this stage has been consumed by an MFMA above,
        // and the prefetched stage is not rewritten before the barrier after next.
        STAMP(c1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(c2);
#ifdef MC_STAMP
        accM += c1 - c0; accW += c2 - c1;
#endif
        stage = nstage;
    }
    if constexpr (KQ == 2) {
        // One tile per workgroup (launch_cfg), so after the last K step's barrier the producers are gone and the three stages
        // are free: the kq = 1 waves park their accumulators there, lane-major (conflict-free), the kq = 0 waves add them.
        // (A barrier counts only the waves that have not ended.)
        float* red = smem + (size_t)w2 * (MT * NT * 16 * 64);
        static_assert((size_t)WM * WN * MT * NT * 16 * 64 <= (size_t)NSTG * STAGE, "accumulator hand-over must fit the stages");
        if (kq == 1) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((a * NT + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        lds_done_barrier();
        if (kq == 1) return;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                __builtin_amdgcn_sched_barrier(0);   // a block at a time: hoisted together the 144 reads need 144 more registers
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * NT + b) * 16 + r) * 64 + lane];
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // k-sums: the two lane halves hold alternate groups of four k values; waves that share the operand hold copies
    if constexpr (SUMA) {
        if (ep.sumOut) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float v = rs[a] + __shfl_xor(rs[a], 32);
                const int i = i0 + (wm * MT + a) * 32 + l31;
                if (ct == 0 && wn == 0 && h == 0 && i < ilim) ep.sumOut[q * ep.sumStride + i + g * ep.rowG] = ep.sumAlpha * v;
            }
        }
    }
    if constexpr (SUMB) {
        if (ep.sumOut) {
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const float v = cs[b] + __shfl_xor(cs[b], 32);
                const int j = coff + col_block(ct) * BN + (wn * NT + b) * 32 + l31;
                if (blockIdx.y == 0 && wm == 0 && h == 0 && j < jlim) ep.sumOut[q * ep.sumStride + j + g * ep.colG] = ep.sumAlpha * v;
            }
        }
    }
    // this tile's K loop is complete: store it while the producers run ahead into the next tile
    STAMP(s0);
    store_tile(coff + col_block(ct) * BN);
    STAMP(s1);
#ifdef MC_STAMP
    accS += s1 - s0;
#endif
    }
#ifdef MC_STAMP
    if (tid == 0) {
        const int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        if (blk < 8192) { g_stamp[blk * 8 + 0] = c2 - cStart; g_stamp[blk * 8 + 1] = accM; g_stamp[blk * 8 + 2] = ((rt1 - rt0) << 32) | (accW & 0xFFFFFFFFull); g_stamp[blk * 8 + 3] = (unsigned long long)nk | (accS << 32); }
    }
#endif

}

// ---------------------------------------------------------------------------------
// Host-side launch helpers
// ---------------------------------------------------------------------------------
template <class AADDR, class BADDR, int MT, int NT, int WM, int WN, int BK, int VEC, int EPI, int NS = MC_NS, int KQ = 1, int MF = 0>
int launch_cfg(const typename AADDR::P& pa, const typename BADDR::P& pb, const ConvG& cg, const Dims& d0,
               const Epi& ep, int chunks, hipStream_t st) {
    constexpr int BM = (MF ? 16 : 32) * MT * WM, BN = (MF ? 16 : 32) * NT * WN;
    typedef LoaderT<AADDR, BM, BN, BK, VEC> LA;
    typedef LoaderT<BADDR, BN, BM, BK, VEC> LB;
    constexpr size_t ldsTile = (n_stages<AADDR, BADDR>(VEC, KQ, MF) * (LA::LDS_FLOATS + LB::LDS_FLOATS) + BM) * sizeof(float);   // the stages + the tile's bias column
    static_assert(ldsTile <= 160 * 1024, "LDS budget");
    auto kern = mc_igemm<AADDR, BADDR, MT, NT, WM, WN, BK, VEC, NS, EPI, KQ, MF>;
    Dims d = d0;
    // the B loader's k table (8 bytes per k, padded to whole K tiles + one tile of slack for the k tail of a chunk)
    static const bool noktab = getenv("MC_IGEMM_NOKTAB") != nullptr;
    d.ktabN = 0;
    if (has_ktab<BADDR>::value && !noktab && d.Kg <= KTAB_MAX && d.batches == 1) {
        const int n = (ceil_div(d.Kg, BK) + 1) * BK;
        if (ldsTile + (size_t)n * 8 <= 160 * 1024) d.ktabN = n;
    }
    static const bool kInterleave = getenv("MC_IGEMM_KINTERLEAVE") != nullptr;   // experiment: interleaved K chunks for the k-contiguous products
    d.kStep = (kInterleave && AADDR::KC && BADDR::KC && chunks > 1 && d.kChunk % BK == 0 && d.Kg % BK == 0) ? chunks * BK : BK;
    static const char* stg = getenv("MC_IGEMM_STAGGER");
    d.stagger = stg ? atoi(stg) : 0;
    const size_t lds = ldsTile + (size_t)d.ktabN * 8;
    constexpr size_t ldsMax = has_ktab<BADDR>::value ? (ldsTile + (size_t)(KTAB_MAX + 2 * BK) * 8 <= 160 * 1024 ? ldsTile + (size_t)(KTAB_MAX + 2 * BK) * 8 : 160 * 1024) : ldsTile;
    // per-device state of this instantiation (a function attribute belongs to the device's own code object)
    static std::atomic<unsigned long long> attrDone{0};
    static std::atomic<int> perCU[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (ldsMax > 48 * 1024 && !((attrDone.load(std::memory_order_acquire) >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax);
        attrDone.fetch_or(1ull << dev, std::memory_order_release);
    }
    dim3 grid(ceil_div(d.Ng, BN), ceil_div(d.Mg, BM), chunks * d.groups * d.batches);
    d.colTiles = (int)grid.x;
    // Short K loops (a handful of steps per tile): fewer workgroups than tiles, as many as are resident at once, each
    // running its tiles as one pipeline (see the kernel) instead of paying the pipeline fill per tile.
    const int stepsPerTile = ceil_div(d.kChunk < d.Kg ? d.kChunk : d.Kg, BK);
    static const bool nopersist = getenv("MC_IGEMM_NOPERSIST") != nullptr;
    // (not the 16x16x4 tiles when k-sums are asked for: their producers form them per workgroup, which then has exactly one tile)
    if (KQ == 1 && !nopersist && !(MF && ep.sumOut) && stepsPerTile <= 24 && (long)grid.x * grid.y * grid.z > 2L * num_cus()) {
        // resident workgroups per CU at this LDS footprint (it varies with the k table): cached as (lds << 8 | count)
        const int cached = perCU[dev].load(std::memory_order_relaxed);
        int pc = (cached >> 8) == (int)(lds >> 8) ? (cached & 0xFF) : 0;
        if (pc == 0) {
            int nb = 0;
            pc = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 2 * THREADS, lds) == hipSuccess && nb > 0) ? (nb > 255 ? 255 : nb) : 1;
            perCU[dev].store((int)((lds >> 8) << 8) | pc, std::memory_order_relaxed);
        }
        long workers = (long)pc * num_cus() / ((long)grid.y * grid.z);
        if (workers >= 8) workers = workers / 8 * 8;
        if (workers >= 1 && workers < (long)grid.x) grid.x = (unsigned)workers;
    }
    static const bool noswz = getenv("MC_IGEMM_NOSWZ") != nullptr;
    d.xcdSwizzle = (d.colTiles % 8 == 0 && grid.x % 8 == 0 && d.colTiles >= 16 && !noswz) ? 1 : 0;
    if (g_describeBuf) {   // dry run (mc_describe_*): which instantiation and grid this shape gets right now
        snprintf(g_describeBuf, (size_t)g_describeLen, "mc_igemm<%s, %s, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d> grid=(%u,%u,%u) tile %dx%d", AADDR::tag(),
                 BADDR::tag(), MT, NT, WM, WN, BK, VEC, NS, EPI, KQ, MF, grid.x, grid.y, grid.z, BM, BN);
        return MC_OK;
    }
    hipLaunchKernelGGL(kern, grid, dim3(2 * THREADS), lds, st, pa, pb, cg, d, ep);
    return launch_status("mc_igemm");
}

// Tile choice.  The MFMA pipe is the bound (64 cycles per 32x32x2 per SIMD), so what matters is
// (a) not padding the row dimension much, (b) enough workgroups to cover 256 CUs.
// ---------------------------------------------------------------------------------
// Tile choice.  Eight workgroup tiles are instantiated; a static rule picks one from the extents, and --
// because the best tile depends on things the rule cannot see (how many workgroups fit a CU at that
// register / LDS footprint, how the K chunks tail off) -- the first idempotent call of every distinct shape
// times all of them on the caller's stream and remembers the winner (MIOpen-"find" style; off with
// MC_IGEMM_AUTOTUNE=0, report with =2).  Every tile adds the k terms of an output element in the same
// order, so the choice never changes a result bit; the split-K chunk count, which would, is not tuned.
// ---------------------------------------------------------------------------------
// M56 .. M33: the 16x16x4 tiles (2 x 2 waves of MT x NT 16-blocks; both operands k-contiguous, plain epilogue only)
enum { T11, T21, T31, T51, T23, T32, T14, T24, M56, M35, M66, M36, M33, NTILES, T33K = NTILES };   // NTILES: the tiles the autotuner may pick among
static const char* const kTileName[NTILES + 1] = {"32x128", "64x128", "96x128", "160x128", "128x192", "192x128", "128x32", "128x64",
                                                  "160x192m", "96x160m", "192x192m", "96x192m", "96x96m", "96x192k2"};
inline bool is_mf_tile(int t) { return t >= M56 && t <= M33; }

// LDS footprint of a tile (three stages + the bias column): a k-contiguous build with a longer K tile (MC_BK_KC=64) keeps
// only the tiles that still fit 160 KB; launch_tile maps the others to the nearest one that does.
template <class AADDR, class BADDR, int MT, int NT, int WM, int WN, int BK, int VEC, int UNIT = 32>
constexpr bool tile_fits() {
    return (n_stages<AADDR, BADDR>(VEC, 1, UNIT == 16) * (size_t)(LoaderT<AADDR, UNIT * MT * WM, UNIT * NT * WN, BK, VEC>::LDS_FLOATS + LoaderT<BADDR, UNIT * NT * WN, UNIT * MT * WM, BK, VEC>::LDS_FLOATS) + UNIT * MT * WM) * sizeof(float) <= 160 * 1024;
}
// the 16x16x4 tiles exist for these products only
template <class AADDR, class BADDR, int BK, int EPI>
constexpr bool mf_ok() { return AADDR::KC && BADDR::KC && BK % 8 == 0 && EPI == EPI_PLAIN; }

template <class AADDR, class BADDR, int BK, int VEC, int EPI>
int launch_tile(int tile, const typename AADDR::P& pa, const typename BADDR::P& pb, const ConvG& cg, const Dims& d, const Epi& ep,
                int chunks, hipStream_t st) {
    // the slow epilogue exists for three row heights only (any tile is correct for any shape): fewer instantiations
    if constexpr (EPI == EPI_SLOW) tile = tile == T51 || tile == T32 ? T31 : tile == T23 ? T21 : tile;
    if constexpr (!tile_fits<AADDR, BADDR, 3, 1, 1, 4, BK, VEC>()) tile = tile == T31 ? T21 : tile;
    if constexpr (!tile_fits<AADDR, BADDR, 5, 1, 1, 4, BK, VEC>()) tile = tile == T51 ? T21 : tile;
    if constexpr (!tile_fits<AADDR, BADDR, 2, 3, 2, 2, BK, VEC>()) tile = tile == T23 ? T21 : tile;
    if constexpr (!tile_fits<AADDR, BADDR, 3, 2, 2, 2, BK, VEC>()) tile = tile == T32 ? T21 : tile;
    if constexpr (mf_ok<AADDR, BADDR, BK, EPI>()) {
        switch (tile) {
// (the two largest tiles keep two producer register sets instead of three: with three the producers spill inside their K loop)
#define MC_MF_CASE(T, MT_, NT_)                                                                                                       \
            case T:                                                                                                                   \
                if constexpr (tile_fits<AADDR, BADDR, MT_, NT_, 2, 2, BK, VEC, 16>())                                                 \
                    return launch_cfg<AADDR, BADDR, MT_, NT_, 2, 2, BK, VEC, EPI, (MT_ * NT_ >= 30 ? 2 : MC_NS), 1, 1>(pa, pb, cg, d, ep, chunks, st); \
                break;
            MC_MF_CASE(M56, 5, 6)
            MC_MF_CASE(M35, 3, 5)
            MC_MF_CASE(M66, 6, 6)
            MC_MF_CASE(M36, 3, 6)
            MC_MF_CASE(M33, 3, 3)
#undef MC_MF_CASE
            default: break;
        }
    }
    if (is_mf_tile(tile)) tile = T31;   // asked for a product that has no 16x16x4 form
    switch (tile) {
        case T11: return launch_cfg<AADDR, BADDR, 1, 1, 1, 4, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
        case T21: return launch_cfg<AADDR, BADDR, 2, 1, 1, 4, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
        // four waves stacked over ONE 32-column strip: products with at most 64 columns (weight gradients of layers with
        // 32 / 64 filters) would otherwise run 128-column tiles that are 3/4 or 1/2 padding
        case T14: return launch_cfg<AADDR, BADDR, 1, 1, 4, 1, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
        case T24: return launch_cfg<AADDR, BADDR, 2, 1, 2, 2, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);   // 2x2 waves, 64 columns
        default: break;
    }
#ifdef MC_IGEMM_KQ_TILE   // see heuristic_tile: measured slower than the padded 4-tile shapes, not built by default
    if constexpr (BK == 32 && EPI == EPI_PLAIN)   // the K-split tile: long-k products (weight gradients) only
        if (tile == T33K) return launch_cfg<AADDR, BADDR, 3, 3, 1, 2, BK, VEC, EPI, MC_NS, 2>(pa, pb, cg, d, ep, chunks, st);
#endif
    if constexpr (EPI != EPI_SLOW) {
        switch (tile) {
            case T51:
                if constexpr (tile_fits<AADDR, BADDR, 5, 1, 1, 4, BK, VEC>()) return launch_cfg<AADDR, BADDR, 5, 1, 1, 4, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
                break;
            case T23:
                if constexpr (tile_fits<AADDR, BADDR, 2, 3, 2, 2, BK, VEC>()) return launch_cfg<AADDR, BADDR, 2, 3, 2, 2, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
                break;
            case T32:
                if constexpr (tile_fits<AADDR, BADDR, 3, 2, 2, 2, BK, VEC>()) return launch_cfg<AADDR, BADDR, 3, 2, 2, 2, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
                break;
            default: break;
        }
    }
    if constexpr (tile_fits<AADDR, BADDR, 3, 1, 1, 4, BK, VEC>()) return launch_cfg<AADDR, BADDR, 3, 1, 1, 4, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
    return launch_cfg<AADDR, BADDR, 2, 1, 1, 4, BK, VEC, EPI>(pa, pb, cg, d, ep, chunks, st);
}

// Static rule (measured on MI355X, tools/tile_sweep.py over the NIN shapes): 128-column tiles (one 32-column
// strip per consumer wave) win almost everywhere; exactly 160 rows (NIN cccp1 / cccp2) take one 160-row tile so
// the big operand is read once; 192-column tiles (2x2 waves, 3 strips each) when they cut the padding of the
// column extent (F = 192 weight gradients: 256 -> 192).
static const int kRowsOf[NTILES + 1] = {32, 64, 96, 160, 128, 192, 128, 128, 160, 96, 192, 96, 96, 96},
                 kColsOf[NTILES + 1] = {128, 128, 128, 128, 192, 128, 32, 64, 192, 160, 192, 192, 96, 192};
inline int heuristic_tile4(const Dims& d, int chunks);
// The 16x16x4 tile for a product whose WHOLE output is one such tile (75-192 rows by 129-192 columns: the dense weight gradients of
// the cccp layers, the first conv layer's weight gradient), or -1.  A pure function of the extents: chunk plans are made with it.
// chunks: K chunks of the launch.  With too few of them for one workgroup per chunk to fill the device the output is cut in two
// 96-row halves, then in 96 x 96 quarters.
inline int whole_output_mf_tile(int Mg, int Ng, int chunks = 1 << 20) {
    if (!mf_whole_output(Mg, Ng)) return -1;
    if (Ng <= 160 && Mg <= 96) return M35;
    if (Mg <= 96) return M36;
    if (4L * chunks >= 3L * num_cus()) return Mg <= 160 ? M56 : M66;
    return 8L * chunks >= 3L * num_cus() ? M36 : M33;
}
// kqOk: the call site has the 96 x 192 K-split tile (long-k products).  It is taken when it cuts the padded work of the
// 4-tile rule's pick by 15 % or more -- decided from the extents alone, see the kernel's comment.
inline int heuristic_tile(const Dims& d, int chunks, bool kqOk = false, bool mfOk = false) {
    if (mfOk && d.batches == 1) {
        const int m = whole_output_mf_tile(d.Mg, d.Ng, chunks);
        if (m >= 0) return m;
    }
    const int t = heuristic_tile4(d, chunks);
#ifndef MC_IGEMM_KQ_TILE
    // Measured on MI355X (NIN weight gradients, tools/build_variant.sh kq -DMC_IGEMM_KQ_TILE): the tile removes the padding
    // (cccp1 160 x 192: 62 % -> 83 % useful MFMA work) but is no faster -- cccp1 112 -> 112 us, cccp2 64 -> 106, conv1 69 -> 78,
    // cccp3 43 -> 50: these products are bound by their k-contiguous loaders (128 bytes a row and K step from 288-320 rows
    // 512 KB apart) and one resident workgroup per CU, not by the matrix pipe.  Kept as a build option, off by default.
    return t;
#endif
    static const bool nokq = getenv("MC_IGEMM_NOKQ") != nullptr;
    if (!kqOk || nokq || d.batches != 1 || d.Ng <= 64) return t;
    auto padded = [&](int tile) { return (double)ceil_div(d.Mg, kRowsOf[tile]) * kRowsOf[tile] * ceil_div(d.Ng, kColsOf[tile]) * kColsOf[tile]; };
    return padded(T33K) <= 0.85 * padded(t) ? (int)T33K : t;
}
inline int heuristic_tile4(const Dims& d, int chunks) {
    const long z = (long)chunks * d.groups * d.batches;
    const int mt32 = ceil_div(d.Mg, 32);
    const long colTiles = ceil_div(d.Ng, 128);
    if (d.Ng <= 32) return T14;   // one 32-column strip: a 128-column tile would be at least 3/4 padding
    if (d.Ng <= 64 && d.Mg >= 128) return T24;   // two strips (weight gradients of 64-filter layers)
    if (mt32 == 5 && colTiles * z >= num_cus()) return T51;
    const long pad128 = (long)ceil_div(d.Ng, 128) * 128, pad192 = (long)ceil_div(d.Ng, 192) * 192;
    if (pad192 < pad128 && mt32 >= 4 && (long)ceil_div(d.Mg, 128) * ceil_div(d.Ng, 192) * z >= num_cus() / 2) return T23;
    // rows: the largest of {96, 64, 32} that divides the 32-padded row count, shrunk while the grid would leave CUs idle
    int MTsel = (mt32 % 3 == 0) ? 3 : (mt32 % 2 == 0 || mt32 >= 8) ? 2 : 1;
    while (MTsel > 1 && (long)ceil_div(d.Mg, 32 * MTsel) * colTiles * z < num_cus()) --MTsel;
    return MTsel == 3 ? T31 : MTsel == 2 ? T21 : T11;
}

struct TuneKey {
    const void* kind;   // one per (loaders, BK, VEC) instantiation
    int v[20];
    bool operator<(const TuneKey& o) const { return kind != o.kind ? kind < o.kind : memcmp(v, o.v, sizeof(v)) < 0; }
};
static std::map<TuneKey, int> g_tuned;
static std::mutex g_tuneMu;

template <class AADDR, class BADDR, int BK, int VEC, int EPI>
int launch_rows_cols_epi(const typename AADDR::P& pa, const typename BADDR::P& pb, const ConvG& cg, const Dims& d,
                     const Epi& ep, int chunks, hipStream_t st) {
    // tuning knob (benchmarks only): MC_IGEMM_TILE="MT,NT" forces the wave tile
    static const char* forced = getenv("MC_IGEMM_TILE");
    if (forced && forced[0] && forced[1] == ',') {
        static const int map[14][3] = {{1, 1, T11}, {2, 1, T21}, {3, 1, T31}, {5, 1, T51}, {2, 3, T23}, {3, 2, T32}, {4, 0, T14}, {4, 1, T24}, {3, 3, T33K},
                                       {5, 6, M56}, {3, 5, M35}, {6, 6, M66}, {3, 6, M36}, {9, 9, M33}};
        for (auto& m : map)
            if (forced[0] - '0' == m[0] && forced[2] - '0' == m[1]) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(m[2], pa, pb, cg, d, ep, chunks, st);
    }
    constexpr bool MFOK = mf_ok<AADDR, BADDR, BK, EPI>();
    static const bool nomf = getenv("MC_IGEMM_NOMF") != nullptr;   // A/B switch: without the 16x16x4 tiles (and their chunk plans)
    const bool mfNow = MFOK && !nomf && !ep.bitsOut && !ep.bitsIn;
    const int rule = heuristic_tile(d, chunks, BK == 32 && EPI == EPI_PLAIN && !ep.sumOut, mfNow);   // (the K-split tile has no k-sums)
    if (rule == T33K) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(rule, pa, pb, cg, d, ep, chunks, st);   // by rule only: its sum order is its own
    // k-sums (bias gradients) are formed in one order by the 32x32 tiles and in another by the 16x16x4 tiles: with sums asked for the
    // family is the rule's (a pure function of the shape), and the timing trials below stay inside it
    const bool sumsFixMf = MFOK && ep.sumOut != nullptr;
    if (sumsFixMf && is_mf_tile(rule)) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(rule, pa, pb, cg, d, ep, chunks, st);
    static const char* at = getenv("MC_IGEMM_AUTOTUNE");
    static const int mode = at ? atoi(at) : 1;
    if (mode == 0) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(rule, pa, pb, cg, d, ep, chunks, st);

    static char kindTag;
    int dev = 0;
    (void)hipGetDevice(&dev);
    TuneKey key;
    key.kind = &kindTag;
    const int vals[20] = {d.Mg, d.Ng, d.Kg, d.kChunk, d.groups, d.batches, d.colB, chunks, cg.N, cg.H, cg.W, cg.fc, cg.fs, cg.pad,
                          cg.stride, cg.MY, cg.MX, cg.F, cg.local, dev * 4 + EPI};
    memcpy(key.v, vals, sizeof(vals));
    {
        std::lock_guard<std::mutex> lk(g_tuneMu);
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(it->second, pa, pb, cg, d, ep, chunks, st);
    }
    // an accumulating call (scaleTargets != 0) is not idempotent: serve it by the rule and let a later plain call tune;
    // so is a call from an entry point that must not create events or block (mc_conv_weight_grad_noalloc)
    if (EPI == EPI_SLOW || ep.beta != 0.f || g_tuneInhibit || g_describeBuf) return launch_tile<AADDR, BADDR, BK, VEC, EPI>(rule, pa, pb, cg, d, ep, chunks, st);

    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return launch_tile<AADDR, BADDR, BK, VEC, EPI>(rule, pa, pb, cg, d, ep, chunks, st);
    int best = rule;
    float bestMs = 1e30f, ms[NTILES];
    for (int t = 0; t < NTILES; ++t) {
        ms[t] = -1.f;
        // skip tiles that would compute more than 4x the needed rows (a 160-row tile for a 10-row product)
        if (t != rule && (long)ceil_div(d.Mg, kRowsOf[t]) * kRowsOf[t] > 4L * ((d.Mg + 31) / 32 * 32)) continue;
        if ((t == T14 || t == T24) && d.Ng > 64) continue;   // the narrow tile only where 128 columns are mostly padding
        if (is_mf_tile(t) && (!mfNow || sumsFixMf || d.batches != 1 || d.Ng <= 64 || d.Mg <= 32)) continue;   // 16x16x4 tiles: k-contiguous products with a wide output
        int rc = launch_tile<AADDR, BADDR, BK, VEC, EPI>(t, pa, pb, cg, d, ep, chunks, st);   // warm (and validates the launch)
        if (rc != MC_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 2; ++r) (void)launch_tile<AADDR, BADDR, BK, VEC, EPI>(t, pa, pb, cg, d, ep, chunks, st);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float el = 0.f;
        if (hipEventElapsedTime(&el, e0, e1) != hipSuccess) continue;
        ms[t] = el * 0.5f;
        if (ms[t] < bestMs) { bestMs = ms[t]; best = t; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // hysteresis: the rule's tile stays unless another one is at least 3 % faster -- two tiles within timing noise of each other
    // would otherwise trade places from run to run (NIN conv2 forward: 96x128 and 192x128), and with them the launch that
    // profiles and benchmarks talk about
    if (ms[rule] >= 0.f && ms[rule] <= bestMs * 1.03f) best = rule;
    if (mode >= 2) {
        fprintf(stderr, "[mc_igemm autotune] %s rows %d cols %d k %d chunks %d batches %d:",
                AADDR::KC ? (BADDR::KC ? "kc/kc" : "kc/nat") : (BADDR::KC ? "nat/kc" : "nat/nat"),
                d.Mg, d.Ng, d.Kg, chunks, d.batches);
        for (int t = 0; t < NTILES; ++t)
            if (ms[t] >= 0.f) fprintf(stderr, " %s=%.1fus%s", kTileName[t], ms[t] * 1e3f, t == best ? "*" : (t == rule ? "(rule)" : ""));
        fprintf(stderr, "\n");
    }
    {
        std::lock_guard<std::mutex> lk(g_tuneMu);
        g_tuned[key] = best;
    }
    return MC_OK;   // the timed launches already produced the result (identical for every tile)
}

// Picks the epilogue flavour.  BITS / BOUT: whether this call site instantiates the kernels that read / write relu bit masks
// (dense backward products read them; dense and conv forward products write them).
template <class AADDR, class BADDR, int BK, int VEC, bool BITS = false, bool BOUT = false>
int launch_rows_cols(const typename AADDR::P& pa, const typename BADDR::P& pb, const ConvG& cg, const Dims& d,
                     const Epi& ep, int chunks, hipStream_t st) {
    const bool plain = ep.beta == 0.f && !ep.mask;
    if (ep.bitsIn || ep.bitsOut) {
        if (!plain || d.groups != 1 || d.batches != 1 || (ep.bitsIn && ep.bitsOut) || ep.chunkStride != 0) {
            set_error("mc_igemm: relu bit masks need scaleTargets == 0, one group, no float mask and an unsplit product");
            return MC_EINVAL;
        }
        if constexpr (BITS)
            if (ep.bitsIn) return launch_rows_cols_epi<AADDR, BADDR, BK, VEC, EPI_BITS>(pa, pb, cg, d, ep, chunks, st);
        if constexpr (BOUT)
            if (ep.bitsOut) return launch_rows_cols_epi<AADDR, BADDR, BK, VEC, EPI_PLAIN_BITSOUT>(pa, pb, cg, d, ep, chunks, st);
        set_error("mc_igemm: relu bit masks are not available for this product");
        return MC_EINVAL;
    }
    if (!plain) return launch_rows_cols_epi<AADDR, BADDR, BK, VEC, EPI_SLOW>(pa, pb, cg, d, ep, chunks, st);
    return launch_rows_cols_epi<AADDR, BADDR, BK, VEC, EPI_PLAIN>(pa, pb, cg, d, ep, chunks, st);
}

inline float finv(int d) { return 1.0f / (float)d; }

int make_convg(ConvG& cg, int N, int C, int H, int W, int F, int fs, int pad, int stride, int MY, int MX, int groups,
               int local, const char* who) {
    MC_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && F > 0 && fs > 0 && stride > 0 && MY > 0 && MX > 0 && groups > 0,
               "%s: non-positive dimension", who);
    MC_REQUIRE(pad <= 0, "%s: paddingStart must be <= 0 (got %d)", who, pad);
    MC_REQUIRE(C % groups == 0 && F % groups == 0, "%s: channels (%d) and filters (%d) must be divisible by groups (%d)",
               who, C, F, groups);
    const long biggest = (long)N * (long)((long)C * H * W > (long)F * MY * MX ? (long)C * H * W : (long)F * MY * MX);
    MC_REQUIRE(biggest < (1L << 30), "%s: tensor with %ld elements exceeds the 4 GiB buffer-descriptor range", who, biggest);
    // fdiv() decodes the K index with a float reciprocal, exact below 2^22
    MC_REQUIRE((long)MY * MX * (local ? (long)(C / groups) * fs * fs : 1) * F < (1L << 30), "%s: filter tensor exceeds the 4 GiB buffer-descriptor range", who);
    MC_REQUIRE((long)MY * MX * N < (1L << 22) && (long)(C / groups) * fs * fs < (1L << 22) && (long)fs * fs * (F / groups) < (1L << 22),
               "%s: modules*images, channels*taps and taps*filters must each be < 2^22", who);
    cg.N = N; cg.H = H; cg.W = W; cg.fc = C / groups; cg.fs = fs; cg.fp = fs * fs; cg.pad = pad; cg.stride = stride;
    cg.MY = MY; cg.MX = MX; cg.M = MY * MX; cg.F = F; cg.fpg = F / groups; cg.local = local; cg.kcUniform = 0;
    cg.inv_fp = finv(cg.fp); cg.inv_fs = finv(fs); cg.inv_fpg = finv(cg.fpg); cg.inv_N = finv(N); cg.inv_MX = finv(MX);
    cg.inv_W = finv(W);
    cg.inv_stride = finv(stride);
    return MC_OK;
}

}  // namespace
}  // namespace mc
