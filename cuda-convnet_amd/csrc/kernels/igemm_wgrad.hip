// igemm_wgrad.hip -- conv / local weight-gradient instantiations and their chunk plan (igemm_core.h holds the engine).
#include "igemm_core.h"

namespace mc {
MC_STAMP_API(wgrad)

// =================================================================================
// conv bwd-weights: C[q][(c,p)][f] ; Mg = fc*fp, Ng = fpg, Kg = M*N, chunk = partialSum*N
// =================================================================================
int conv_weight_acts_impl(const float* images, const float* hidActs, float* targets, int N, int C, int H, int W, int F,
                          int fs, int pad, int stride, int MY, int MX, int groups, int partialSum, int kChunk, float scaleTargets,
                          float scaleOutput, float* sumOut, long sumStride, float sumAlpha, hipStream_t st) {
    ConvG cg;
    int rc = make_convg(cg, N, C, H, W, F, fs, pad, stride, MY, MX, groups, 0, "convWeightActs");
    if (rc) return rc;
    cg.kcUniform = (N % MC_BK_KC == 0) ? 1 : 0;   // chunks and tiles start at multiples of BK: a tile stays inside one module
    int chunks;
    if (kChunk > 0) {   // module-unaligned chunks of the (module, image) axis (mc_conv_weight_grad)
        MC_REQUIRE(kChunk % MC_BK_KC == 0, "convWeightActs: k chunk (%d) must be a multiple of %d", kChunk, MC_BK_KC);
        chunks = ceil_div((long)cg.M * N, kChunk);
    } else {
        if (partialSum <= 0) partialSum = cg.M;
        MC_REQUIRE(cg.M % partialSum == 0, "convWeightActs: partialSum (%d) must divide numModules (%d)", partialSum, cg.M);
        chunks = cg.M / partialSum;
        kChunk = partialSum * N;
    }
    Dims d{cg.fc * cg.fp, cg.fpg, cg.M * N, kChunk, groups, 0, 0, 1, 0};
    Epi ep{targets, F, 0, cg.fpg, (long)cg.fc * cg.fp * F, scaleOutput, scaleTargets, nullptr, MC_NEURON_NONE, nullptr, 0, (unsigned)((size_t)cg.fc * cg.fp * F * 4), 0};
    // by-product: chunk q's sum of hidActs over its (module, image) range, per filter -- the shared-bias gradient
    ep.sumOut = sumOut;
    ep.sumStride = sumStride;
    ep.sumAlpha = sumAlpha;
    ConvP pa{images, cg.fc * cg.fp, (unsigned)((size_t)C * H * W * N * 4)};
    MatP pb{hidActs, cg.M * N, cg.fpg, cg.fpg * cg.M * N, (unsigned)((size_t)F * cg.M * N * 4), 0};  // hid[(g*fpg+f)*M*N + k]
    const bool vec = (N % 4 == 0) && aligned16(images) && aligned16(hidActs);
    if (vec) return launch_rows_cols<PatchKcAddr, MatKcSumAddr, MC_BK_KC, 4>(pa, pb, cg, d, ep, chunks, st);
    return launch_rows_cols<PatchKcAddr, MatKcSumAddr, MC_BK_KC, 1>(pa, pb, cg, d, ep, chunks, st);
}

// Chunking of the (module, image) axis for mc_conv_weight_grad: as many chunks as keep about one workgroup per CU
// (the k-contiguous tiles need ~80-120 KB of LDS, one workgroup per CU; a grid of 2.4 "rounds" costs 3), each chunk
// a multiple of the K tile and at least 256 long.  A pure function of the shape and the CU count: the chunk count
// fixes the order of the sum, so it must not depend on timing.
int conv_weight_grad_plan(int N, int C, int F, int fs, int M, int groups, int* kChunkOut) {
    const int rows = (C / groups) * fs * fs, fpg = F / groups;
    const long K = (long)M * N;
    Dims d{rows, fpg, (int)K, (int)K, groups, 0, 0, 1, 0};
    // the tile the static rule would take for a grid that fills the device
    static const bool nomf = getenv("MC_IGEMM_NOMF") != nullptr;
    const int t = heuristic_tile(d, ceil_div(num_cus(), (long)ceil_div(rows, 128) * ceil_div(fpg, 128) * groups), true, !nomf);
    const long tiles = (long)ceil_div(rows, kRowsOf[t]) * ceil_div(fpg, kColsOf[t]) * groups;
    long chunks = num_cus() / tiles;
    if (chunks < 1) chunks = 1;
    static const char* forced = getenv("MC_WGRAD_CHUNKS");   // tuning knob (benchmarks only)
    if (forced && atoi(forced) > 0) chunks = atoi(forced);
    long kc = (ceil_div(K, chunks) + MC_BK_KC - 1) / MC_BK_KC * MC_BK_KC;
    if (kc < 256) kc = 256;
    if (kc > K) kc = (K + MC_BK_KC - 1) / MC_BK_KC * MC_BK_KC;
    *kChunkOut = (int)kc;
    return ceil_div(K, kc);
}

}  // namespace mc
