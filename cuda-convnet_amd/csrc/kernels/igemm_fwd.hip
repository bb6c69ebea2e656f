// igemm_fwd.hip -- conv / local forward instantiations, the K-split plan and the autotune inhibit flag (igemm_core.h holds the engine).
#include "igemm_core.h"

namespace mc {
MC_STAMP_API(fwd)

thread_local bool g_tuneInhibit = false;
thread_local char* g_describeBuf = nullptr;
thread_local int g_describeLen = 0;
void set_describe(char* buf, int len) { g_describeBuf = buf; g_describeLen = len; }
bool describing() { return g_describeBuf != nullptr; }
void describe_write(const char* text) { if (g_describeBuf) snprintf(g_describeBuf, (size_t)g_describeLen, "%s", text); }
// While set (per host thread), a shape that has not been tuned yet runs the static rule's tile instead of the timed trial.
void set_tune_inhibit(bool on) { g_tuneInhibit = on; }
bool tune_inhibited() { return g_tuneInhibit; }

// =================================================================================
// conv forward: C[f][(m,n)] ; Mg = F/groups, Ng = M*N, Kg = fc*fp
// =================================================================================
int conv_filter_acts_impl(const float* images, const float* filters, float* targets, const float* biases, int neuron,
                          int N, int C, int H, int W, int F, int fs, int pad, int stride, int MY, int MX, int groups,
                          int local, float scaleTargets, float scaleOutput, float* slabs, int kChunk, unsigned* reluBits, hipStream_t st) {
    ConvG cg;
    int rc = make_convg(cg, N, C, H, W, F, fs, pad, stride, MY, MX, groups, local, local ? "localFilterActs" : "convFilterActs");
    if (rc) return rc;
    int chunks = 1;
    // shared filters: one product over all M*N columns.  Unshared (filter_acts.cu conv=false): M products in one
    // launch, batch = module, each over its own N columns with its own filter bank W[m].
    Dims d{cg.fpg, local ? N : cg.M * N, cg.fc * cg.fp, cg.fc * cg.fp, groups, 0, 0, local ? cg.M : 1, local ? N : 0};
    Epi ep{targets, cg.M * N, cg.fpg, 0, 0, scaleOutput, scaleTargets, biases, neuron, nullptr, 0, (unsigned)((size_t)F * cg.M * N * 4), 0};
    if (slabs && kChunk > 0 && !local) {   // split K: every chunk writes a plain slab, the caller reduces and applies the epilogue
        d.kChunk = kChunk;
        chunks = ceil_div(d.Kg, kChunk);
        ep = Epi{slabs, cg.M * N, cg.fpg, 0, (long)F * cg.M * N, 1.f, 0.f, nullptr, MC_NEURON_NONE, nullptr, 0, (unsigned)((size_t)F * cg.M * N * 4), 0};
    }
    if (reluBits && chunks == 1) {
        ep.bitsOut = reluBits;
        ep.wBytes = (unsigned)((size_t)ceil_div(F, 32) * cg.M * N * 4);
    }
    MatP pa{filters, F, cg.fpg, cg.fpg, (unsigned)((size_t)(local ? cg.M : 1) * cg.fc * cg.fp * F * 4), local ? cg.fc * cg.fp * F : 0};  // filters[k*F + g*fpg + f]
    ConvP pb{images, cg.M * N, (unsigned)((size_t)C * H * W * N * 4)};
    const bool vec = (N % 4 == 0) && (cg.fpg % 4 == 0) && (F % 4 == 0) && aligned16(images) && aligned16(filters);
    if (vec) return launch_rows_cols<MatNatAddr, PatchNatAddr, MC_BK, 4, false, true>(pa, pb, cg, d, ep, chunks, st);
    return launch_rows_cols<MatNatAddr, PatchNatAddr, MC_BK, 1, false, true>(pa, pb, cg, d, ep, chunks, st);
}

// K split for the forward / backward-data products of layers with few columns (NIN conv3: 64 modules x 128 images):
// with 96x128 tiles the grid would cover half the device, so K is cut into as many chunks as bring it to about one
// workgroup per CU (at most 8, each >= 512 long, a multiple of 32); slabs + a fixed-order reduction as for the
// weight gradients.  A pure function of the shape and the CU count.  Returns the chunk count (1 = no split).
int conv_split_plan(int rows, long cols, int K, int groups, int* kChunkOut) {
    const long tiles = (long)ceil_div(rows, 96) * ceil_div(cols, 128) * groups;
    // (measured and dropped: aiming at 2-4 workgroups per CU for products with <= 64 rows -- layers-80sec conv2 forward, 256 tiles
    // -- by splitting further: no faster, the extra reduction eats what the split gains)
    long chunks = num_cus() / tiles;
    // Between half and all of the CUs covered, one workgroup each: every SIMD's matrix pipe hangs on a single wave's LDS
    // latency and nothing overlaps the workgroups' prologues and stores.  With a long K, two half-K workgroups per CU and a
    // reduction are faster (NIN conv2 backward-data, 256 tiles x K = 4800: 296 -> 271 us).  Aiming at two per CU everywhere
    // was measured too and loses on the short products (conv3 forward 69 -> 75 us).
    if (chunks < 2 && 2 * tiles > num_cus() && tiles <= num_cus() && K >= 2048) chunks = 2;
    *kChunkOut = 0;
    if (chunks < 2 || K < 512) return 1;
    if (chunks > 8) chunks = 8;
    long kc = (ceil_div(K, chunks) + 31) / 32 * 32;
    const long kcMin = K >= 1024 ? 512 : 256;   // mid-size K (layers-80sec conv3 forward: 800 on 64 tiles) splits into shorter chunks
    if (kc < kcMin) kc = kcMin;
    if (kc >= K) return 1;
    *kChunkOut = (int)kc;
    return ceil_div(K, kc);
}

}  // namespace mc
