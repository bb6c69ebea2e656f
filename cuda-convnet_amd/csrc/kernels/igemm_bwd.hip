// igemm_bwd.hip -- conv / local backward-data instantiations (igemm_core.h holds the engine).
#include "igemm_core.h"

namespace mc {
MC_STAMP_API(bwd)

// =================================================================================
// conv bwd-data: C[c][(px,n)] ; Mg = fc, Ng = H*W*N, Kg = fp*fpg
// =================================================================================
int conv_img_acts_impl(const float* hidActs, const float* filters, float* targets, int N, int C, int H, int W, int F,
                       int fs, int pad, int stride, int MY, int MX, int groups, int local, float scaleTargets,
                       float scaleOutput, float* slabs, int kChunk, hipStream_t st) {
    ConvG cg;
    int rc = make_convg(cg, N, C, H, W, F, fs, pad, stride, MY, MX, groups, local, local ? "localImgActs" : "convImgActs");
    if (rc) return rc;
    int chunks = 1;
    // unshared (img_acts.cu conv=false): one product per image pixel (batch = pixel), whose A operand gathers, tap
    // by tap, the filter bank of the module that sees the pixel at that tap (FiltTAddr)
    Dims d{cg.fc, local ? N : H * W * N, cg.fp * cg.fpg, cg.fp * cg.fpg, groups, 0, 0, local ? H * W : 1, local ? N : 0};
    Epi ep{targets, H * W * N, cg.fc, 0, 0, scaleOutput, scaleTargets, nullptr, MC_NEURON_NONE, nullptr, 0, (unsigned)((size_t)C * H * W * N * 4), 0};
    if (slabs && kChunk > 0 && !local) {
        d.kChunk = kChunk;
        chunks = ceil_div(d.Kg, kChunk);
        ep = Epi{slabs, H * W * N, cg.fc, 0, (long)C * H * W * N, 1.f, 0.f, nullptr, MC_NEURON_NONE, nullptr, 0, (unsigned)((size_t)C * H * W * N * 4), 0};
    }
    ConvP pa{filters, cg.fc, (unsigned)((size_t)(local ? cg.M : 1) * cg.fc * cg.fp * F * 4)};
    ConvP pb{hidActs, H * W * N, (unsigned)((size_t)F * cg.M * N * 4)};
    // group offset of the A operand rows: channel c of group g uses filter rows (cl*fp+p), columns g*fpg+fl
    const bool vec = (N % 4 == 0) && (cg.fpg % 4 == 0) && (F % 4 == 0) && aligned16(hidActs) && aligned16(filters);
    if (stride == 1 && groups == 1 && !local) {
        // with one group the transposed-filter operand is a plain k-contiguous matrix: row c, k = p*F + f at c*fp*F + k
        MatP pm{filters, cg.fp * F, cg.fc, 0, pa.bytes, 0};
        if (vec) return launch_rows_cols<MatKcAddr, HidNatS1Addr, MC_BK, 4>(pm, pb, cg, d, ep, chunks, st);
        return launch_rows_cols<MatKcAddr, HidNatS1Addr, MC_BK, 1>(pm, pb, cg, d, ep, chunks, st);
    }
    if (vec) return launch_rows_cols<FiltTAddr, HidNatAddr, MC_BK, 4>(pa, pb, cg, d, ep, chunks, st);
    return launch_rows_cols<FiltTAddr, HidNatAddr, MC_BK, 1>(pa, pb, cg, d, ep, chunks, st);
}

}  // namespace mc
