// igemm_gemm.hip -- dense product instantiations (FC / cccp) (igemm_core.h holds the engine).
#include "igemm_core.h"

namespace mc {
MC_STAMP_API(gemm)

// =================================================================================
// dense: C(m,n) = beta*C + alpha*op(A)*op(B)
// =================================================================================
int gemm_impl(const float* A, const float* B, float* Cm, int m, int n, int k, int transA, int transB, int lda, int ldb,
              int ldc, float beta, float alpha, const float* rowBias, int neuron, const float* mask, int ldmask, int chunks,
              long chunkStride, unsigned* bitsOut, const unsigned* bitsIn, float* sumOut, long sumStride, float sumAlpha, hipStream_t st) {
    ConvG cg{};
    int kChunk = k;
    if (chunks > 1) {
        kChunk = ceil_div(k, chunks);
        kChunk = (kChunk + 31) / 32 * 32;
        chunks = ceil_div(k, kChunk);
    }
    Dims d{m, n, k, kChunk, 1, 0, 0, 1, 0};
    Epi ep{Cm, ldc, 0, 0, chunkStride, alpha, beta, rowBias, neuron, mask, ldmask, (unsigned)(((size_t)(m - 1) * ldc + n) * 4),
           mask ? (unsigned)(((size_t)(m - 1) * ldmask + n) * 4) : 0u};
    ep.bitsOut = bitsOut;
    ep.bitsIn = bitsIn;
    ep.wBytes = (unsigned)(((size_t)(ceil_div(m, 32) - 1) * ldc + n) * 4);
    if (sumOut && !(!transA && transB)) {
        set_error("mc_gemm: row sums of A come with the weight-gradient orientation (transA = 0, transB = 1) only");
        return MC_EINVAL;
    }
    ep.sumOut = sumOut;   // by-product: chunk q's sum of A over its k range, per row
    ep.sumStride = sumStride;
    ep.sumAlpha = sumAlpha;
    // operand extents in bytes: (rows - 1) * ld + row length
    const size_t ea = transA ? (size_t)(k - 1) * lda + m : (size_t)(m - 1) * lda + k;
    const size_t eb = transB ? (size_t)(n - 1) * ldb + k : (size_t)(k - 1) * ldb + n;
    MatP pa{A, lda, m, 0, (unsigned)(ea * 4), 0};
    MatP pb{B, ldb, n, 0, (unsigned)(eb * 4), 0};
    const bool al = aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0;
    // vector loads need the contiguous extent of each operand to be a multiple of 4
    const bool va = transA ? (m % 4 == 0) : (k % 4 == 0);
    const bool vb = transB ? (k % 4 == 0) : (n % 4 == 0);
    const bool vec = al && va && vb;
#define MC_GEMM_CASE(AA, BB, BI, BO)                                                       \
    (vec ? launch_rows_cols<AA, BB, MC_BK, 4, BI, BO>(pa, pb, cg, d, ep, chunks, st)          \
         : launch_rows_cols<AA, BB, MC_BK, 1, BI, BO>(pa, pb, cg, d, ep, chunks, st))
    // relu bit masks: read by the backward product (weights transposed: transA), written by the forward product
    if (transA && !transB) return MC_GEMM_CASE(MatNatAddr, MatNatAddr, true, false);
    if (!transA && !transB) return MC_GEMM_CASE(MatKcAddr, MatNatAddr, false, true);
    if (!transA && transB)
        return vec ? launch_rows_cols<MatKcSumAddr, MatKcAddr, MC_BK_KC, 4>(pa, pb, cg, d, ep, chunks, st)
                   : launch_rows_cols<MatKcSumAddr, MatKcAddr, MC_BK_KC, 1>(pa, pb, cg, d, ep, chunks, st);
    return MC_GEMM_CASE(MatNatAddr, MatKcAddr, false, false);
#undef MC_GEMM_CASE
}

}  // namespace mc
