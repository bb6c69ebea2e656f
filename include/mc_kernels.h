/*
 * mc_kernels.h -- C-ABI of libmckernels.so: the MI355X (gfx950) kernels behind the
 * cuda-convnet training step.  One entry point per reference host wrapper; the
 * comment on each cites the wrapper it replaces (paths relative to the reference
 * tree).  Plain device pointers and sizes only: no C++ types, no torch types.
 *
 * Conventions (the reference's, kept):
 *   - every activation / gradient matrix is float32, row-major (features, numImages),
 *     i.e. (channel, y, x, image) with the image index innermost
 *     (Kernel/src/conv/filter_acts.cu:39-43);
 *   - conv filters are (filterChannels * filterPixels, numFilters) row-major, unshared
 *     (local) filters (numModules * filterChannels * filterPixels, numFilters);
 *   - paddingStart <= 0 is the (negated) padding, as the reference passes it
 *     (Python/layer.py:959);
 *   - target = scaleTargets * target + scaleOutput * result; scaleTargets == 0 means the
 *     target is write-only (never read, may hold NaN).
 *   - all launches are asynchronous on `stream` (a hipStream_t; NULL = default stream).
 *
 * Every function returns 0 on success and a negative MC_E* code on a rejected call
 * (the reference assert()s / exit()s instead: Kernel/include/nvmatrix/nvmatrix.cuh:54-59).
 * mc_last_error() returns a static description of the last rejection in this thread.
 */
#ifndef MC_KERNELS_H
#define MC_KERNELS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mcStream_t; /* hipStream_t */

#define MC_OK 0
#define MC_EINVAL (-1)     /* shape / argument rejected */
#define MC_ELAUNCH (-2)    /* HIP launch error */
#define MC_EWORKSPACE (-3) /* workspace too small */

const char* mc_last_error(void);
/* Library / device identification: returns the gfx arch string of the current device. */
int mc_device_info(char* archName, int archNameLen, int* numCUs);

/* ---------------- conv / local contractions (fp32 MFMA implicit GEMM) ----------------
 * Replace Kernel/include/conv/cudaconv2.cuh:36-78:
 *   convFilterActs  (Kernel/src/conv/filter_acts.cu:2206-2211)
 *   localFilterActs (filter_acts.cu:2219-2224)
 *   convImgActs     (Kernel/src/conv/img_acts.cu:2040-2044)
 *   localImgActs    (img_acts.cu:2051-2056)
 *   convWeightActs  (Kernel/src/conv/weight_acts.cu:1092-1096)
 *   localWeightActs (weight_acts.cu:1103-1108)
 * images (numImgColors*imgSizeY*imgSizeX, numImages); hidActs/targets
 * (numFilters*numModulesY*numModulesX, numImages).
 *
 * mc_conv_filter_acts_fused additionally adds biases[f] (shared biases, the
 * reshape+addVector of Kernel/src/layers/layer.cu:481-484) and applies `neuron`
 * (MC_NEURON_NONE / MC_NEURON_RELU) in the epilogue; bit-identical to the unfused sequence.
 */
#define MC_NEURON_NONE 0
#define MC_NEURON_RELU 1

int mc_conv_filter_acts(const float* images, const float* filters, float* targets, int numImages,
                        int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                        int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                        int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
int mc_conv_filter_acts_fused(const float* images, const float* filters, float* targets,
                              const float* biases, int neuron, int numImages, int numImgColors,
                              int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                              int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                              int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
int mc_local_filter_acts(const float* images, const float* filters, float* targets, int numImages,
                         int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                         int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                         int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
int mc_conv_img_acts(const float* hidActs, const float* filters, float* targets, int numImages,
                     int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                     int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                     int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
int mc_local_img_acts(const float* hidActs, const float* filters, float* targets, int numImages,
                      int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                      int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                      int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
/* targets: ((numModules/partialSum) * filterChannels*filterPixels, numFilters); chunk q holds
 * the sum over modules [q*partialSum, (q+1)*partialSum).  partialSum == 0: all modules. */
int mc_conv_weight_acts(const float* images, const float* hidActs, float* targets, int numImages,
                        int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                        int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                        int numGroups, int partialSum, float scaleTargets, float scaleOutput,
                        mcStream_t stream);
int mc_local_weight_acts(const float* images, const float* hidActs, float* targets, int numImages,
                         int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                         int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                         int numGroups, float scaleTargets, float scaleOutput, mcStream_t stream);
/* ConvLayer::bpropWeights as a whole (layer.cu:503-523): convWeightActs into chunks of the (module, image) axis
 * followed by the reduction of the chunks, weightGrad = scaleTargets*weightGrad + scaleOutput * sum.  Unlike
 * mc_conv_weight_acts the chunks need not hold whole modules: the library sizes them so the launch fills the
 * device in one round (a function of the shape and the CU count only, so results are run-to-run identical).
 * workspace: mc_conv_weight_grad_workspace_bytes(...) bytes (0 = none needed), 16-byte aligned. */
size_t mc_conv_weight_grad_workspace_bytes(int numImages, int numImgColors, int numFilters,
                                           int filterSize, int numModules, int numGroups);
int mc_conv_weight_grad(const float* images, const float* hidActs, float* weightGrad, int numImages,
                        int numImgColors, int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                        int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                        int numGroups, float scaleTargets, float scaleOutput, void* workspace,
                        size_t workspaceBytes, mcStream_t stream);
/* ConvLayer::bpropBiases + bpropWeights in one pass over hidActs (layer.cu:489-523): the weight-gradient product reads every
 * element of hidActs anyway, so the gradient of biases shared per filter -- the reference's separate
 * `_biases->getGrad().addSum(v reshaped (numFilters, modules*images), 1, ...)`, layer.cu:491-497 -- comes out of the same kernel:
 * biasGrad[f] = biasScale * sum over modules and images of hidActs[f] (overwritten), weightGrad as mc_conv_weight_grad.
 * The sum is formed in a fixed order that depends on the shape only.  workspace: mc_conv_weight_grad_bias_workspace_bytes. */
size_t mc_conv_weight_grad_bias_workspace_bytes(int numImages, int numImgColors, int numFilters,
                                                int filterSize, int numModules, int numGroups);
int mc_conv_weight_grad_bias(const float* images, const float* hidActs, float* weightGrad, float* biasGrad,
                             int numImages, int numImgColors, int imgSizeY, int imgSizeX, int numFilters,
                             int filterSize, int paddingStart, int moduleStride, int numModulesY,
                             int numModulesX, int numGroups, float scaleTargets, float scaleOutput,
                             float biasScale, void* workspace, size_t workspaceBytes, mcStream_t stream);

/* mc_conv_weight_grad for callers that cannot allocate device memory or hand over a workspace: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  The split-K slabs live
 * in a device array owned by the library (MC_NOALLOC_SCRATCH_BYTES per GPU the library runs on, held until the process
 * ends); the call performs no hipMalloc / hipFree, creates no events and never blocks (the tile autotuner is skipped:
 * a shape not tuned yet runs the static rule's tile -- same bits either way).  Output is bit-identical to
 * mc_conv_weight_grad.  A shape whose slabs exceed the array (mc_conv_weight_grad_workspace_bytes(...) >
 * mc_noalloc_scratch_bytes()) is rejected with MC_EWORKSPACE.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (different streams / threads) race; order them on one stream. */
#define MC_NOALLOC_SCRATCH_BYTES ((size_t)32 << 20)
size_t mc_noalloc_scratch_bytes(void);
int mc_conv_weight_grad_noalloc(const float* images, const float* hidActs, float* weightGrad,
                                int numImages, int numImgColors, int imgSizeY, int imgSizeX,
                                int numFilters, int filterSize, int paddingStart, int moduleStride,
                                int numModulesY, int numModulesX, int numGroups, float scaleTargets,
                                float scaleOutput, mcStream_t stream);

/* Split-K scratch of mc_conv_filter_acts(_fused) / mc_conv_img_acts (layers with few modules, e.g. NIN conv3: the K
 * axis is cut so the launch fills the device).  Their signatures are the reference's host wrappers' and carry no
 * workspace, so the library keeps one grow-only buffer per (device, stream).  mc_conv_split_workspace_bytes gives
 * what a layer's forward / backward-data calls need (0 = no split); mc_reserve_conv_split_workspace sizes the
 * buffer of `stream` on the current device ahead of time, after which those entry points allocate nothing. */
size_t mc_conv_split_workspace_bytes(int numImages, int numImgColors, int imgSizeY, int imgSizeX,
                                     int numFilters, int filterSize, int numModulesY, int numModulesX,
                                     int numGroups);
int mc_reserve_conv_split_workspace(size_t bytes, mcStream_t stream);

/* Dry run of mc_conv_filter_acts_fused (reluBits = 0) / mc_conv_filter_acts_relu_bits (reluBits = 1): writes into `out` the
 * kernel instantiation, grid and workgroup tile the call would launch for this shape right now -- the autotuner's cached
 * choice if the shape has been tuned in this process, else the static rule's -- and launches nothing.  bench.py uses it to
 * check that the kernel it timed is the one the committed PMC counters were collected on. */
int mc_describe_conv_filter_acts(char* out, int outLen, int reluBits, int numImages, int numImgColors,
                                 int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                                 int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                                 int numGroups);

/* A partialSum the library considers a good split of the module axis for this shape on this
 * device (divides numModules; the reference takes it from the layer-def file instead). */
int mc_conv_weight_acts_pick_partial_sum(int numImages, int numImgColors, int numFilters,
                                         int filterSize, int numModules, int numGroups);

/* ---------------- dense product ----------------
 * Replaces NVMatrix::addProduct / rightMult -> cublasSgemm
 * (Kernel/src/nvmatrix/nvmatrix.cu:189-239), used by FCLayer (Kernel/src/layers/layer.cu:
 * 373-409) and the cccp plugin (PluginsSrc/src/extralayers.cu:66-174).
 * Row-major: C(m,n) = scaleTargets*C + scaleOutput * op(A) * op(B); transA: A is stored (k,m);
 * transB: B is stored (n,k).  lda/ldb/ldc are row strides in floats.
 * rowBias (nullable) adds rowBias[i] to row i and `neuron` is applied, after the scaling.
 * workspace (nullable) enables split-K for long k; mc_gemm_workspace_bytes gives the size
 * the call would like (0 = none needed). */
size_t mc_gemm_workspace_bytes(int m, int n, int k);
int mc_gemm(const float* A, const float* B, float* C, int m, int n, int k, int transA, int transB,
            int lda, int ldb, int ldc, float scaleTargets, float scaleOutput, const float* rowBias,
            int neuron, void* workspace, size_t workspaceBytes, mcStream_t stream);
/* The weight-gradient orientation, C(m,n) = scaleTargets*C + scaleOutput * A(m,k) * B(n,k)^T, that also delivers
 * rowSums[i] = sumScale * sum_k A(i,k) (overwritten) from the same pass over A: FCLayer / CCCPLayer bpropBiases
 * (layer.cu:397-400 `addSum(v, 1, ...)`, extralayers.cu:130-138) fused into bpropWeights (layer.cu:402-409,
 * extralayers.cu:140-174), where A is the layer's output gradient.  workspace: mc_gemm_row_sums_workspace_bytes. */
size_t mc_gemm_row_sums_workspace_bytes(int m, int n, int k);
int mc_gemm_row_sums(const float* A, const float* B, float* C, int m, int n, int k, int lda, int ldb, int ldc,
                     float scaleTargets, float scaleOutput, float* rowSums, float sumScale, void* workspace,
                     size_t workspaceBytes, mcStream_t stream);
/* The same product with the gradient mask of a relu fused into the store: every written element whose
 * reluActs(i, j) (row stride ldr) is not > 0 becomes 0.  Replaces the pair
 * FCLayer/CCCPLayer::bpropActs (layer.cu:380-384, extralayers.cu:103-124) + ReluNeuron's gradient
 * `actGrad * (act > 0)` (Kernel/include/layers/neuron.cuh:173-178) of the relu layer that feeds this
 * layer, when that relu has no other gradient producer: one pass over the gradient instead of three. */
int mc_gemm_relu_grad(const float* A, const float* B, float* C, int m, int n, int k, int transA,
                      int transB, int lda, int ldb, int ldc, float scaleTargets, float scaleOutput,
                      const float* reluActs, int ldr, void* workspace, size_t workspaceBytes,
                      mcStream_t stream);

/* ---------------- relu bit masks ----------------
 * The gradient of a relu needs only (output > 0).  A forward product with a fused relu can write that as ONE BIT per
 * element next to its output, and the backward product of the layer above reads the bits instead of the float
 * activation (NIN cccp1 backward: 3 MB instead of 100 MB).  No counterpart in the reference, which keeps the
 * activation for `actGrad * (act > 0)` (Kernel/include/layers/neuron.cuh:173-178).
 * Layout (chosen so that the MFMA accumulator layout writes and reads it with no cross-lane traffic): for a
 * (rows, cols) matrix with row stride ld, one 32-bit word per (32-row block, column): word (i >> 5) * ld + j; inside
 * the word, bit p = 16 * h + r (h = 0..1, r = 0..15) stands for row (i & ~31) + (r & 3) + 8 * (r >> 2) + 4 * h.
 * mc_relu_bits_words(rows, ld) words must be allocated.  Bits of rows >= rows are unspecified.
 *   mc_conv_filter_acts_relu_bits  mc_conv_filter_acts_fused with neuron = relu, one group, scaleTargets = 0,
 *                                  scaleOutput = 1, that also writes the bits of its output (ld = modules * numImages)
 *   mc_gemm_relu_bits              mc_gemm (transA = transB = 0, scaleTargets = 0, scaleOutput = 1, relu) likewise (ld = ldc)
 *   mc_gemm_relu_grad_bits         mc_gemm_relu_grad (transA = 1, transB = 0, scaleTargets = 0) with the mask given as bits (ld = ldc)
 *   mc_relu_bits_from_acts         the bits of an activation matrix that is already in memory */
size_t mc_relu_bits_words(int rows, int ld);
int mc_conv_filter_acts_relu_bits(const float* images, const float* filters, float* targets,
                                  const float* biases, unsigned* reluBits, int numImages, int numImgColors,
                                  int imgSizeY, int imgSizeX, int numFilters, int filterSize,
                                  int paddingStart, int moduleStride, int numModulesY, int numModulesX,
                                  mcStream_t stream);
int mc_gemm_relu_bits(const float* A, const float* B, float* C, int m, int n, int k, int transA,
                      int transB, int lda, int ldb, int ldc, const float* rowBias, unsigned* reluBits,
                      mcStream_t stream);
int mc_gemm_relu_grad_bits(const float* A, const float* B, float* C, int m, int n, int k, int transA,
                           int transB, int lda, int ldb, int ldc, float scaleOutput,
                           const unsigned* reluBits, mcStream_t stream);
int mc_relu_bits_from_acts(const float* acts, unsigned* reluBits, int rows, int cols, int ld,
                           mcStream_t stream);

/* ---------------- pooling ----------------
 * convLocalPool<MaxPooler|AvgPooler> (Kernel/include/conv/conv_util.cuh:335-524),
 * convLocalMaxUndo (Kernel/src/conv/conv_util.cu:1879-1968),
 * convLocalAvgUndo (conv_util.cu:1978-2072). */
#define MC_POOL_MAX 0
#define MC_POOL_AVG 1
int mc_conv_local_pool(const float* images, float* target, int numImages, int numFilters,
                       int imgSize, int subsX, int startX, int strideX, int outputsX, int pooler,
                       mcStream_t stream);
int mc_conv_local_max_undo(const float* images, const float* maxGrads, const float* maxActs,
                           float* target, int numImages, int numFilters, int imgSize, int subsX,
                           int startX, int strideX, int outputsX, float scaleTargets,
                           float scaleOutput, mcStream_t stream);
/* convLocalMaxUndo when `images` is the output of a relu layer with no other gradient producer: the relu's
 * gradient mask (neuron.cuh:173-178: grad * (act > 0)) is applied to this call's contribution before it is
 * stored, so the relu layer's own backward pass over the tensor is not needed. */
int mc_conv_local_max_undo_relu_grad(const float* images, const float* maxGrads, const float* maxActs,
                                     float* target, int numImages, int numFilters, int imgSize,
                                     int subsX, int startX, int strideX, int outputsX,
                                     float scaleTargets, float scaleOutput, mcStream_t stream);
/* convLocalPool (Kernel/include/conv/conv_util.cuh:335-524) followed by the relu neuron that `neuron=relu` puts on a pool layer
 * (NeuronLayer::fpropActs, Kernel/src/layers/layer.cu:260-262; ReluNeuron, Kernel/include/layers/neuron.cuh:152-181), applied in
 * place or into its own matrix: target = pool(images), reluTarget = max(0, target), one pass (reluTarget == target: in place; max
 * pools keep the pooled tensor for their backward pass, so the reference's parser gives the neuron a matrix of its own).
 * Bit-identical to mc_conv_local_pool + mc_neuron_activate(MC_N_RELU) on the result. */
int mc_conv_local_pool_relu(const float* images, float* target, float* reluTarget, int numImages, int numFilters, int imgSize, int subsX,
                            int startX, int strideX, int outputsX, int pooler, mcStream_t stream);
/* convLocalMaxUndo (Kernel/src/conv/conv_util.cu:1502-1572, 1868-1900) for a pool whose output carries that in-place relu: maxActs is
 * the pooled tensor (relu'd or not: the two agree on which entries are positive), maxGrads the gradient with respect to it BEFORE the relu's own backward step (NeuronLayer::bpropActs,
 * layer.cu:256-258; ReluGradientOperator, neuron.cuh:173-181), which is applied here
 * to every window's gradient (maxActs > 0 ? maxGrads : 0).  reluGrad != 0 also applies the relu gradient of the layer below, as
 * mc_conv_local_max_undo_relu_grad does.  Bit-identical to mc_neuron_input_grad(MC_N_RELU) + the plain undo. */
int mc_conv_local_max_undo_relu_out(const float* images, const float* maxGrads, const float* maxActs, float* target, int numImages,
                                    int numFilters, int imgSize, int subsX, int startX, int strideX, int outputsX, float scaleTargets,
                                    float scaleOutput, int reluGrad, mcStream_t stream);

/* convLocalPool followed by the plugin's dropout neuron (PluginsSrc/include/extralayers.cuh:39-77, train mode: u < prob ? 0 : x,
 * no rescale) in ONE pass over the pooled tensor: `target` gets the pooled values (convLocalMaxUndo routes by them,
 * conv_util.cu:1567), `dropTarget` (another buffer) what mc_dropout_counter(target, dropTarget, prob, 0, n, seed, offset) would
 * write, bit for bit, and `keep` one byte per four elements of the (features, numImages) tensor in flat order -- bit e of byte
 * i / 4 = element 4 (i / 4) + e survived -- mc_dropout_keep_bytes(n) bytes.  NIN's pool -> dropout pairs (NIN/cifar-10_def). */
size_t mc_dropout_keep_bytes(size_t n);
int mc_conv_local_pool_dropout(const float* images, float* target, float* dropTarget, unsigned char* keep,
                               int numImages, int numFilters, int imgSize, int subsX, int startX,
                               int strideX, int outputsX, int pooler, float prob, unsigned long long seed,
                               unsigned long long offset, mcStream_t stream);
/* convLocalMaxUndo whose incoming gradient still has to pass the dropout neuron's backward step (extralayers.cuh:60-77,
 * grad where the element survived, else 0): the keep bytes of mc_conv_local_pool_dropout are applied to maxGrads as they are
 * read.  Equals mc_dropout_counter on maxGrads followed by mc_conv_local_max_undo (reluGrad = 0) or
 * mc_conv_local_max_undo_relu_grad (reluGrad = 1), bit for bit. */
int mc_conv_local_max_undo_dropout(const float* images, const float* maxGrads, const float* maxActs,
                                   const unsigned char* keep, float* target, int numImages, int numFilters,
                                   int imgSize, int subsX, int startX, int strideX, int outputsX,
                                   float scaleTargets, float scaleOutput, int reluGrad, mcStream_t stream);
int mc_conv_local_avg_undo(const float* avgGrads, float* target, int numImages, int numFilters,
                           int imgSize, int subsX, int startX, int strideX, int outputsX,
                           float scaleTargets, float scaleOutput, mcStream_t stream);
/* mc_conv_local_avg_undo followed by the relu gradient of the layer below (NeuronLayer::bpropActs with
 * ReluNeuron, neuron.cuh:173-178) in one pass: target = reluActs > 0 ? scaleOutput * undo : 0.  reluActs are the
 * pooled images (the relu's output), same shape as target; no accumulate form. */
int mc_conv_local_avg_undo_relu_grad(const float* avgGrads, const float* reluActs, float* target,
                                     int numImages, int numFilters, int imgSize, int subsX, int startX,
                                     int strideX, int outputsX, float scaleOutput, mcStream_t stream);

/* ---------------- response normalisation ----------------
 * convResponseNorm (conv_util.cu:2074-2076), convContrastNorm (:2084-2222),
 * convResponseNormUndo / convContrastNormUndo (:2224-2424),
 * convResponseNormCrossMap (:2753-2809, 2902-2904), convResponseNormCrossMapUndo (:2811-2900).
 * The undo calls OVERWRITE `acts`, as the reference does (conv_util.cu:2236). */
int mc_conv_response_norm(const float* images, float* denoms, float* target, int numImages,
                          int numFilters, int imgSize, int sizeX, float addScale, float powScale,
                          mcStream_t stream);
int mc_conv_contrast_norm(const float* images, const float* meanDiffs, float* denoms, float* target,
                          int numImages, int numFilters, int imgSize, int sizeX, float addScale,
                          float powScale, mcStream_t stream);
int mc_conv_response_norm_undo(const float* outGrads, const float* denoms, const float* inputs,
                               float* acts, float* target, int numImages, int numFilters,
                               int imgSize, int sizeX, float addScale, float powScale,
                               float scaleTargets, float scaleOutput, mcStream_t stream);
int mc_conv_response_norm_cross_map(const float* images, float* denoms, float* target,
                                    int numImages, int numFilters, int imgSize, int sizeF,
                                    float addScale, float powScale, int blocked, float k,
                                    mcStream_t stream);
int mc_conv_response_norm_cross_map_undo(const float* outGrads, const float* denoms,
                                         const float* inputs, float* acts, float* target,
                                         int numImages, int numFilters, int imgSize, int sizeF,
                                         float addScale, float powScale, int blocked,
                                         float scaleTargets, float scaleOutput, mcStream_t stream);

/* ---------------- image layers (nailbed, blur, resize, rgb2yuv, rgb2lab) ----------------
 * Tensors are (channels, imgSize*imgSize, numImages), image index innermost, dense.
 * convBedOfNails / convBedOfNailsUndo (conv_util.cu:667-676, kernel :463-537): output (oy, ox) is pixel
 *   (startX + oy*strideX, startX + ox*strideX), outputsX = ceil(imgSize / strideX), strideX > 1.  The undo with
 *   scaleTargets == 0 zeroes every non-nail pixel; with scaleTargets != 0 it touches the nail pixels only.
 * convGaussianBlur (conv_util.cu:684-728, kernel :358-443): one direction of a separable blur,
 *   out[x] = sum_d in[x+d]*filter[radius+d], zero outside the line; `filter` is 2*radius+1 floats on the
 *   device, radius 1..4, imgSize >= 2*radius+1; target may be the same buffer as images.
 * convResizeBilinear (conv_util.cu:2426-2474, kernel :290-341): source coordinate
 *   clamp(px*scale + imgSize/2 - tgtSize/2*scale, 0, imgSize - 1.01).
 * convRGBToYUV (conv_util.cu:2480-2524, kernel :172-194), convRGBToLAB (:2526-2601, kernel :237-268;
 *   `center` subtracts 50 from L*): 3-channel inputs. */
int mc_conv_bed_of_nails(const float* images, float* target, int numImages, int numChannels,
                         int imgSize, int startX, int strideX, float scaleTargets, float scaleOutput,
                         mcStream_t stream);
int mc_conv_bed_of_nails_undo(const float* actsGrad, float* target, int numImages, int numChannels,
                              int imgSize, int startX, int strideX, float scaleTargets,
                              float scaleOutput, mcStream_t stream);
int mc_conv_gaussian_blur(const float* images, const float* filter, float* target, int numImages,
                          int numChannels, int imgSize, int radius, int horiz, float scaleTargets,
                          float scaleOutput, mcStream_t stream);
int mc_conv_resize_bilinear(const float* images, float* target, int numImages, int numChannels,
                            int imgSize, int tgtSize, float scale, mcStream_t stream);
int mc_conv_rgb_to_yuv(const float* images, float* target, int numImages, int imgPixels,
                       mcStream_t stream);
int mc_conv_rgb_to_lab(const float* images, float* target, int numImages, int imgPixels, int center,
                       mcStream_t stream);

/* ---------------- element-wise max layer ----------------
 * EltwiseMaxLayer::fpropActs (Kernel/src/layers/layer.cu:670-676: applyBinary(NVMatrixAggs::Max)) and
 * computeEltwiseMaxGrad (Kernel/src/layers/layer_kernels.cu:155-186):
 * target = (add ? target : 0) + actGrad * (output == input). */
int mc_eltwise_max(const float* a, const float* b, float* out, size_t n, mcStream_t stream);
int mc_compute_eltwise_max_grad(const float* actGrad, const float* input, const float* output,
                                float* target, int add, size_t n, mcStream_t stream);

/* ---------------- random sparse connectivity (randSparse conv / local layers) ----------------
 * convFilterActsSparse / convImgActsSparse / convWeightActsSparse and the local*Sparse twins
 * (Kernel/include/conv/cudaconv2.cuh:52-78; filter_acts.cu:2226-2245, img_acts.cu:2058-2079,
 * weight_acts.cu:1110-1127) connect filter group g to the channels dColorIndices[g*numFilterColors + c].
 * Here that is three steps, each through this header:
 *   forward / weight gradient: mc_conv_gather_channels(images, dColorIndices, gathered, ...) and then the dense
 *     grouped entry point on `gathered` with numImgColors = numGroups*numFilterColors;
 *   backward data: the dense grouped mc_conv_img_acts into a (numGroups*numFilterColors)-channel temporary, then
 *     mc_conv_scatter_channels sums, for every real channel c, the `overSample` gathered channels
 *     inverseIndices[c*overSample + o] that were copies of it (the reference leaves that sum to the layer,
 *     layer.cu:531-535): target = scaleTargets*target + scaleOutput*sum.
 * Index arrays are int32 on the device; an index outside the source tensor contributes zeros. */
int mc_conv_gather_channels(const float* images, const int* channelIndices, float* target,
                            int numImages, int numImgColors, int numIndices, int imgPixels,
                            mcStream_t stream);
int mc_conv_scatter_channels(const float* gathered, const int* inverseIndices, float* target,
                             int numImages, int numImgColors, int overSample, int imgPixels,
                             float scaleTargets, float scaleOutput, mcStream_t stream);

/* ---------------- group sparsity over same-label cases (plugin cost.gsinlabel) ----------------
 * CalculateSqrtSumSquareMatrix / CalculateGradient (PluginsSrc/src/extrautils.cu:128-158, 201-211).
 * acts is (channels*imagePixels, numCases), labels (1, numCases).  For every channel and every distinct label
 * value: S = sqrt(count * sum over the label's cases and all pixels of acts^2); the cost is the sum of all S.
 * The cost call fills sqrts (channels, numCases: S of each case's own label), counts (numCases: size of the
 * case's label group), costTerms (channels, numCases: S at the first case of each group, 0 elsewhere -- their
 * sum is the cost) and uses pixelSumSq (channels, numCases) as scratch.
 * The grad call: target = scaleTargets*target + scaleOutput * count*acts / (S + 2.5e-7). */
int mc_group_sparsity_in_label_cost(const float* labels, const float* acts, float* pixelSumSq,
                                    float* sqrts, float* counts, float* costTerms, int numCases,
                                    int channels, int imagePixels, mcStream_t stream);
int mc_group_sparsity_in_label_grad(const float* acts, const float* sqrts, const float* counts,
                                    float* target, int numCases, int channels, int imagePixels,
                                    float scaleTargets, float scaleOutput, mcStream_t stream);

/* ---------------- neurons ----------------
 * Neuron functors of Kernel/include/layers/neuron.cuh:120-462 and the dropout neuron of
 * PluginsSrc/include/extralayers.cuh:34-102.  `a`,`b` are the neuron parameters
 * (brelu[a], tanh[a,b], linear[a,b], dropout: a = prob).
 * mc_neuron_input_grad: target = (add ? target : 0) + f'(...)*actsGrad, where `actsOrInputs`
 * is the neuron's output for relu/brelu/logistic/tanh/softrelu/sqrt and its input for
 * abs/square (neuron.cuh:255-260, 384-389); ignored for ident/linear. */
enum {
    MC_N_IDENT = 0, MC_N_RELU, MC_N_BRELU, MC_N_ABS, MC_N_LOGISTIC, MC_N_TANH, MC_N_SOFTRELU,
    MC_N_SQUARE, MC_N_SQRT, MC_N_LINEAR
};
int mc_neuron_activate(int type, float a, float b, const float* inputs, float* outputs, size_t n,
                       mcStream_t stream);
int mc_neuron_input_grad(int type, float a, float b, const float* actsGrad,
                         const float* actsOrInputs, float* target, int add, size_t n,
                         mcStream_t stream);
/* dropout (train): out = mask < prob ? 0 : in; the same rule routes the gradient
 * (extralayers.cuh:81-96).  `mask` holds uniforms in [0,1).  mc_rand_uniform fills it with a
 * counter-based (Philox4x32-10) stream: element i depends on (seed, offset + i) only. */
int mc_dropout_apply(const float* in, const float* mask, float* out, float prob, int add, size_t n,
                     mcStream_t stream);
int mc_rand_uniform(float* out, size_t n, unsigned long long seed, unsigned long long offset,
                    mcStream_t stream);
/* The two calls above in one, with the uniforms never stored: out = u_i < prob ? 0 : in_i (+ out when `add`) where
 * u_i is element (offset + i) of mc_rand_uniform's stream for `seed` -- bit-identical to mc_rand_uniform +
 * mc_dropout_apply.  The backward pass passes the same (seed, offset) and regenerates the mask instead of
 * reading it (DropoutNeuron keeps a mask matrix in the reference, extralayers.cuh:60-77). */
int mc_dropout_counter(const float* in, float* out, float prob, int add, size_t n,
                       unsigned long long seed, unsigned long long offset, mcStream_t stream);

/* ---------------- generic NVMatrix-style ops ----------------
 * Replace the NVMatrix element-wise / vector / aggregation kernels on the path
 * (Kernel/include/nvmatrix/nvmatrix_kernels.cuh): addVector (bias add: layer.cu:376,483),
 * addSum (bias gradient layer.cu:392,496; partial weight-gradient reduction layer.cu:520),
 * add/scale (eltsum layer, Weights::update). */
/* acts(rows, cols) row-major: acts[r][c] += scaleVec * vec[r]  (bias per feature row) */
int mc_add_row_vector(float* acts, const float* vec, int rows, int cols, float scaleVec,
                      mcStream_t stream);
/* out[r] = scaleTargets*out[r] + scaleOutput * sum_c v[r][c] */
int mc_sum_rows(const float* v, float* out, int rows, int cols, float scaleTargets,
                float scaleOutput, mcStream_t stream);
/* out[c] = scaleTargets*out[c] + scaleOutput * sum_r v[r][c]  (fixed summation order) */
int mc_sum_cols(const float* v, float* out, int rows, int cols, float scaleTargets,
                float scaleOutput, mcStream_t stream);
/* y = scaleY*y + scaleX*x */
int mc_axpby(float* y, const float* x, float scaleY, float scaleX, size_t n, mcStream_t stream);
/* y = scaleX * x (y may alias x) */
int mc_scale(float* y, const float* x, float scaleX, size_t n, mcStream_t stream);
/* out[0] = sum of v[0..n) accumulated in double, fixed order (cost read-back). */
int mc_sum_all(const float* v, size_t n, double* out, mcStream_t stream);
/* dst[i] += src[i], i < n, in double: accumulates a minibatch's cost sums into a running total on the device.  No counterpart
 * in the reference (its costs go to the host every minibatch, Kernel/src/convnet/cost.cu); used by the runtime's captured
 * training step, whose kernels must write to fixed addresses. */
int mc_accumulate_f64(double* dst, const double* src, int n, mcStream_t stream);
/* The contraction entry points time their candidate tiles the first time they see a shape (events + a blocking wait).
 * mc_set_autotune(0) switches that off for the calling thread (a shape not tuned yet runs the static rule's tile: same bits),
 * mc_set_autotune(1) back on: for callers that are capturing a HIP graph or otherwise must not block.  Returns MC_OK. */
int mc_set_autotune(int enabled);
/* convFilterActs (Kernel/src/conv/filter_acts.cu:2206-2211) has a second kernel family for small-filter layers -- stride 1, one
 * group, 5x5 filters, numImages a multiple of 32, and either 3 input channels with 32 / 64 / 192 filters or 32 filters over a multiple
 * of 8 channels: the input band of a 2x2-pixel output block is staged in LDS once and all 25 taps are formed from it
 * (csrc/kernels/conv_band.hip).  It returns the same bits as the general engine, so the choice is made by one timed trial per
 * shape, like the tile choice above.  mode -1: that trial (default; MC_CONV_BAND in the environment sets the initial mode),
 * 0: general engine only, 1: the band kernel wherever it exists.  Process-wide.  No counterpart in the reference. */
int mc_set_conv_band(int mode);

/* ---------------- softmax / logistic regression cost ----------------
 * SoftmaxLayer::fpropActs (Kernel/src/layers/layer.cu:612-622, five NVMatrix ops there),
 * computeLogregCost / computeLogregGrad / computeSoftmaxGrad / computeLogregSoftmaxGrad
 * (Kernel/src/layers/layer_kernels.cu:198-294).  probs/acts are (numOut, numCases) row-major,
 * labels (1, numCases) float class indices.
 * Two kernel families, chosen by the shape alone: below 64 classes one lane walks the classes of a case (sums in class
 * order); from 64 classes on (and numCases a multiple of 4) a workgroup shares four cases and adds in a fixed tree
 * (1000 classes x 128 cases: softmax 385 -> 7.5 us, cost 100 -> 8 us).  Deterministic either way; mc_softmax_logreg
 * takes the same family as the separate calls, so its bit-equality with them holds for every shape. */
int mc_softmax(const float* input, float* target, int numCases, int numOut, mcStream_t stream);
int mc_compute_logreg_cost(const float* labels, const float* probs, float* labelLogProbs,
                           float* correctProbs, int numCases, int numOut, mcStream_t stream);
/* mc_compute_logreg_cost and the two sums LogregCostLayer::fpropActs takes of its outputs (layer.cu:993-1005,
 * labelLogProbs.sum() and correctProbs.sum(), each a reduction launch + a D->H sync there) in one launch:
 * sums[0] = sum labelLogProbs, sums[1] = sum correctProbs, in double, same order as mc_sum_all. */
int mc_compute_logreg_cost_sums(const float* labels, const float* probs, float* labelLogProbs,
                                float* correctProbs, double* sums, int numCases, int numOut,
                                mcStream_t stream);
/* SoftmaxLayer::fpropActs + LogregCostLayer::fpropActs + SoftmaxLayer::bpropActs under a logreg cost (layer.cu:617-631,
 * 993-1005; kernels layer_kernels.cu:41-74, 137-152) for one minibatch in ONE launch: probs = softmax(input),
 * labelLogProbs / correctProbs / sums as mc_compute_logreg_cost_sums, and -- when gradTarget is not NULL --
 * gradTarget = (add ? gradTarget : 0) + coeff * (1[label] - probs), the gradient with respect to `input`.  Every output is
 * bit-identical to mc_softmax, mc_compute_logreg_cost_sums and mc_compute_logreg_softmax_grad run one after the other; meant
 * for classifier tails of up to a few thousand (class, case) pairs, where three dependent launches are pure latency. */
int mc_softmax_logreg(const float* labels, const float* input, float* probs, float* labelLogProbs,
                      float* correctProbs, double* sums, float* gradTarget, int add, float coeff,
                      int numCases, int numOut, mcStream_t stream);
int mc_compute_logreg_grad(const float* labels, const float* probs, float* target, int add,
                           float coeff, int numCases, int numOut, mcStream_t stream);
int mc_compute_softmax_grad(const float* acts, const float* actsGrad, float* target, int add,
                            int numCases, int numOut, mcStream_t stream);
int mc_compute_logreg_softmax_grad(const float* labels, const float* probs, float* target, int add,
                                   float coeff, int numCases, int numOut, mcStream_t stream);

/* ---------------- SGD update ----------------
 * Weights::update (Kernel/include/layers/weights.cuh:150-164) as one pass:
 *   inc = mom*inc + grad; if (wc > 0) inc -= wc*eps*W; W += inc. */
int mc_weights_update(float* weights, float* inc, const float* grad, size_t n, float mom, float eps,
                      float wc, mcStream_t stream);

/* The same update for every tensor of a net in ONE launch: weights / inc / grad are the three
 * arenas the tensors live in (same offsets in each), `spansDevice` a device array of spans; a span
 * lies inside one tensor, starts 16-byte aligned, holds at most a few thousand elements and carries
 * that tensor's momentum and wceps = (wc > 0 ? -wc*eps : 0).  One workgroup per span.
 * No counterpart in the reference, which launches 2-3 element-wise kernels per tensor
 * (Weights::update, weights.cuh:150-164). */
typedef struct {
    unsigned long long start; /* element offset into the arenas */
    unsigned count;           /* elements in this span */
    float mom, wceps;
    unsigned pad;
} mcUpdateSpan;
int mc_weights_update_spans(float* weights, float* inc, const float* grad, const mcUpdateSpan* spansDevice,
                            int numSpans, mcStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MC_KERNELS_H */
