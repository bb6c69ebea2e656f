/*
 * oracle/golden.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the layer arithmetic of the cuda-convnet training
 * step.  It is the checker the HIP path is compared with (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg) and nothing else may
 * link, load or call it.
 *
 * Pinning: the reference cannot be compiled here (nvcc / CUDA SDK cutil /
 * cuBLAS / MKL / python2.7 are absent) and ships no stored vectors; its own
 * unit tests are MATLAB generators (Kernel/unittest/{filter,img,weight}_acts/
 * DataGen.m) whose expected values are convn() calls.  tests/test_oracle_pin.py
 * re-creates those three cases with scipy.signal and checks this file against
 * them, and runs the reference's finite-difference gradient check rule
 * (Kernel/src/convnet/convnet.cu:285-335) over the whole oracle network.
 *
 * Layout everywhere (Kernel/src/conv/filter_acts.cu:39-43): activations are
 * (channels, y, x, case) with the case index innermost; conv filters are
 * (filterChannels, filterPixels, numFilters) with the filter index innermost;
 * unshared (local) filters carry a leading module index.
 *
 * Accumulation is in double and rounded once to float at the end: the oracle is
 * the "true value" side of a tolerance comparison, not a bit pattern.
 * Max-pool / MaxUndo use float equality exactly as the reference does.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 4096 /* largest minibatch the oracle accepts */

static inline float blend(float old, double v, float scaleTargets, float scaleOutput) {
    return scaleTargets == 0.0f ? (float)(scaleOutput * v)
                                : (float)((double)scaleTargets * old + (double)scaleOutput * v);
}

/* ------------------------------------------------------------------------
 * convFilterActs / localFilterActs
 * follows Kernel/src/conv/filter_acts.cu:58-226 (colour), :259-413 (many
 * channels + groups), host wrapper :620-672, entry points :2206-2224.
 *   images  (C*imgSizeY*imgSizeX, N)
 *   filters conv: (filterColors*fs*fs, F)   local: (M*filterColors*fs*fs, F)
 *   targets (F*modulesY*modulesX, N)
 * paddingStart <= 0 (Python/layer.py:959 negates the user's padding).
 * ---------------------------------------------------------------------- */
void gold_filter_acts(const float* images, const float* filters, float* targets,
                      int N, int C, int imgSizeY, int imgSizeX, int F, int fs,
                      int paddingStart, int stride, int modulesY, int modulesX, int groups,
                      float scaleTargets, float scaleOutput, int conv) {
    const int M = modulesY * modulesX, fp = fs * fs;
    const int fc = C / groups;            /* filter_acts.cu:641 numFilterColors */
    const int fpg = F / groups;           /* filters per group */
    const long imgPix = (long)imgSizeY * imgSizeX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int f = 0; f < F; f++) {
        for (int m = 0; m < M; m++) {
            double acc[MAXN];
            for (int n = 0; n < N; n++) acc[n] = 0.0;
            const int g = f / fpg;
            const int my = m / modulesX, mx = m % modulesX;
            const int y0 = paddingStart + my * stride, x0 = paddingStart + mx * stride;
            for (int c = 0; c < fc; c++) {
                for (int fy = 0; fy < fs; fy++) {
                    const int y = y0 + fy;
                    if (y < 0 || y >= imgSizeY) continue;   /* zero padding, filter_acts.cu:154-179 */
                    for (int fx = 0; fx < fs; fx++) {
                        const int x = x0 + fx;
                        if (x < 0 || x >= imgSizeX) continue;
                        const long k = (long)c * fp + fy * fs + fx;
                        const double w = filters[((conv ? 0 : (long)m * fc * fp) + k) * F + f];
                        const float* src = images + (((long)(g * fc + c)) * imgPix + (long)y * imgSizeX + x) * N;
                        for (int n = 0; n < N; n++) acc[n] += w * (double)src[n];
                    }
                }
            }
            float* dst = targets + ((long)f * M + m) * N;
            for (int n = 0; n < N; n++) dst[n] = blend(dst[n], acc[n], scaleTargets, scaleOutput);
        }
    }
}

/* ------------------------------------------------------------------------
 * convImgActs / localImgActs  (backward w.r.t. the layer input)
 * follows Kernel/src/conv/img_acts.cu:473-658 (many colours), :63-236 (<=3
 * colours), host wrapper :1043-1104, entry points :2040-2056.
 *   hidActs (F*M, N), filters as above, targets (C*imgSizeY*imgSizeX, N)
 * ---------------------------------------------------------------------- */
void gold_img_acts(const float* hidActs, const float* filters, float* targets,
                   int N, int C, int imgSizeY, int imgSizeX, int F, int fs,
                   int paddingStart, int stride, int modulesY, int modulesX, int groups,
                   float scaleTargets, float scaleOutput, int conv) {
    const int M = modulesY * modulesX, fp = fs * fs;
    const int fc = C / groups, fpg = F / groups;
    const long imgPix = (long)imgSizeY * imgSizeX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSizeY * imgSizeX; px++) {
            double acc[MAXN];
            for (int n = 0; n < N; n++) acc[n] = 0.0;
            const int y = px / imgSizeX, x = px % imgSizeX;
            const int g = c / fc, cl = c % fc;
            /* modules whose receptive field covers (y,x): img_acts.cu:540-555 */
            for (int my = 0; my < modulesY; my++) {
                const int fy = y - (paddingStart + my * stride);
                if (fy < 0 || fy >= fs) continue;
                for (int mx = 0; mx < modulesX; mx++) {
                    const int fx = x - (paddingStart + mx * stride);
                    if (fx < 0 || fx >= fs) continue;
                    const int m = my * modulesX + mx;
                    const long k = (long)cl * fp + fy * fs + fx;
                    const float* wrow = filters + ((conv ? 0 : (long)m * fc * fp) + k) * F;
                    for (int f = g * fpg; f < (g + 1) * fpg; f++) {
                        const double w = wrow[f];
                        const float* src = hidActs + ((long)f * M + m) * N;
                        for (int n = 0; n < N; n++) acc[n] += w * (double)src[n];
                    }
                }
            }
            float* dst = targets + ((long)c * imgPix + px) * N;
            for (int n = 0; n < N; n++) dst[n] = blend(dst[n], acc[n], scaleTargets, scaleOutput);
        }
    }
}

/* ------------------------------------------------------------------------
 * convWeightActs / localWeightActs  (backward w.r.t. the filters)
 * follows Kernel/src/conv/weight_acts.cu:58-220 (colour), :256-409 (many
 * channels), host wrapper :608-681, entry points :1092-1108.
 *   images (C*imgPix, N), hidActs (F*M, N)
 *   targets conv:  ((M/partialSum)*filterColors*fs*fs, F), chunk q sums
 *                  modules [q*partialSum, (q+1)*partialSum)
 *           local: (M*filterColors*fs*fs, F)  (partialSum forced to 1,
 *                  weight_acts.cu:1103-1108)
 * partialSum == 0 means "all modules" (weight_acts.cu:627).
 * ---------------------------------------------------------------------- */
void gold_weight_acts(const float* images, const float* hidActs, float* targets,
                      int N, int C, int imgSizeY, int imgSizeX, int F, int fs,
                      int paddingStart, int stride, int modulesY, int modulesX, int groups,
                      int partialSum, float scaleTargets, float scaleOutput, int conv) {
    const int M = modulesY * modulesX, fp = fs * fs;
    const int fc = C / groups, fpg = F / groups;
    const long imgPix = (long)imgSizeY * imgSizeX;
    if (!conv) partialSum = 1;
    if (partialSum <= 0) partialSum = M;
    const int chunks = M / partialSum;
#pragma omp parallel for collapse(2) schedule(static)
    for (int q = 0; q < chunks; q++) {
        for (int k = 0; k < fc * fp; k++) {
            const int cl = k / fp, p = k % fp, fy = p / fs, fx = p % fs;
            for (int f = 0; f < F; f++) {
                const int g = f / fpg;
                double acc = 0.0;
                for (int m = q * partialSum; m < (q + 1) * partialSum; m++) {
                    const int my = m / modulesX, mx = m % modulesX;
                    const int y = paddingStart + my * stride + fy, x = paddingStart + mx * stride + fx;
                    if (y < 0 || y >= imgSizeY || x < 0 || x >= imgSizeX) continue;
                    const float* a = images + (((long)(g * fc + cl)) * imgPix + (long)y * imgSizeX + x) * N;
                    const float* h = hidActs + ((long)f * M + m) * N;
                    double s = 0.0;
                    for (int n = 0; n < N; n++) s += (double)a[n] * (double)h[n];
                    acc += s;
                }
                float* dst = targets + ((long)q * fc * fp + k) * F + f;
                *dst = blend(*dst, acc, scaleTargets, scaleOutput);
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convLocalPool<MaxPooler|AvgPooler>
 * follows Kernel/include/conv/conv_util.cuh:71-194 (poolers + kLocalPool).
 *   imgs (C*imgSize^2, N) -> target (C*outputsX^2, N)
 * window [start + o*stride, +sizeX) clipped to the image; avg divides by the
 * clipped region size (:169,190); max starts from -2e38 (:89-91).
 * float arithmetic in the reference's order (y outer, x inner) so that the
 * result is the same bit pattern a straightforward fp32 kernel produces.
 * ---------------------------------------------------------------------- */
void gold_pool(const float* imgs, float* target, int N, int C, int imgSize,
               int sizeX, int start, int stride, int outputsX, int isMax) {
    const long imgPix = (long)imgSize * imgSize;
    const int numOut = outputsX * outputsX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int o = 0; o < numOut; o++) {
            const int oy = o / outputsX, ox = o % outputsX;
            const int sy = start + oy * stride, sx = start + ox * stride;
            const int y0 = sy < 0 ? 0 : sy, x0 = sx < 0 ? 0 : sx;
            const int y1 = sy + sizeX < imgSize ? sy + sizeX : imgSize;
            const int x1 = sx + sizeX < imgSize ? sx + sizeX : imgSize;
            const int region = (y1 - y0) * (x1 - x0);
            float* dst = target + ((long)c * numOut + o) * N;
            for (int n = 0; n < N; n++) {
                float a = isMax ? -2e38f : 0.0f;
                for (int y = y0; y < y1; y++)
                    for (int x = x0; x < x1; x++) {
                        const float v = imgs[((long)c * imgPix + (long)y * imgSize + x) * N + n];
                        a = isMax ? fmaxf(a, v) : a + v;
                    }
                dst[n] = isMax ? a : a / region;
            }
        }
    }
}

/* outputs whose window contains pixel coordinate p along one axis
 * (Kernel/src/conv/conv_util.cu:1517-1520, C integer division) */
static inline void out_range(int p, int start, int sizeX, int stride, int outputsX, int* lo, int* hi) {
    *lo = (p - start < sizeX) ? 0 : 1 + (p - start - sizeX) / stride;
    const int e = 1 + (p - start) / stride;
    *hi = e < outputsX ? e : outputsX;
}

/* ------------------------------------------------------------------------
 * convLocalMaxUndo  -- Kernel/src/conv/conv_util.cu:1502-1597, host :1879-1968
 * target[px] = sT*target + sO * sum_{windows containing px} (img[px]==maxAct[w]) * maxGrad[w]
 * Equality routing: ties receive the gradient more than once (:1567).
 * ---------------------------------------------------------------------- */
void gold_max_undo(const float* imgs, const float* maxGrads, const float* maxActs, float* target,
                   int N, int C, int imgSize, int sizeX, int start, int stride, int outputsX,
                   float scaleTargets, float scaleOutput) {
    const long imgPix = (long)imgSize * imgSize;
    const int numOut = outputsX * outputsX;
    const int lim = start + stride * (outputsX - 1) + sizeX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int y = px / imgSize, x = px % imgSize;
            int ylo, yhi, xlo, xhi;
            out_range(y, start, sizeX, stride, outputsX, &ylo, &yhi);
            out_range(x, start, sizeX, stride, outputsX, &xlo, &xhi);
            const int inside = x >= start && x < lim && y >= start && y < lim;  /* :1540-1541 */
            for (int n = 0; n < N; n++) {
                float prod = 0.0f;
                if (inside) {
                    const float img = imgs[((long)c * imgPix + px) * N + n];
                    for (int my = ylo; my < yhi; my++)
                        for (int mx = xlo; mx < xhi; mx++) {
                            const long o = ((long)c * numOut + my * outputsX + mx) * N + n;
                            prod += (img == maxActs[o]) * maxGrads[o];
                        }
                }
                float* d = target + ((long)c * imgPix + px) * N + n;
                *d = scaleTargets == 0.0f ? scaleOutput * prod : scaleTargets * *d + scaleOutput * prod;
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convLocalAvgUndo -- Kernel/src/conv/conv_util.cu:1395-1480, host :1978-2072
 * target[px] = sT*target + sO * sum_{windows containing px} avgGrad[w] / clippedRegionSize(w)
 * ---------------------------------------------------------------------- */
void gold_avg_undo(const float* avgGrads, float* target, int N, int C, int imgSize,
                   int sizeX, int start, int stride, int outputsX,
                   float scaleTargets, float scaleOutput) {
    const long imgPix = (long)imgSize * imgSize;
    const int numOut = outputsX * outputsX;
    const int lim = start + stride * (outputsX - 1) + sizeX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int y = px / imgSize, x = px % imgSize;
            int ylo, yhi, xlo, xhi;
            out_range(y, start, sizeX, stride, outputsX, &ylo, &yhi);
            out_range(x, start, sizeX, stride, outputsX, &xlo, &xhi);
            const int inside = x >= start && x < lim && y >= start && y < lim;
            for (int n = 0; n < N; n++) {
                float prod = 0.0f;
                if (inside) {
                    for (int my = ylo; my < yhi; my++) {
                        const float ry0 = fmaxf(0, start + my * stride);
                        const float ry1 = fminf(imgSize, start + my * stride + sizeX);
                        for (int mx = xlo; mx < xhi; mx++) {
                            const float rx0 = fmaxf(0, start + mx * stride);
                            const float rx1 = fminf(imgSize, start + mx * stride + sizeX);
                            const float inv = 1.0f / ((rx1 - rx0) * (ry1 - ry0));
                            prod += avgGrads[((long)c * numOut + my * outputsX + mx) * N + n] * inv;
                        }
                    }
                }
                float* d = target + ((long)c * imgPix + px) * N + n;
                *d = scaleTargets == 0.0f ? scaleOutput * prod : scaleTargets * *d + scaleOutput * prod;
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convResponseNorm / convContrastNorm (within-map)
 * follows Kernel/src/conv/conv_util.cu:831-895 (kCNorm_manyfilter), hosts
 * :2074-2222.  denom = 1 + addScale * sum_{size x size window} meanDiffs^2,
 * window [p - size/2, p - size/2 + size) clipped; target = imgs * denom^-pow.
 * convResponseNorm passes images as meanDiffs (:2074-2076).
 * ---------------------------------------------------------------------- */
void gold_cnorm(const float* imgs, const float* meanDiffs, float* denoms, float* target,
                int N, int C, int imgSize, int size, float addScale, float powScale) {
    const long imgPix = (long)imgSize * imgSize;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int y = px / imgSize, x = px % imgSize;
            const int sy = -size / 2 + y, sx = -size / 2 + x;
            const int y0 = sy < 0 ? 0 : sy, x0 = sx < 0 ? 0 : sx;
            const int y1 = sy + size < imgSize ? sy + size : imgSize;
            const int x1 = sx + size < imgSize ? sx + size : imgSize;
            for (int n = 0; n < N; n++) {
                double s = 0.0;
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) {
                        const double v = meanDiffs[((long)c * imgPix + (long)yy * imgSize + xx) * N + n];
                        s += v * v;
                    }
                const float d = (float)(1.0 + (double)addScale * s);
                const long i = ((long)c * imgPix + px) * N + n;
                denoms[i] = d;
                target[i] = (float)((double)imgs[i] * pow((double)d, -(double)powScale));
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convResponseNormUndo / convContrastNormUndo (within-map)
 * follows Kernel/src/conv/conv_util.cu:1602-1615 (prelims: acts is
 * OVERWRITTEN with -2*addScale*pow*outGrads*acts/denoms), :1639-1722 (main),
 * host :2238-2424.  Mirrored window [p + size/2 - size + 1, p + size/2 + 1).
 * ---------------------------------------------------------------------- */
void gold_rnorm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                     float* target, int N, int C, int imgSize, int size, float addScale,
                     float powScale, float scaleTargets, float scaleOutput) {
    const long imgPix = (long)imgSize * imgSize;
    const long total = (long)C * imgPix * N;
    const float sc = -2 * addScale * powScale;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < total; i++) acts[i] = sc * outGrads[i] * acts[i] / denoms[i];
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int y = px / imgSize, x = px % imgSize;
            int y0 = y + size / 2 - size + 1, x0 = x + size / 2 - size + 1;
            int y1 = y + size / 2 + 1, x1 = x + size / 2 + 1;
            if (y0 < 0) y0 = 0;
            if (x0 < 0) x0 = 0;
            if (y1 > imgSize) y1 = imgSize;
            if (x1 > imgSize) x1 = imgSize;
            for (int n = 0; n < N; n++) {
                double s = 0.0;
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++)
                        s += acts[((long)c * imgPix + (long)yy * imgSize + xx) * N + n];
                const long i = ((long)c * imgPix + px) * N + n;
                const double v = (double)inputs[i] * s + (double)outGrads[i] * pow((double)denoms[i], -(double)powScale);
                target[i] = blend(target[i], v, scaleTargets, scaleOutput);
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convResponseNormCrossMap -- Kernel/src/conv/conv_util.cu:1041-1091 (kFCNorm),
 * host :2753-2809, entry :2902-2904 (images passed as meanDiffs).
 * Window over channels [f - size/2, f - size/2 + size) clipped, or the
 * enclosing block of `size` channels when blocked; denom = k + addScale*sum x^2.
 * ---------------------------------------------------------------------- */
void gold_cmrnorm(const float* imgs, float* denoms, float* target, int N, int C, int imgSize,
                  int size, float addScale, float powScale, int blocked, float k) {
    const long imgPix = (long)imgSize * imgSize;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int startF = blocked ? (c / size) * size : -size / 2 + c;
            const int f0 = blocked ? startF : (startF < 0 ? 0 : startF);
            const int f1 = startF + size < C ? startF + size : C;
            for (int n = 0; n < N; n++) {
                double s = 0.0;
                for (int f = f0; f < f1; f++) {
                    const double v = imgs[((long)f * imgPix + px) * N + n];
                    s += v * v;
                }
                const float d = (float)((double)k + (double)addScale * s);
                const long i = ((long)c * imgPix + px) * N + n;
                denoms[i] = d;
                target[i] = (float)((double)imgs[i] * pow((double)d, -(double)powScale));
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * convResponseNormCrossMapUndo -- Kernel/src/conv/conv_util.cu:1115-1181
 * (kFRNormUndo), host :2811-2900.  acts is overwritten by the same prelims
 * pass; undo window start f - size + size/2 + 1 (:1144).
 * ---------------------------------------------------------------------- */
void gold_cmrnorm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                       float* target, int N, int C, int imgSize, int size, float addScale,
                       float powScale, int blocked, float scaleTargets, float scaleOutput) {
    const long imgPix = (long)imgSize * imgSize;
    const long total = (long)C * imgPix * N;
    const float sc = -2 * addScale * powScale;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < total; i++) acts[i] = sc * outGrads[i] * acts[i] / denoms[i];
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; c++) {
        for (int px = 0; px < imgSize * imgSize; px++) {
            const int startF = blocked ? (c / size) * size : -size + size / 2 + 1 + c;
            const int f0 = blocked ? startF : (startF < 0 ? 0 : startF);
            const int f1 = startF + size < C ? startF + size : C;
            for (int n = 0; n < N; n++) {
                double s = 0.0;
                for (int f = f0; f < f1; f++) s += acts[((long)f * imgPix + px) * N + n];
                const long i = ((long)c * imgPix + px) * N + n;
                const double v = (double)inputs[i] * s + (double)outGrads[i] * pow((double)denoms[i], -(double)powScale);
                target[i] = blend(target[i], v, scaleTargets, scaleOutput);
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * Dense product used by FCLayer (Kernel/src/layers/layer.cu:373-409) and the
 * cccp plugin (PluginsSrc/src/extralayers.cu:66-174), which call
 * NVMatrix::addProduct -> cublasSgemm (Kernel/src/nvmatrix/nvmatrix.cu:189-239).
 * Row-major C(m,n) = sT*C + sO * op(A)(m,k) * op(B)(k,n); ta/tb: the operand is
 * stored transposed ((k,m) resp. (n,k) row-major).
 * ---------------------------------------------------------------------- */
void gold_gemm(const float* A, const float* B, float* Cm, int m, int n, int k, int ta, int tb,
               float scaleTargets, float scaleOutput) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; i++) {
        double* acc = (double*)malloc(sizeof(double) * n);
        for (int j = 0; j < n; j++) acc[j] = 0.0;
        for (int kk = 0; kk < k; kk++) {
            const double a = ta ? A[(long)kk * m + i] : A[(long)i * k + kk];
            if (!tb) {
                const float* b = B + (long)kk * n;
                for (int j = 0; j < n; j++) acc[j] += a * (double)b[j];
            } else {
                for (int j = 0; j < n; j++) acc[j] += a * (double)B[(long)j * k + kk];
            }
        }
        for (int j = 0; j < n; j++) Cm[(long)i * n + j] = blend(Cm[(long)i * n + j], acc[j], scaleTargets, scaleOutput);
        free(acc);
    }
}
