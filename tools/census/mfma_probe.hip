// Where does an fp32-MFMA tile loop lose time?  Variants add one ingredient at a time.
// usage: mfma_probe <blocks> <ldsBytes>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MT = 3, SLABS = 16, STEPS = 150;

template <int V>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ g, float* out, int ld) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    f32x16 acc[MT];
    for (int a = 0; a < MT; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = tid; i < 2 * 32 * (96 + 128); i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    float af[MT] = {1.f + lane, 2.f, 3.f}, bf = 0.5f;
    float4 r[4];
    const float* gp = g + (size_t)blockIdx.x * 4096 + tid * 4;
    for (int t = 0; t < STEPS; ++t) {
        const float* cA = sm + (t & 1) * 32 * 96;
        const float* cB = sm + 2 * 32 * 96 + (t & 1) * 32 * 128;
        if (V >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(gp + (size_t)((t * 4 + i) % 64) * ld);
        }
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
            if (V >= 1) {
                const int k = (s >> 2) * 8 + 4 * h + (s & 3);
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a] = cA[k * 96 + a * 32 + l31];
                bf = cB[k * 128 + wave * 32 + l31];
            }
#pragma unroll
            for (int a = 0; a < MT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf, acc[a], 0, 0, 0);
        }
        if (V >= 3) {
            float4* l4 = reinterpret_cast<float4*>(sm + ((t + 1) & 1) * 32 * 96);
            float4 v = V >= 4 ? r[0] : make_float4(af[0], af[1], af[2], bf);
#pragma unroll
            for (int i = 0; i < 4; ++i) l4[tid + i * 256 < 768 ? tid + i * 256 : tid] = V >= 4 ? r[i] : v;
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < MT; ++a) for (int r2 = 0; r2 < 16; ++r2) s += acc[a][r2];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(int blocks, int lds, const float* g, float* out) {
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), lds, 0, g, out, 1 << 20);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), lds, 0, g, out, 1 << 20);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    double flops = (double)blocks * 4 * STEPS * SLABS * MT * 4096.0;
    printf("variant %d (%s): %.3f ms  %.1f TF/s\n", V, V == 0 ? "MFMA only" : V == 1 ? "+LDS fragment reads" : V == 2 ? "+barrier per step" : V == 3 ? "+LDS stores" : "+global loads", ms, flops / ms / 1e9);
}
int main(int argc, char** argv) {
    int blocks = atoi(argv[1]), lds = atoi(argv[2]);
    float *g, *out; hipMalloc(&g, (size_t)(64 + 4) * (1 << 20) * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemset(g, 0, (size_t)(64 + 4) * (1 << 20) * 4);
    printf("blocks=%d lds=%d\n", blocks, lds);
    run<0>(blocks, lds, g, out); run<1>(blocks, lds, g, out); run<2>(blocks, lds, g, out); run<3>(blocks, lds, g, out); run<4>(blocks, lds, g, out);
    return 0;
}
