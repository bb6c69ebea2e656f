// What does v_mfma_f32_16x16x4_f32 sustain from one wave per SIMD, bare and with the ingredients of the 16x16x4 consumer loop of
// igemm_core.h?  Prints shader cycles per matrix instruction (s_memtime) and the clock (against s_memrealtime, 100 MHz).
// usage: mfma16_probe [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEPS = 64;

// V 0: 30 accumulators of 16x16x4, nothing else      V 1: 6 accumulators of 32x32x2, nothing else
// V 2: V 0 + two VALU selects per MFMA               V 3: V 2 + 11 ds_read_b128 per 60 MFMAs (single fragment set)
// V 4: V 0 with 8 accumulators only                  V 5: V 0 with a second, idle-ish wave per SIMD (512 threads; waves 4-7 spin on s_sleep)
template <int V>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float sm[8 * 400 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8 * 400 * 4; i += blockDim.x) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    if (tid >= 256) {   // V 5 only
        for (int t = 0; t < STEPS * 30; ++t) __builtin_amdgcn_s_sleep(8);
        return;
    }
    const bool hi = lane >= 32;
    f32x4 acc[30];
    f32x16 big[6];
    for (int a = 0; a < 30; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) big[a][r] = 0.f;
    float4 fa[5], fb[6];
    for (int a = 0; a < 5; ++a) fa[a] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
    for (int b = 0; b < 6; ++b) fb[b] = make_float4(0.5f, 0.25f, 1.f, 2.f);
    float a0[5], b0[6];
    for (int a = 0; a < 5; ++a) a0[a] = fa[a].x;
    for (int b = 0; b < 6; ++b) b0[b] = fb[b].x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < STEPS; ++t) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {   // 8 x 30 = 240 MFMAs per step (V 1: 8 x 15 of twice the work each)
            if (V == 1) {
#pragma unroll
                for (int r = 0; r < 15; ++r) big[r % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r % 5], b0[r % 6], big[r % 6], 0, 0, 0);
                continue;
            }
            if (V == 3) {
#pragma unroll
                for (int a = 0; a < 5; ++a) fa[a] = reinterpret_cast<const float4*>(sm)[((g & 7) * 2 + (lane >> 4 & 1)) * 193 + a * 16 + (lane & 15)];
#pragma unroll
                for (int b = 0; b < 6; ++b) fb[b] = reinterpret_cast<const float4*>(sm)[((g & 7) * 2 + (lane >> 4 & 1)) * 193 + 80 + b * 16 + (lane & 15)];
            }
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const int i = V == 4 ? (a * 6 + b) % 8 : a * 6 + b;
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[a], b0[b], acc[i], 0, 0, 0);
                }
            if (V == 2 || V == 3) {
#pragma unroll
                for (int a = 0; a < 5; ++a) a0[a] = hi ? fa[a].y : fa[a].x;
#pragma unroll
                for (int b = 0; b < 6; ++b) b0[b] = hi ? fb[b].w : fb[b].z;
#pragma unroll
                for (int a = 0; a < 5; ++a) { fa[a].x += 1.f; fa[a].y += 1.f; }
#pragma unroll
                for (int b = 0; b < 6; ++b) { fb[b].z += 1.f; fb[b].w += 1.f; }
#pragma unroll
                for (int i = 0; i < 30; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < 30; ++a) s += acc[a][0] + acc[a][3];
    for (int a = 0; a < 6; ++a) s += big[a][0];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int V>
void run(int blocks, float* out, unsigned long long* cyc, const char* what) {
    const int threads = V == 5 ? 512 : 256;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
    hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16);
    hipMemcpy(h, cyc, blocks * 16, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int b = 0; b < blocks; b++) { c += h[2 * b]; r += h[2 * b + 1]; }
    c /= blocks; r /= blocks;
    const double per = c / (STEPS * (V == 1 ? 120.0 : 240.0));
    printf("V%d %-58s %7.1f cycles per MFMA (ideal %d)  clock %.2f GHz\n", V, what, per, V == 1 ? 64 : 32, c / r * 0.1);
    free(h);
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 16);
    run<0>(blocks, out, cyc, "16x16x4, 30 accumulators, bare");
    run<4>(blocks, out, cyc, "16x16x4, 8 accumulators, bare");
    run<1>(blocks, out, cyc, "32x32x2, 6 accumulators, bare");
    run<2>(blocks, out, cyc, "16x16x4 + 2 VALU per MFMA");
    run<3>(blocks, out, cyc, "16x16x4 + 2 VALU per MFMA + 11 ds_read_b128 per 30");
    run<5>(blocks, out, cyc, "16x16x4 bare, a sleeping second wave per SIMD");
    return 0;
}
