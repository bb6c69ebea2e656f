// How fast can the producer waves of a workgroup gather im2col-like segments out of an L2-resident tensor?
// One 256-thread workgroup per CU (the engine's four producer waves), each thread keeps INFL 16-byte loads in flight and
// writes what arrives to LDS.  Per wave-instruction the 64 lanes read SEGS segments of 1024 / SEGS contiguous bytes each,
// at pseudo-random segment starts inside a footprint of FOOT bytes per XCD-sized region.
//   hipcc --offload-arch=gfx950 -O3 tools/census/l2_gather_probe.hip -o l2_gather_probe && ./l2_gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INFL, bool BUF>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ src, unsigned footBytes, int segBytes, int iters, float* sink) {
    __shared__ float4 lds[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lanesPerSeg = segBytes / 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, footBytes, 0x00020000);
    unsigned state = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    u32x4 r[INFL];
    auto addr = [&](int j) {
        // every group of lanesPerSeg lanes shares one segment start (wave-uniform hash + the segment's index in the instruction)
        state = state * 1664525u + 1013904223u;
        const unsigned seg = (state >> 8) + (unsigned)(lane / lanesPerSeg) * 7919u;
        const unsigned start = (seg % (footBytes / (unsigned)segBytes)) * (unsigned)segBytes;
        return start + (unsigned)(lane % lanesPerSeg) * 16u;
    };
#pragma unroll
    for (int j = 0; j < INFL; ++j) {
        const unsigned off = addr(j);
        if (BUF) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        else r[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + off);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INFL; ++j) {
            lds[tid + 256 * (j & 7)] = make_float4(__uint_as_float(r[j].x), __uint_as_float(r[j].y), __uint_as_float(r[j].z), __uint_as_float(r[j].w));
            const unsigned off = addr(j);
            if (BUF) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            else r[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + off);
        }
        __syncthreads();
        const float4 v = lds[(tid * 7 + it) & 2047];
        acc.x += v.x;
    }
#pragma unroll
    for (int j = 0; j < INFL; ++j) acc.y += __uint_as_float(r[j].x);
    if (acc.x + acc.y == 123.456f) sink[0] = acc.x;
}

template <int INFL, bool BUF>
void run(const float* d, unsigned foot, int seg, float* sink, int cus) {
    const int iters = 400;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<INFL, BUF>), dim3(cus), dim3(256), 0, 0, d, foot, seg, 20, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<INFL, BUF>), dim3(cus), dim3(256), 0, 0, d, foot, seg, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)cus * 256 * 16 * INFL * (iters + 1);
    printf("%s  in flight %3d KB/CU  segments of %4d B  footprint %5.1f MB : %6.1f GB/s per CU, %5.2f TB/s chip\n", BUF ? "buffer_load" : "global_load",
           INFL * 4, seg, foot / 1048576.0, bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t maxFoot = 256u << 20;
    float *d, *sink;
    hipMalloc(&d, maxFoot); hipMalloc(&sink, 64);
    hipMemset(d, 0, maxFoot);
    for (unsigned foot : {1u << 20, 4u << 20, 32u << 20}) {
        for (int seg : {1024, 512, 128}) {
            run<3, true>(d, foot, seg, sink, cus);
            run<9, true>(d, foot, seg, sink, cus);
            run<18, true>(d, foot, seg, sink, cus);
            run<9, false>(d, foot, seg, sink, cus);
        }
    }
    return 0;
}
