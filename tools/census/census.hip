// Block -> CU placement census: how many workgroups of a grid land on each CU at the same time.
// usage: census <blocks> <threads> <ldsBytes> <spin_us>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ void k(unsigned* out, long spin) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        sm[0] = 1.f;
    }
    long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) {}
}
int main(int argc, char** argv) {
    int blocks = atoi(argv[1]), threads = atoi(argv[2]), lds = atoi(argv[3]);
    long spin = atol(argv[4]) * 100;   // 100 MHz ticks
    unsigned* d; hipMalloc(&d, blocks * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 2); hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> perCU;
    for (int b = 0; b < blocks; b++) {
        unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xF;
        unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        perCU[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : perCU) hist[kv.second]++;
    printf("blocks=%d threads=%d lds=%d: distinct CUs used=%zu; blocks-per-CU histogram:", blocks, threads, lds, perCU.size());
    for (auto& kv : hist) printf(" %dx:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
