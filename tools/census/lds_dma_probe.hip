// What does an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) write for a lane whose offset is out of the descriptor's range:
// zeros (like the register form returns), or nothing?  Prints the LDS image of one wave: even lanes in range, odd lanes out of range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) smem[i] = -7.f;   // marker
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, bytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned off = (threadIdx.x & 1) ? 0xFFFFFFFFu : threadIdx.x * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = smem[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.f + i;
    float *g, *o;
    hipMalloc(&g, 4096); hipMalloc(&o, 2048);
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 2048, 0, g, o, 4096u);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 6; ++lane) printf("wave 0 lane %d: %g %g %g %g\n", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
    for (int lane = 0; lane < 4; ++lane) printf("wave 1 lane %d: %g %g %g %g\n", lane, r[256 + lane * 4], r[256 + lane * 4 + 1], r[256 + lane * 4 + 2], r[256 + lane * 4 + 3]);
    return 0;
}
