#pragma once
// Philox4x32-10 as the library's counter-based uniform stream: block b of the stream (seed, offset) gives the four uniforms of
// elements 4b .. 4b + 3 (offset is a multiple of 4).  One definition for every kernel that regenerates a mask instead of reading it.
#include <hip/hip_runtime.h>

namespace mc {
namespace {

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// the four uniforms of block b of stream (seed, offset): what k_rand_uniform writes to elements 4b .. 4b + 3
__device__ __forceinline__ void philox_uniform4(unsigned long long seed, unsigned long long offset, size_t b, float u[4]) {
    const unsigned long long ctr = offset / 4 + b;
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = (unsigned)(offset & 3), c3 = 0x6d633535u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u[0] = u01(c0); u[1] = u01(c1); u[2] = u01(c2); u[3] = u01(c3);
}

}  // namespace
}  // namespace mc
