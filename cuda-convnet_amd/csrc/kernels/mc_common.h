// Shared helpers for the libmckernels.so translation units (HIP, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "mc_kernels.h"

namespace mc {

void set_error(const char* fmt, ...);

inline hipStream_t S(mcStream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that was just issued; kernels here are asynchronous, so this only
// reports configuration errors (bad grid, too much LDS...), as hipGetLastError does.
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MC_ELAUNCH;
    }
    return MC_OK;
}

#define MC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::mc::set_error(__VA_ARGS__);     \
            return MC_EINVAL;                 \
        }                                     \
    } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

int num_cus();

// A k-contiguous product whose WHOLE output is one 16x16x4 workgroup tile (igemm_core.h, MF tiles): 65-192 rows by 129-192
// columns.  Chunk plans are made with it, so it is a pure function of the extents.
inline bool mf_whole_output(int rows, int cols) { return rows > 64 && rows <= 192 && cols > 128 && cols <= 192; }

}  // namespace mc
