// common.h -- shared helpers for the gfx950 kernels of libir2rgb_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ir2rgb_hip.h"

#define IR2RGB_WAVE 64

static inline int ir2rgb_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IR2RGB_OK : (int)e;
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid size for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8 blocks.
static inline int stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}
