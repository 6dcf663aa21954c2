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

// Exact unsigned 32-bit division by a launch-time constant d >= 2 (branch-free round-up method):
//   q = (t + ((n - t) >> 1)) >> sh,  t = umulhi(m, n).  d == 1 is encoded as m = 0, sh = 32 (q = n).
struct FastDiv {
    unsigned m, sh;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    if (d <= 1) { f.m = 0; f.sh = 32; return f; }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;  // ceil(log2 d) >= 1
    f.m = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
    f.sh = s - 1;
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) {
    if (f.sh == 32) return n;
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> f.sh;
}
// sum over the 16 lanes of a DPP row (all lanes end with the row total): two quad permutes, a
// half-row mirror and a row mirror, each folded into the v_add as a DPP operand.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
#endif
