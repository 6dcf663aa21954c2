// correlation_mfma.hip -- FlowNetC's cost volume on half-precision NHWC feature maps as banded MFMA products.
//
// Reference: Correlation(pad_size 20, kernel_size 1, max_displacement 20, stride1 1, stride2 2) of FlowNetC.py:27 /
// correlation_cuda_kernel.cu:  out[n][(dy+10)*21 + (dx+10)][y][x] = (1/C) sum_c f1[n][c][y][x] * f2[n][c][y+2dy][x+2dx],
// dy, dx in -10..10, zero outside the image.  In the FlowNet2 pipeline of this package f1 / f2 are the half-precision
// NHWC outputs of conv3 (flownet2_hip.flownetc), so every product is exact in fp32 and only the order of the fp32
// additions differs from the scalar kernel (correlation.hip, which stays the operator for fp32 NCHW inputs).
//
// The displacements are even, so pixels of column parity p meet only pixels of parity p: per output row y and parity p
// the 21 displacement rows are 21 banded products  M[i][j] = sum_c f1[y][2i+p][c] * f2[y+2dy][2j+p][c],  |j - i| <= 10,
// i.e. for a 16-pixel block of i the three 16-pixel blocks j = ib-1, ib, ib+1 of a 16x16x32 MFMA (21 of 48 columns used:
// 4.2 GFLOP of MFMA work for 1.85 GFLOP of correlation at [1,256,64,128] -- 2 us of matrix time; the kernel is bound by
// streaming the f2 rows into LDS).  A workgroup owns (n, y, p, half of the dy range): four multiplying waves (one i-block
// each, their f1 fragments live in registers for the whole kernel), two staging waves that keep three f2 rows in flight
// by LDS-DMA; one s_barrier per f2 row.  LDS row image: pixel j at j * 2C bytes, 16-byte chunk c of a pixel at position
// c ^ (j & 15) (conflict-free ds_read_b128 of a fragment: 16 pixels x 4 chunks).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 cm_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 cm_f16x8;
typedef __attribute__((ext_vector_type(4))) float cm_f32x4;
typedef __attribute__((address_space(3))) void *cm_lptr_t;
typedef __amdgpu_buffer_rsrc_t cm_rsrc_t;
#define CM_OOB 0x80000000u

template <int DT> struct CmHalf;
template <> struct CmHalf<IR2RGB_BF16> {
    typedef cm_bf16x8 frag;
    static __device__ __forceinline__ cm_f32x4 mfma(frag a, frag b, cm_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
};
template <> struct CmHalf<IR2RGB_F16> {
    typedef cm_f16x8 frag;
    static __device__ __forceinline__ cm_f32x4 mfma(frag a, frag b, cm_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
};

struct CorrMfmaGeom {
    int N, H, W, C;
    int lda, offa, ldb, offb;        // channel-slice views of the two feature maps (elements)
    int ldo, offo;                   // OUT 1: output pixel stride / first channel (elements)
    float scale, slope;              // 1 / C; LeakyReLU slope of OUT 1 (1 = none)
    unsigned a_bytes, b_bytes;
};

constexpr int CM_RAD = 10, CM_D = 21, CM_NBUF = 4, CM_MAXC = 256, CM_NPX = 64;

// OUT 0: fp32 planes [N][441][H][W] (the reference operator's layout); OUT 1: half NHWC channel slice with LeakyReLU
template <int DT, int OUT, int KC>
__global__ void __launch_bounds__(384, 1)
corr_mfma_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B, void *__restrict__ out, const CorrMfmaGeom g) {
    typedef CmHalf<DT> Hf;
    typedef typename Hf::frag frag;
    constexpr int KMAX = KC;                                                // K-steps of 32 channels: C / 32
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CM_NBUF * CM_NPX * CM_MAXC * 2 + 1024 + 4 * 16 * CM_D * 4];
    unsigned char *const dummy = smem + CM_NBUF * CM_NPX * CM_MAXC * 2;
    float *const tiles = reinterpret_cast<float *>(dummy + 1024);          // per multiplying wave: 16 pixels x 21 values
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (n, y, p, h); consecutive ids on one XCD (they share f2 rows in its L2)
    int id;
    {
        const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
        id = (nwg & 7) ? b : (b & 7) * (nwg >> 3) + (b >> 3);
    }
    const int h = id & 1, p = (id >> 1) & 1, y = (id >> 2) % g.H, n = (id >> 2) / g.H;
    const int dr0 = h ? 1 : -CM_RAD, dr1 = h ? CM_RAD : 0;              // displacement rows of this workgroup (inclusive)
    const int W2 = (g.W - p + 1) >> 1;                                     // pixels of parity p in a row
    const int nib = (W2 + 15) >> 4;                                        // 16-pixel blocks (<= 4)
    constexpr int pitch = KC * 64, cpp = KC * 4;                           // K-steps, LDS bytes per pixel, chunks per pixel
    const int rowbytes = CM_NPX * pitch;
    // valid f2 rows: y2 = y + 2 dr in [0, H)
    int v0 = dr0, v1 = dr1;
    while (v0 <= v1 && y + 2 * v0 < 0) ++v0;
    while (v1 >= v0 && y + 2 * v1 >= g.H) --v1;
    const int nv = v1 - v0 + 1;                                            // (may be <= 0)

    if (wave >= 4) {
        // =============================== staging waves ===============================
        const int lw = wave - 4;
        const cm_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(B), 0, (int)g.b_bytes, 0x00020000);
        const int ninstr = rowbytes >> 10, per = (ninstr + 1) >> 1;        // 1 KB pieces per row, per staging wave (<= 16)
        unsigned voff[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int pid = lw + 2 * t;
            unsigned v = CM_OOB;
            if (t < per && pid < ninstr) {
                const int slot = pid * 64 + lane, j = slot / cpp, cpos = slot - j * cpp;
                const int c = cpos ^ (j & 15), x2 = 2 * j + p;
                if (x2 < g.W) v = (unsigned)(((long)x2 * g.ldb + g.offb + c * 8) * 2);
            }
            voff[t] = v;
        }
        auto issue = [&](int s) {
            if (s < nv) {
                const int y2 = y + 2 * (v0 + s);
                const unsigned soff = (unsigned)((((long)n * g.H + y2) * g.W) * g.ldb * 2);
                unsigned char *dst = smem + (s % CM_NBUF) * rowbytes;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int pid = lw + 2 * t;
                    if (t < per && pid < ninstr)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (cm_lptr_t)(dst + pid * 1024), 16, voff[t], soff, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (cm_lptr_t)dummy, 16, CM_OOB, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (cm_lptr_t)dummy, 16, CM_OOB, 0, 0, 0);
            }
        };
        issue(0); issue(1); issue(2);
        for (int s = 0; s < nv; ++s) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");             // row s has landed (rows s+1, s+2 may be in flight)
            __builtin_amdgcn_s_barrier();                                  // ... and row s-1 is consumed
            issue(s + 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =============================== multiplying waves ===============================
    const int ib = wave;                                                   // this wave's 16-pixel block of i
    const bool have = ib < nib;
    const int i_lane = ib * 16 + l15;                                      // A-fragment row of this lane
    frag a[KMAX];
    {
        const int x = 2 * i_lane + p;
        const bool ok = have && x < g.W;
        const uint16_t *src = A + (((long)n * g.H + y) * g.W + (ok ? x : 0)) * g.lda + g.offa + grp * 8;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4 *>(src + k * 32);
            a[k] = __builtin_bit_cast(frag, v);
        }
    }
    // Results leave through a wave-private LDS tile: the 16 x 16 accumulator tiles hold a pixel's 21 values on 21 lanes of
    // three tiles, and stored from there every lane of a store would touch its own cache line (measured: 27 us, all of it
    // stores).  Tile order = output order: OUT 1 [pixel][dj] (a pixel's 21 channels of this displacement row are
    // contiguous in the NHWC slice), OUT 0 [dj][pixel] (a channel plane's row; this parity's pixels are 8 bytes apart).
    float *const tile = tiles + wave * (16 * CM_D);
    auto flush = [&](int dr, bool zero) {
        if (!zero) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own tile writes have landed
        for (int e = lane; e < 16 * CM_D; e += 64) {
            const int il = OUT == 1 ? e / CM_D : e % 16, djx = OUT == 1 ? e % CM_D : e / 16;
            const int i = ib * 16 + il, x = 2 * i + p, ch = (dr + CM_RAD) * CM_D + djx;
            if (i < W2) {
                float v = zero ? 0.f : tile[e];
                if (OUT == 0) {
                    reinterpret_cast<float *>(out)[(((long)n * (CM_D * CM_D) + ch) * g.H + y) * g.W + x] = v;
                } else {
                    v = v > 0.f ? v : v * g.slope;
                    reinterpret_cast<uint16_t *>(out)[(((long)n * g.H + y) * g.W + x) * g.ldo + g.offo + ch] = Hf::cvt(v);
                }
            }
        }
    };
    // displacement rows that fall outside the image: zeros
    for (int dr = dr0; dr <= dr1; ++dr)
        if ((dr < v0 || dr > v1) && have) flush(dr, true);
    for (int s = 0; s < nv; ++s) {
        __builtin_amdgcn_s_barrier();                                      // row s has landed
        const unsigned char *row = smem + (s % CM_NBUF) * rowbytes;
        cm_f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = (cm_f32x4){0.f, 0.f, 0.f, 0.f};
        if (have) {
            // straight-line: all 3 x KC fragment reads are issued before the first MFMA waits on one (blocks outside the row
            // image are read from a clamped address and their products dropped)
            frag b[3][KMAX];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int jb = ib - 1 + t, jc = jb < 0 ? 0 : (jb > 3 ? 3 : jb);
                const int j = jc * 16 + l15;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)   // chunk k*4 + grp of pixel j sits at position (k*4 + grp) ^ (j & 15)
                    b[t][k] = *reinterpret_cast<const frag *>(row + j * pitch + (((k * 4 + grp) ^ (j & 15)) * 16));
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = Hf::mfma(a[k], b[t][k], acc[t]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int jb = ib - 1 + t;
                if (jb < 0 || jb > 3) acc[t] = (cm_f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // (the previous row's flush has read the tile: same wave, LDS operations of a wave complete in order)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int j = (ib - 1 + t) * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = grp * 4 + r, dj = j - (ib * 16 + il);
                    if (dj >= -CM_RAD && dj <= CM_RAD)
                        tile[OUT == 1 ? il * CM_D + dj + CM_RAD : (dj + CM_RAD) * 16 + il] = acc[t][r] * g.scale;
                }
            }
            flush(v0 + s, false);
        }
    }
}

// a / b: [N][H][W][lda / ldb] half, channels [offa, offa + C) / [offb, offb + C).
// out_mode 0: out = fp32 [N][441][H][W];  1: out = half [N][H][W][ldo], channels [offo, offo + 441) = LeakyReLU_slope(corr).
extern "C" int ir2rgb_correlation_nhwc_half(const void *a, int lda, int offa, const void *b, int ldb, int offb, void *out,
                                            int out_mode, int ldo, int offo, float slope, int N, int C, int H, int W,
                                            int dtype, void *stream) {
    if (!a || !b || !out || N < 1 || H < 1 || W < 1) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_EINVAL;
    if (C < 128 || C > CM_MAXC || (C % 128) || W > 2 * CM_NPX) return IR2RGB_ENOSUP;      // (the scalar operator covers the rest)
    if ((lda & 7) || (offa & 7) || (ldb & 7) || (offb & 7) || offa + C > lda || offb + C > ldb) return IR2RGB_EINVAL;
    if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return IR2RGB_EALIGN;
    if (out_mode == 1 && (offo + CM_D * CM_D > ldo || offo < 0)) return IR2RGB_EINVAL;
    const long ab = (long)N * H * W * lda * 2, bb = (long)N * H * W * ldb * 2;
    if (ab >= (1L << 31) || bb >= (1L << 31)) return IR2RGB_EINVAL;
    CorrMfmaGeom g;
    g.N = N; g.H = H; g.W = W; g.C = C; g.lda = lda; g.offa = offa; g.ldb = ldb; g.offb = offb; g.ldo = ldo; g.offo = offo;
    g.scale = 1.0f / (float)C; g.slope = slope; g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
    const unsigned grid = (unsigned)(N * H * 4);
    hipStream_t s = as_stream(stream);
    const uint16_t *A = (const uint16_t *)a, *B = (const uint16_t *)b;
#define CM_LAUNCH(DT, OUT, KC) corr_mfma_kernel<DT, OUT, KC><<<grid, 384, 0, s>>>(A, B, out, g)
    const int kc = C / 32;
    if (dtype == IR2RGB_BF16) {
        if (out_mode == 0) { if (kc == 8) CM_LAUNCH(IR2RGB_BF16, 0, 8); else CM_LAUNCH(IR2RGB_BF16, 0, 4); }
        else { if (kc == 8) CM_LAUNCH(IR2RGB_BF16, 1, 8); else CM_LAUNCH(IR2RGB_BF16, 1, 4); }
    } else {
        if (out_mode == 0) { if (kc == 8) CM_LAUNCH(IR2RGB_F16, 0, 8); else CM_LAUNCH(IR2RGB_F16, 0, 4); }
        else { if (kc == 8) CM_LAUNCH(IR2RGB_F16, 1, 8); else CM_LAUNCH(IR2RGB_F16, 1, 4); }
    }
#undef CM_LAUNCH
    return ir2rgb_launch_status();
}
