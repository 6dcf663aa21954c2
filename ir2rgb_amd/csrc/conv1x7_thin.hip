// conv1x7_thin.hip -- the 1x7 pass of the separable generator heads (reference models/networks.py:165-171:
// ReflectionPad2d(3) + Conv2d(ngf, 3 | 2 | 1, 7); ir2rgb_amd.layers.head_stage evaluates the 7x7 kernel as a 1x7 pass
// producing Cout*7 <= 24 row responses per pixel, then a 7-tap vertical gather with bias / tanh / sigmoid).
//
// On the general kernel (conv_igemm_kernel<.., 1, 7, THIN>) this layer re-staged the 128-channel activations once per
// tap -- 7 x 134 MB through L2 -> LDS for 22 GFLOP -- and ran 195 us inside the forward where its tensors are 30 us of
// HBM time.  Here:
//   * a workgroup (4 waves, one per SIMD, the whole 512-register file each) walks row segments of 128 output pixels;
//     the segment's 134 input pixels (3 reflected halo pixels each side) x CIN channels are staged ONCE by LDS-DMA and
//     every tap reads them shifted by kx rows -- the activations cross L2 -> LDS once;
//   * the weights never enter LDS: all CIN/32 x 7 K-steps x 2 row tiles of MFMA A-fragments (<= 224 registers) are
//     loaded from the packed weights once per workgroup and stay in registers for all its segments;
//   * LDS image: pixel rows of CIN*2 bytes, 16-byte chunk c of row p at position c ^ (p & 15) (CIN 128) resp.
//     c ^ ((p >> 1) & 7) (CIN 64: two pixels per 256-byte bank row): the 16 pixels a ds_read_b128 lane group covers
//     hit 16 distinct bank windows for every tap shift; the swizzle is applied on the DMA's per-lane SOURCE address.
//   * two segment buffers: the next segment's DMA is issued before the current one is multiplied.
// Output: fp32 [pixels][ldy] row responses (the same tensor head_finish reads).  Bound: HBM (input once + 96 B/pixel out).
#include <utility>

#include "common.h"
#include "conv1x7_thin.h"

typedef __attribute__((ext_vector_type(8))) __bf16 t7_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 t7_f16x8;
typedef __attribute__((ext_vector_type(4))) float t7_f32x4;
typedef __attribute__((address_space(3))) void *t7_lptr_t;
typedef __amdgpu_buffer_rsrc_t t7_rsrc_t;
#define T7_OOB 0x80000000u

template <int DT> struct T7Half;
template <> struct T7Half<IR2RGB_BF16> {
    typedef t7_bf16x8 frag;
    static __device__ __forceinline__ t7_f32x4 mfma(frag a, frag b, t7_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct T7Half<IR2RGB_F16> {
    typedef t7_f16x8 frag;
    static __device__ __forceinline__ t7_f32x4 mfma(frag a, frag b, t7_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <int DT, int CIN>
__global__ void __launch_bounds__(256, 1)
conv1x7_thin_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, float *__restrict__ Y, const T7Geom g) {
    typedef T7Half<DT> Hf;
    typedef typename Hf::frag frag;
    constexpr int TW = 128, KW = 7, PAD = 3, ROWS = TW + KW - 1;          // 134 staged pixels per segment
    constexpr int NC = CIN / 32;                                          // 32-channel K-steps per tap
    constexpr int RB = CIN * 2;                                           // bytes per LDS pixel row
    constexpr int RPI = 1024 / RB;                                        // pixel rows per DMA instruction (4 | 8)
    constexpr int NDMA = (ROWS + RPI - 1) / RPI;                          // DMA instructions per segment
    constexpr int NDW = (NDMA + 3) / 4;                                   // ... per wave
    constexpr int SEG = NDMA * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SEG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const t7_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(X), 0, (int)g.x_bytes, 0x00020000);
    const t7_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(Wp), 0, (int)g.w_bytes, 0x00020000);

    // ---- this wave's MFMA A-fragments: rows (row responses) mt*16 + l15, K-step j = kx*NC + c: channels c*32 + grp*8 .. +7 of
    // tap kx.  Packed weights Wp[row][CIN/64][7][64]; rows >= Cout lie behind the resource's extent and read zeros.
    frag A[KW * NC][2];
#pragma unroll
    for (int j = 0; j < KW * NC; ++j) {
        const int kx = j / NC, c = j % NC;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 16 + l15;
            const unsigned off = (unsigned)((((row * (CIN / 64) + (c >> 1)) * KW + kx) * 64 + (c & 1) * 32 + grp * 8) * 2);
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, row < g.Cout ? off : T7_OOB, 0, 0);
            A[j][mt] = __builtin_bit_cast(frag, v);
        }
    }

    // ---- staging roles: DMA instruction i = wave + 4*q covers LDS rows RPI*i .. RPI*i + RPI - 1
    auto seg_coords = [&](long seg, int &n, int &y, int &x0) {
        const int sx = (int)(seg % g.nsx);
        const long r = seg / g.nsx;
        y = (int)(r % g.H);
        n = (int)(r / g.H);
        x0 = sx * TW;
    };
    auto issue = [&](long seg, int buf) {
        int n, y, x0;
        seg_coords(seg, n, y, x0);
        const unsigned rowbase = (unsigned)(((long)n * g.H + y) * g.W);
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            const int i = wave + 4 * q;                                   // wave-uniform
            const int row = i * RPI + (CIN == 128 ? (lane >> 4) : (lane >> 3));
            const int pos = CIN == 128 ? (lane & 15) : (lane & 7);
            const int chunk = CIN == 128 ? (pos ^ (row & 15)) : (pos ^ ((row >> 1) & 7));
            int xs = x0 - PAD + row;
            xs = xs < 0 ? -xs : xs;
            xs = xs >= g.W ? 2 * g.W - 2 - xs : xs;
            const bool ok = row < ROWS && xs >= 0 && xs < g.W && seg < g.nseg;
            const unsigned v = ok ? ((rowbase + (unsigned)xs) * (unsigned)g.ldx + (unsigned)g.ci_off + (unsigned)chunk * 8u) * 2u : T7_OOB;
            if (4 * q + 3 < NDMA || i < NDMA)                             // (only the last round can run past the segment)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (t7_lptr_t)(smem + buf * SEG + i * 1024), 16, v, 0, 0, 0);
        }
    };

    // ---- fragment addresses: output pixel p = wave*32 + b*16 + l15 reads LDS row p + kx for tap kx.  With cg = c*4 + grp and
    // grp < 4 the swizzled position is (c*4) ^ (grp ^ f(row)): the address of K-step (kx, c), block b is
    // (base[kx] ^ (c << 6)) + b * 16 rows -- seven per-lane bases, one v_xor per read, b as an immediate.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem;
    unsigned base[KW];
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
        const int row = wave * 32 + l15 + kx;
        const int f = CIN == 128 ? (row & 15) : ((row >> 1) & 7);
        base[kx] = lds0 + (unsigned)(row * RB + ((grp ^ f) & (CIN == 128 ? 15 : 7)) * 16);
    }
    // The reads are inline asm with hand-counted waits, R fragments in flight: left to itself hipcc keeps ONE pixel
    // fragment register (the weights fill the rest of the file) and waits for every read right behind its issue -- 56 LDS
    // round trips per segment, 4.7 us where the MFMAs need 0.9.  A wait names the fragment it makes valid ("+v"), so the
    // MFMAs that read it cannot move above it; sched_barrier keeps the order.
    constexpr int R = 8, NT = KW * NC * 2;                              // reads per segment: (kx, c, b), b fastest
    auto rd = [&]<int IMM>(frag &dst, unsigned addr, std::integral_constant<int, IMM>) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
    };

    long seg = blockIdx.x;
    int buf = 0;
    issue(seg, 0);
    for (; seg < g.nseg; seg += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this segment has landed (this wave's part); the previous one's reads have returned
        __builtin_amdgcn_s_barrier();                                     // ... for every wave
        issue(seg + gridDim.x, buf ^ 1);                                  // next segment into the other buffer (no-op rows past the end)
        const unsigned boff = (unsigned)(buf * SEG);
        t7_f32x4 acc[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[b][mt] = (t7_f32x4){0.f, 0.f, 0.f, 0.f};
        frag B[R];
        auto fetch = [&]<int T>(std::integral_constant<int, T>) {
            constexpr int j = T / 2, bb = T % 2, kx = j / NC, c = j % NC;
            rd(B[T % R], (base[kx] ^ (unsigned)(c << 6)) + boff, std::integral_constant<int, bb * 16 * RB>{});
        };
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (fetch(std::integral_constant<int, Ts>{}), ...); }(std::make_integer_sequence<int, R>{});
        auto body = [&]<int T>(std::integral_constant<int, T>) {
            constexpr int j = T / 2, bb = T % 2;
            constexpr int after = (NT - 1 - T) < (R - 1) ? (NT - 1 - T) : (R - 1);   // reads issued after read T, still allowed in flight
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(B[T % R]) : "n"(after) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[bb][0] = Hf::mfma(A[j][0], B[T % R], acc[bb][0]);
            acc[bb][1] = Hf::mfma(A[j][1], B[T % R], acc[bb][1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (T + R < NT) fetch(std::integral_constant<int, T + R>{});
        };
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (body(std::integral_constant<int, Ts>{}), ...); }(std::make_integer_sequence<int, NT>{});
        // ---- store: lane holds rows mt*16 + grp*4 .. +3 of pixel p
        int n, y, x0;
        seg_coords(seg, n, y, x0);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int x = x0 + wave * 32 + b * 16 + l15;
            if (x < g.W) {
                float *dst = Y + (((long)n * g.H + y) * g.W + x) * g.ldy + g.co_off;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int r0 = mt * 16 + grp * 4;
                    if (r0 + 3 < g.Cout) {
                        *reinterpret_cast<float4 *>(dst + r0) = make_float4(acc[b][mt][0], acc[b][mt][1], acc[b][mt][2], acc[b][mt][3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r0 + r < g.Cout) dst[r0 + r] = acc[b][mt][r];
                    }
                }
            }
        }
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the padding DMAs of the last round)
}

// ----------------------------------------------------------------------------------------
// host side (called from conv_mfma.hip)
// ----------------------------------------------------------------------------------------
static bool t7_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV1X7_THIN"); v = e ? atoi(e) : 1; }
    return v != 0;
}

bool conv1x7_thin_plan(const ir2rgb_conv_desc *d, T7Geom *g) {
    if (!t7_enabled() || d->transposed || d->kh != 1 || d->kw != 7 || d->stride_h != 1 || d->stride_w != 1) return false;
    if (d->pad_h != 0 || d->pad_w != 3 || d->pad_mode != 1 || !d->out_f32 || d->act != 0 || d->stats_per_sample) return false;
    if (d->Cout < 1 || d->Cout > 32 || (d->Cin != 64 && d->Cin != 128) || d->Win < 4 || d->Hout != d->Hin || d->Wout != d->Win) return false;
    if (d->dtype != IR2RGB_BF16 && d->dtype != IR2RGB_F16) return false;
    const int ldx = d->ldx > 0 ? d->ldx : d->Cin, ldy = d->ldy > 0 ? d->ldy : d->Cout;
    if ((ldx & 7) || (d->ci_off & 7) || (ldy & 3) || (d->co_off & 3)) return false;
    const long xb = (long)d->N * d->Hin * d->Win * ldx * 2;
    if (xb >= (1L << 31)) return false;
    *g = T7Geom{};
    g->N = d->N; g->H = d->Hin; g->W = d->Win; g->Cout = d->Cout;
    g->ldx = ldx; g->ci_off = d->ci_off; g->ldy = ldy; g->co_off = d->co_off;
    g->nsx = (d->Win + 127) / 128;
    g->nseg = (long)d->N * d->Hin * g->nsx;
    g->x_bytes = (unsigned)xb; g->w_bytes = (unsigned)((long)d->Cout * d->Cin * 7 * 2);
    return true;
}

int conv1x7_thin_launch(const T7Geom &g, int dtype, int cin, const void *x, const void *wp, void *y, hipStream_t s) {
    // 512 workgroups (two rounds of the chip): >= 8 segments each at 512 x 1024, which pays for the register-resident weights
    const long want = g.nseg < 512 ? g.nseg : 512;
    const unsigned grid = (unsigned)(want < 1 ? 1 : want);
    const uint16_t *X = (const uint16_t *)x, *W = (const uint16_t *)wp;
    float *Y = (float *)y;
    if (dtype == IR2RGB_BF16) {
        if (cin == 128) conv1x7_thin_kernel<IR2RGB_BF16, 128><<<grid, 256, 0, s>>>(X, W, Y, g);
        else conv1x7_thin_kernel<IR2RGB_BF16, 64><<<grid, 256, 0, s>>>(X, W, Y, g);
    } else {
        if (cin == 128) conv1x7_thin_kernel<IR2RGB_F16, 128><<<grid, 256, 0, s>>>(X, W, Y, g);
        else conv1x7_thin_kernel<IR2RGB_F16, 64><<<grid, 256, 0, s>>>(X, W, Y, g);
    }
    return ir2rgb_launch_status();
}
