// conv7x1_col.hip -- the 7x1 pass of the generators' first layers (reference models/networks.py:141, :150, :253-255:
// ReflectionPad2d(3) + Conv2d(input_nc, ngf, 7); ir2rgb_amd.layers evaluates it as an x-im2col of the image to 64
// channels (ci*7 + kx) followed by a (7 x 1) convolution with reflection padding in y, 64 -> ngf channels).
//
// On the general kernel (conv_igemm_kernel<.., 7, 1>) every one of the seven taps re-staged its own shifted copy of the
// 64-channel rows -- 7 x 67 MB through L2 -> LDS at 512 x 1024, 138 us in the forward (541 us at 1024 x 2048) where the
// tensors are 35 us of HBM time.  Here a workgroup owns an 8 x 32 pixel tile and stages the 14 x 32 pixel rows it needs
// ONCE (LDS-DMA, 56 KB); tap ky of output row ty is simply staged row ty + ky.
//   * four waves (one per SIMD, up to 512 registers each): a wave owns 32 output channels -- its 2 x 14 MFMA A-fragments
//     (112 registers) are loaded once per workgroup from the packed weights, which never enter LDS -- and walks the
//     tile's 16-pixel blocks (all 16 at 128 output channels; at 64 the two wave pairs take 8 blocks each);
//   * workgroups are persistent over tiles with two tile buffers: the next tile's DMA is issued before the current one is
//     multiplied;
//   * LDS image: pixel rows of 128 B, 16-byte chunk c of pixel row R at position c ^ ((R >> 1) & 7) (applied on the DMA's
//     per-lane SOURCE address): two pixels share a 256-byte bank row, so the 16 pixels a ds_read_b128 lane group covers
//     hit 16 distinct bank windows.  With 32-pixel rows the swizzle term does not depend on the staged row: every
//     fragment address is one of two per-lane bases + a compile-time immediate;
//   * fragment reads are inline asm with hand-counted waits, eight in flight (see conv1x7_thin.hip for why);
//   * epilogue per pixel block: bias, BatchNorm partial sums (one statistics row per tile), 8-byte stores.
// Bound: HBM (input 1.75 x 67 MB, mostly L2 hits for the halo rows, + the output).
#include <utility>

#include "common.h"
#include "conv7x1_col.h"

typedef __attribute__((ext_vector_type(8))) __bf16 c7_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 c7_f16x8;
typedef __attribute__((ext_vector_type(4))) float c7_f32x4;
typedef __attribute__((address_space(3))) void *c7_lptr_t;
typedef __amdgpu_buffer_rsrc_t c7_rsrc_t;
#define C7_OOB 0x80000000u

template <int DT> struct C7Half;
template <> struct C7Half<IR2RGB_BF16> {
    typedef c7_bf16x8 frag;
    static __device__ __forceinline__ c7_f32x4 mfma(frag a, frag b, c7_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
};
template <> struct C7Half<IR2RGB_F16> {
    typedef c7_f16x8 frag;
    static __device__ __forceinline__ c7_f32x4 mfma(frag a, frag b, c7_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
};

template <int DT, int COUT>
__global__ void __launch_bounds__(256, 1)
conv7x1_col_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias,
                   uint16_t *__restrict__ Y, float *__restrict__ stats_partial, const C7Geom g) {
    typedef C7Half<DT> Hf;
    typedef typename Hf::frag frag;
    constexpr int TH = 8, TW = 32, KH = 7, PAD = 3, SR = TH + KH - 1;     // 14 staged rows of 32 pixels
    constexpr int NW = 4, WC = COUT / 32, WP = NW / WC;                   // waves: WC along the couts x WP along the pixel blocks
    constexpr int NDMA = SR * TW / 8;                                     // 56 one-KB DMA instructions (8 pixel rows each)
    constexpr int NDW = NDMA / NW;
    constexpr int NKS = KH * 2;                                           // K-steps: (ky, 32-channel half)
    constexpr int NB = TH * TW / 16 / WP;                                 // pixel blocks per wave (16 | 8)
    constexpr int SEG = NDMA * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SEG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int l15 = lane & 15, grp = lane >> 4;
    const c7_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(X), 0, (int)g.x_bytes, 0x00020000);
    const c7_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(Wp), 0, (int)g.w_bytes, 0x00020000);
    const c7_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)g.y_bytes, 0x00020000);
    const int ntiles = g.N * g.nty * g.ntx;

    // ---- stage a tile: instruction i = wave + 4*q covers pixel rows R = 8*i .. 8*i+7 (staged row R / 32, pixel R % 32)
    auto issue = [&](int tile, int buf) {
        const int txi = tile % g.ntx, tyi = (tile / g.ntx) % g.nty, n = tile / (g.ntx * g.nty);
        const int y0 = tyi * TH, x0 = txi * TW;
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            const int i = wave + NW * q;
            const int R = i * 8 + (lane >> 3), pos = lane & 7, chunk = pos ^ ((R >> 1) & 7);
            const int r = R >> 5, px = R & 31;
            int ys = y0 - PAD + r;
            ys = ys < 0 ? -ys : ys;
            ys = ys >= g.H ? 2 * g.H - 2 - ys : ys;
            const int xs = x0 + px;
            const bool ok = tile < ntiles && ys >= 0 && ys < g.H && xs < g.W;
            const unsigned v = ok ? ((unsigned)((n * g.H + ys) * g.W + xs) * (unsigned)g.ldx + (unsigned)g.ci_off + (unsigned)chunk * 8u) * 2u : C7_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (c7_lptr_t)(smem + buf * SEG + i * 1024), 16, v, 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    issue(tile, 0);

    // ---- this wave's A-fragments: couts wc*32 + mt*16 + l15; K-step j = ky*2 + c: channels c*32 + grp*8 .. +7 of tap ky.
    // Packed weights Wp[cout][1][7][64].
    frag A[NKS][2];
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
        const int ky = j >> 1, c = j & 1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = wc * 32 + mt * 16 + l15;
            const unsigned off = (unsigned)(((co * KH + ky) * 64 + c * 32 + grp * 8) * 2);
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            A[j][mt] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rw, co < g.Cout ? off : C7_OOB, 0, 0));
        }
    }
    float bv[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = wc * 32 + mt * 16 + grp * 4 + r;
            bv[mt][r] = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
        }

    // ---- fragment addresses: block b of this wave is tile block wp*NB + b: pixel px = (b & 1)*16 + l15 of staged row
    // (wp*NB + b) / 2 + ky.  (R >> 1) & 7 = (px >> 1) & 7 = (l15 >> 1): a per-lane constant, so
    // address = base[c] + ((blk >> 1) + ky) * 4096 + (blk & 1) * 2048 with base[c] = base[0] ^ 64.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem;
    const unsigned base0 = lds0 + (unsigned)(l15 * 128 + ((grp ^ (l15 >> 1)) & 7) * 16) + (unsigned)(wp * (NB / 2) * 4096);
    auto rd = [&]<int IMM>(frag &dst, unsigned addr, std::integral_constant<int, IMM>) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
    };
    constexpr int R = 8, NT = NB * NKS;                                   // reads per tile and wave: (block, ky, c), c fastest
    static_assert(((NB - 1) / 2 + KH - 1) * 4096 + 2048 + (WP - 1) * (NB / 2) * 4096 < SEG, "reads stay inside the tile");

    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this tile has landed (this wave's part), the stores are out, the previous tile's reads have returned
        __builtin_amdgcn_s_barrier();                                     // ... for every wave
        issue(tile + gridDim.x, buf ^ 1);                                 // next tile into the other buffer (zeros past the end)
        unsigned base[2];
        base[0] = base0 + (unsigned)(buf * SEG);
        base[1] = (base0 ^ 64u) + (unsigned)(buf * SEG);
        const int txi = tile % g.ntx, tyi = (tile / g.ntx) % g.nty, n = tile / (g.ntx * g.nty);
        const int y0 = tyi * TH + wp * (NB / 2), x0 = txi * TW;
        frag B[R];
        auto fetch = [&]<int T>(std::integral_constant<int, T>) {
            constexpr int b = T / NKS, j = T % NKS, ky = j >> 1, c = j & 1;
            rd(B[T % R], base[c], std::integral_constant<int, ((b >> 1) + ky) * 4096 + (b & 1) * 2048>{});
        };
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (fetch(std::integral_constant<int, Ts>{}), ...); }(std::make_integer_sequence<int, R>{});
        float s1[2][4], s2[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[mt][r] = s2[mt][r] = 0.f;
        c7_f32x4 acc[2];
        auto body = [&]<int T>(std::integral_constant<int, T>) {
            constexpr int b = T / NKS, j = T % NKS;
            constexpr int after = (NT - 1 - T) < (R - 1) ? (NT - 1 - T) : (R - 1);
            if constexpr (j == 0) { acc[0] = (c7_f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (c7_f32x4){0.f, 0.f, 0.f, 0.f}; }
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(B[T % R]) : "n"(after) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = Hf::mfma(A[j][0], B[T % R], acc[0]);
            acc[1] = Hf::mfma(A[j][1], B[T % R], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (T + R < NT) fetch(std::integral_constant<int, T + R>{});
            if constexpr (j == NKS - 1) {
                // pixel block b is complete: lane holds couts wc*32 + mt*16 + grp*4 .. +3 of pixel (ty, tx)
                const int oy = y0 + (b >> 1), ox = x0 + (b & 1) * 16 + l15;
                const bool valid = oy < g.H && ox < g.W;
                const unsigned pixoff = (unsigned)((n * g.H + oy) * g.W + ox) * (unsigned)g.ldy + (unsigned)g.co_off;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][r] + bv[mt][r];
                        if (valid) { s1[mt][r] += v[r]; s2[mt][r] += v[r] * v[r]; }
                    }
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    u32x2 pk;
                    pk.x = (uint32_t)Hf::cvt(v[0]) | ((uint32_t)Hf::cvt(v[1]) << 16);
                    pk.y = (uint32_t)Hf::cvt(v[2]) | ((uint32_t)Hf::cvt(v[3]) << 16);
                    const int co = wc * 32 + mt * 16 + grp * 4;
                    // (an out-of-range offset drops the store: no branch in the stream)
                    __builtin_amdgcn_raw_buffer_store_b64(pk, ry, (valid && co < g.Cout) ? (pixoff + (unsigned)co) * 2u : C7_OOB, 0, 0);
                }
            }
        };
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (body(std::integral_constant<int, Ts>{}), ...); }(std::make_integer_sequence<int, NT>{});
        if (stats_partial != nullptr) {      // one statistics row per (tile, pixel half wp)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row16_sum(s1[mt][r]), b2 = row16_sum(s2[mt][r]);
                    const int co = wc * 32 + mt * 16 + grp * 4 + r;
                    if (l15 == 0 && co < g.Cout) {
                        stats_partial[((long)(tile * WP + wp) * 2 + 0) * g.Cout + co] = a;
                        stats_partial[((long)(tile * WP + wp) * 2 + 1) * g.Cout + co] = b2;
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
static bool c7_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV7X1_COL"); v = e ? atoi(e) : 1; }
    return v != 0;
}

bool conv7x1_col_plan(const ir2rgb_conv_desc *d, C7Geom *g) {
    if (!c7_enabled() || d->transposed || d->kh != 7 || d->kw != 1 || d->stride_h != 1 || d->stride_w != 1) return false;
    if (d->pad_h != 3 || d->pad_w != 0 || d->pad_mode != 1 || d->out_f32 || d->act != 0 || d->stats_per_sample) return false;
    if (d->Cin != 64 || (d->Cout != 64 && d->Cout != 128) || d->Hin < 4 || d->Hout != d->Hin || d->Wout != d->Win) return false;
    if (d->dtype != IR2RGB_BF16 && d->dtype != IR2RGB_F16) return false;
    const int ldx = d->ldx > 0 ? d->ldx : d->Cin, ldy = d->ldy > 0 ? d->ldy : d->Cout;
    if ((ldx & 7) || (d->ci_off & 7) || (ldy & 3) || (d->co_off & 3)) return false;
    const long xb = (long)d->N * d->Hin * d->Win * ldx * 2, yb = (long)d->N * d->Hout * d->Wout * ldy * 2;
    if (xb >= (1L << 31) || yb >= (1L << 31)) return false;
    *g = C7Geom{};
    g->N = d->N; g->H = d->Hin; g->W = d->Win; g->Cout = d->Cout;
    g->ldx = ldx; g->ci_off = d->ci_off; g->ldy = ldy; g->co_off = d->co_off;
    g->nty = (d->Hin + 7) / 8; g->ntx = (d->Win + 31) / 32;
    g->x_bytes = (unsigned)xb; g->y_bytes = (unsigned)yb; g->w_bytes = (unsigned)((long)d->Cout * 64 * 7 * 2);
    return true;
}

// rows of the BatchNorm statistics buffer: one per tile and pixel half (two wave pairs split the tile at 64 output channels)
int conv7x1_col_tiles(const C7Geom &g) { return g.N * g.nty * g.ntx * (g.Cout == 64 ? 2 : 1); }

int conv7x1_col_launch(const C7Geom &g, int dtype, const void *x, const void *wp, const float *bias, void *y, float *stats,
                       hipStream_t s) {
    const int tiles = g.N * g.nty * g.ntx;
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);      // persistent over tiles: weights loaded once per workgroup
    const uint16_t *X = (const uint16_t *)x, *W = (const uint16_t *)wp;
    uint16_t *Y = (uint16_t *)y;
    if (dtype == IR2RGB_BF16) {
        if (g.Cout == 128) conv7x1_col_kernel<IR2RGB_BF16, 128><<<grid, 256, 0, s>>>(X, W, bias, Y, stats, g);
        else conv7x1_col_kernel<IR2RGB_BF16, 64><<<grid, 256, 0, s>>>(X, W, bias, Y, stats, g);
    } else {
        if (g.Cout == 128) conv7x1_col_kernel<IR2RGB_F16, 128><<<grid, 256, 0, s>>>(X, W, bias, Y, stats, g);
        else conv7x1_col_kernel<IR2RGB_F16, 64><<<grid, 256, 0, s>>>(X, W, bias, Y, stats, g);
    }
    return ir2rgb_launch_status();
}
