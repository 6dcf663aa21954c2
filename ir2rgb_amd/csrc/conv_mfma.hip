// conv_mfma.hip -- implicit-GEMM convolution on the gfx950 matrix cores.
//
// Covers every dense convolution of the vid2vid generator / discriminator bodies
// (reference models/networks.py:141-171, :253-271, :556-580, :678-699): 3x3 stride 1 with
// reflection padding (ResnetBlock), 3x3 / 4x4 stride 2 with zero padding, 4x4 stride 1, and
// the 3x3 stride-2 transposed convolutions (as four sub-pixel convolutions), all with
// Cin % 64 == 0.  In the reference each of these is a cuDNN call on NCHW fp32.
//
// Data layout (HBM)
//   activations  X [N*Hin*Win][Cin], Y [N*Hout*Wout][Cout]   half precision (bf16 or f16), NHWC
//   weights      Wp[class][Cout][Cin/64][ntaps][64]          half precision, K-contiguous rows;
//                K index k = (cin_chunk, tap, cin_in_chunk): the 64-channel slice of the
//                activations is re-used by all taps back-to-back (L2 / L1 hits for taps 2..n)
//   bias         [Cout] fp32; stats_partial [rows][2][Cout] fp32 (sum, sum of squares of the
//                fp32 conv outputs per pixel tile: the BatchNorm batch statistics, reduced
//                deterministically by bn_finalize -- no atomics)
//
// GEMM view: D[cout][pixel] = sum_k Wp[cout][k] * Xg[pixel][k].  Couts are the MFMA M
// dimension so that a lane ends up with 4 consecutive couts of one pixel (one 8-byte NHWC store).
//
// Workgroup = 512 threads = 8 waves (2 per SIMD), tile TC=128 couts x TP pixels, BK = 64.
// Wave grid 2 (cout) x 4 (pixel); each wave owns 64 x TP/4 outputs as 4 x (TP/64)
// v_mfma_f32_16x16x32 tiles.  Both operands are staged global -> LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4; the per-lane SOURCE address carries the im2col gather -- reflection,
// stride, sub-pixel phase -- and the XOR swizzle), three stages deep: loads for K-steps k+1
// and k+2 are in flight while k is consumed; one raw s_barrier and one counted vmcnt per
// K-step.  LDS rows are 128 B (64 halves); 16-byte slot s of row r holds chunk s ^ ((r>>1)&7),
// which makes every ds_read_b128 fragment read bank-conflict free.
//
// Algorithmic flops = 2 * pixels * Cout * ntaps * Cin; roofline bound: MFMA.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"
#include "conv1x7_thin.h"
#include "conv7x1_col.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define IR2RGB_MAX_TAPS 49

struct ConvGeom {
    int N, Hin, Win, Cin;
    int Hsub, Wsub;           // sub-grid of output positions computed by this launch
    int Hout, Wout, Cout;
    int s_in_y, s_in_x;       // input coordinate  = sub * s_in  + tap offset
    int s_out_y, s_out_x, off_y, off_x;  // output coordinate = sub * s_out + off
    int ntaps, pad_mode;      // pad_mode 0: zeros outside, 1: reflect (no edge repeat)
    int kchunks;              // Cin / 64
    int act;                  // 0: none, 1: LeakyReLU(0.2), 2: LeakyReLU(0.1) after bias
    int ldx, ci_off, ldy, co_off;  // channel-slice views: pixel stride (elements) and first channel of X / Y
    int stats_row0;           // first row of stats_partial written by this launch
    int out_f32;              // 1: Y is fp32 NHWC (head convolutions), 0: half
    FastDiv div_hw, div_w;    // exact division by Hsub*Wsub and by Wsub
    unsigned x_bytes, w_bytes; // extents of X and of this class's packed weights (buffer resources, < 2^31)
    int variant;              // tuning switches (bit 0: waves 4-7 issue their LDS-DMA after their MFMA block)
    int cout_major;           // XCD mapping: 1 = an XCD sweeps the pixel tiles of few cout tiles (weights stay in its L2)
    // taps form an nty x ntx grid: tap (ty,tx) reads input offset (dy0 + ty*dys, dx0 + tx*dxs).
    // Pure scalar arithmetic: no table load sits between the LDS-DMA issues of the K loop.
    int ntx, dy0, dys, dx0, dxs;
    int tpi;                  // > 0: pixel tiles are cut per sample, tpi tiles each (ir2rgb_conv_desc.stats_per_sample)
};


template <int DT> struct Half;
template <> struct Half<IR2RGB_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        __bf16 h = (__bf16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
};
template <> struct Half<IR2RGB_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Buffer-addressed LDS-DMA: 16 B per lane from (SGPR base + per-lane 32-bit byte offset + scalar byte
// offset) to LDS at wave-uniform base + lane * 16.  The per-lane offset is range-checked against the
// resource extent and out-of-range lanes deliver ZEROS: zero padding costs no instruction, and the
// scalar K offset rides in an SGPR, so a K-step's staging issues no VALU address arithmetic at all.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define IR2RGB_OOB 0x80000000u
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void lds_dma16(rsrc_t r, unsigned voff, unsigned soff, unsigned char *dst_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)dst_wave_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ int reflect(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * n - 2 - v : v;
}

// NTY x NTX > 0: the tap grid is a compile-time constant, the tap loop is fully unrolled and the
// gather offsets of this thread's pixel rows are precomputed for every tap (PROWS*NTY*NTX
// registers), so a K-step's staging costs ~3 VALU per row.  NTY == 0: runtime tap grid (any
// shape), offsets recomputed per K-step.
// TP = 256: 3-stage LDS ring (144 KB, one workgroup per CU, 2 waves per SIMD) for long-K layers.
// TP = 128: 2-stage ring (64 KB) and TP = 64: 3-stage ring (72 KB): two workgroups per CU (4 waves per
// SIMD, <= 128 VGPRs), so one workgroup's prologue / epilogue / barrier stalls hide under the other's
// MFMA work -- the configuration for short-K layers and for tile counts just above the CU count.
template <int TP> struct ConvTile {
    static constexpr int TC = 128;
    static constexpr int STAGE = (TC + TP) * 128;   // bytes per stage
    static constexpr int NSTAGE = TP == 128 ? 2 : 3;
    static constexpr int LDS = NSTAGE * STAGE;
};

// The kernel body; `block` is the workgroup's index inside its launch (or inside its class of a
// multi-class launch) and `smem` the workgroup's LDS ring (ConvTile<TP>::LDS bytes, 1024-aligned).
// THIN: at most 32 output channels (the 24-response 1x7 head pass, the 1-channel PatchGAN logits): only
// the waves of the first 64-row half multiply, and only its first two 16-row fragments -- an eighth of
// the MFMAs of the full 128-row tile; staging and epilogue are unchanged.
template <int DT, int TP, int NTY, int NTX, bool THIN = false>
__device__ __forceinline__ void
conv_igemm_body(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias,
                uint16_t *__restrict__ Y, float *__restrict__ stats_partial, const ConvGeom &g, const int block,
                unsigned char *const smem) {
    constexpr int TC = 128;
    constexpr int NI = TP / 64;            // 16-pixel MFMA tiles per wave along pixels
    constexpr int WROWS = THIN ? 1 : TC / 64;  // weight rows staged per thread per K-step (THIN: rows 0..63 only)
    constexpr int PROWS = TP / 64;         // pixel rows staged per thread per K-step
    constexpr int LOADS = WROWS + PROWS;   // LDS-DMA instructions per thread per K-step
    constexpr int STAGE = (TC + TP) * 128; // bytes per stage
    constexpr int NSTAGE = TP == 128 ? 2 : 3;
    constexpr int AHEAD = NSTAGE - 1;      // K-steps staged ahead of the one being multiplied
    constexpr bool STATIC_TAPS = NTY > 0;
    constexpr int NT = STATIC_TAPS ? NTY * NTX : 1;
    typedef Half<DT> H;
    typedef typename H::frag frag;

    static_assert(NSTAGE * STAGE == ConvTile<TP>::LDS, "LDS ring size");

    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as an SGPR: keeps the LDS-DMA destination (M0) provably wave-uniform, so hipcc
    // emits no waterfall loop around global_load_lds.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned P = (unsigned)g.N * g.Hsub * g.Wsub;  // host guarantees < 2^31
    const unsigned HW = (unsigned)g.Hsub * g.Wsub;
    const int npt = g.tpi ? g.N * g.tpi : (int)((P + TP - 1) / TP), nct = (g.Cout + TC - 1) / TC;

    // XCD-aware bijective remap: consecutive tile ids land on one XCD (blocks b, b+8, ... share an L2).
    // Pixel-major ids (an XCD owns a run of pixel tiles and streams ALL weights through its L2) suit layers
    // whose activations outweigh their weights; cout-major ids (an XCD owns about Cout/8 output channels,
    // whose weight slice then lives in its 4 MB L2, and reads every pixel) suit the weight-heavy deep
    // layers: measured on 1024->1024 3x3 @32x64, fabric traffic 158 MB -> see profiles (the weights were
    // fetched 8 times).
    int tile;
    {
        const int nwg = npt * nct, b = block, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        if (b >= nwg) return;   // padding workgroups of a multi-class launch (workgroup-uniform)
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int pt, ct;
    if (g.cout_major) { ct = tile / npt; pt = tile - ct * npt; }
    else              { pt = tile / nct; ct = tile - pt * nct; }
    // pixels of this tile: p0 + i, valid below plim; flattened over the batch, or (tpi > 0) inside sample pimg only, so
    // that a row of the statistics buffer never mixes samples (BatchNorm per sample group, ir2rgb_conv_desc)
    unsigned p0 = (unsigned)pt * TP, plim = P, pimg = 0;
    if (g.tpi) { pimg = (unsigned)(pt / g.tpi); p0 = (unsigned)(pt - (int)pimg * g.tpi) * TP; plim = HW; }
    auto pix_decode = [&](unsigned p, unsigned &n, unsigned &rem) {
        if (g.tpi) { n = pimg; rem = p; }
        else { n = fdiv(p, g.div_hw); rem = p - n * HW; }
    };

    // ---------------- per-thread staging roles ----------------
    const int r8 = tid >> 3, slot = tid & 7;
    const int chunk = slot ^ ((r8 >> 1) & 7);  // source chunk for this thread's LDS slot (swizzle)
    const long Ktot = (long)g.kchunks * g.ntaps * 64;

    const rsrc_t rw = make_rsrc(Wp, g.w_bytes), rx = make_rsrc(X, g.x_bytes);
    unsigned woff[WROWS];  // byte offset of this thread's chunk in weight row co (K-step offset is scalar)
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
        int co = ct * TC + r8 + 64 * i;
        co = co < g.Cout ? co : g.Cout - 1;  // clamp: rows past Cout are never stored
        woff[i] = (unsigned)(((long)co * Ktot + chunk * 8) * 2);
    }
    int psy[PROWS], psx[PROWS];  // sub * s_in: input coordinates before the tap offset
    unsigned pbase[PROWS];       // n * Hin * Win (pixel index)
#pragma unroll
    for (int i = 0; i < PROWS; ++i) {
        unsigned p = p0 + r8 + 64 * i;
        const bool v = p < plim;
        p = v ? p : 0u;
        unsigned n, rem;
        pix_decode(p, n, rem);
        unsigned sy = fdiv(rem, g.div_w);
        // rows past the last pixel: zero padding -> parked outside the image (zero fill);
        // reflection -> pixel 0 (harmless, never stored)
        psy[i] = v ? (int)sy * g.s_in_y : (g.pad_mode ? 0 : -(1 << 20));
        psx[i] = (int)(rem - sy * g.Wsub) * g.s_in_x;
        pbase[i] = n * (unsigned)(g.Hin * g.Win);
    }
    unsigned char *const wave_dst = smem + wave * 1024;  // + stage*STAGE + 8192*i (+TC*128 for pixels)

    // byte offset of (row i, tap) incl. this thread's chunk; IR2RGB_OOB = padded with zeros
    auto gather_off = [&](int i, int dy, int dx) -> unsigned {
        int iy = psy[i] + dy, ix = psx[i] + dx;
        const bool inb = ((unsigned)iy < (unsigned)g.Hin) & ((unsigned)ix < (unsigned)g.Win);
        iy = g.pad_mode ? reflect(iy, g.Hin) : iy;
        ix = g.pad_mode ? reflect(ix, g.Win) : ix;
        const unsigned off = ((pbase[i] + (unsigned)(iy * g.Win + ix)) * (unsigned)g.ldx + g.ci_off + chunk * 8) * 2u;
        return (g.pad_mode || inb) ? off : IR2RGB_OOB;
    };

    unsigned poff[STATIC_TAPS ? PROWS : 1][NT];
    if constexpr (STATIC_TAPS) {
        // the tap grid is separable: fold / range-check the NTY row and NTX column coordinates once per
        // pixel row, then combine (NTY + NTX coordinate evaluations instead of NTY * NTX)
#pragma unroll
        for (int i = 0; i < PROWS; ++i) {
            int ry[NTY > 0 ? NTY : 1], rx[NTX > 0 ? NTX : 1];  // row * Win, column; -1 = outside (zero padding)
#pragma unroll
            for (int ty = 0; ty < NTY; ++ty) {
                int iy = psy[i] + g.dy0 + ty * g.dys;
                const bool in = (unsigned)iy < (unsigned)g.Hin;
                iy = g.pad_mode ? reflect(iy, g.Hin) : iy;
                ry[ty] = (g.pad_mode || in) ? iy * g.Win : -1;
            }
#pragma unroll
            for (int tx = 0; tx < NTX; ++tx) {
                int ix = psx[i] + g.dx0 + tx * g.dxs;
                const bool in = (unsigned)ix < (unsigned)g.Win;
                ix = g.pad_mode ? reflect(ix, g.Win) : ix;
                rx[tx] = (g.pad_mode || in) ? ix : -1;
            }
            const unsigned cbytes = (unsigned)(g.ci_off + chunk * 8) * 2u, ld2 = (unsigned)g.ldx * 2u;
#pragma unroll
            for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
                for (int tx = 0; tx < NTX; ++tx)
                    poff[i][ty * NTX + tx] = ((ry[ty] | rx[tx]) < 0) ? IR2RGB_OOB
                                                                      : (pbase[i] + (unsigned)(ry[ty] + rx[tx])) * ld2 + cbytes;
        }
    }

    // runtime-tap path: running (chunk, tap-row, tap-col) of the next K-step to be issued
    int i_cc = 0, i_ty = 0, i_tx = 0, i_t = 0;
    auto issue_weights = [&](int ks, unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < WROWS; ++i) lds_dma16(rw, woff[i], (unsigned)ks * 128u, dst + 8192 * i);
    };
    auto issue_dynamic = [&](int ks, int buf) {
        unsigned char *dst = wave_dst + buf * STAGE;
        issue_weights(ks, dst);
        const int dy = g.dy0 + i_ty * g.dys, dx = g.dx0 + i_tx * g.dxs;
#pragma unroll
        for (int i = 0; i < PROWS; ++i) lds_dma16(rx, gather_off(i, dy, dx), (unsigned)i_cc * 128u, dst + TC * 128 + 8192 * i);
        if (++i_tx == g.ntx) { i_tx = 0; ++i_ty; }
        if (++i_t == g.ntaps) { i_t = 0; i_ty = 0; i_tx = 0; ++i_cc; }
    };

    // ---------------- per-wave compute roles ----------------
    const int wc = wave & 1, wp = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int sw0 = (lq ^ (l15 >> 1)) << 4;  // byte offset of k-group lq      in a swizzled row
    const int sw1 = sw0 ^ 64;                // byte offset of k-group lq + 4
    const int offA = (wc * 64 + l15) * 128;                  // + mi * 2048
    const int offB = TC * 128 + (wp * (TP / 4) + l15) * 128; // + ni * 2048

    f32x4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.kchunks * g.ntaps;
    const bool late_dma = (g.variant & 1) && wave >= 4 && AHEAD >= 2;  // wave-uniform (SGPR)

    auto wait_stage = [&](int ks) {
        // stage ks has landed for THIS wave once all but the newest LOADS*(AHEAD-1) (later stages) are done.
        // lgkmcnt(0): this wave's fragment reads of stage ks-1 have RETURNED before it arrives at the barrier behind which
        // the other waves restage that buffer.  A raw s_barrier is no fence: without the wait hipcc sinks the last reads'
        // s_waitcnt (and the MFMAs that need them) below the barrier, and a read still in flight can lose against the
        // other waves' LDS-DMA when something slows the LDS -- it did, in the 2-stage ring, with a second stream's
        // bank-conflicted pack kernel resident on the same CU: rows 32..63 of a wave's weight fragment (read last) came
        // from the step after next.  tools/check_lds_war.py checks every kernel's code for this.
        if (AHEAD >= 2 && ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS * (AHEAD - 1)) : "memory");
        else                           asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... and for every wave; also: everyone is done reading stage ks-1
    };
    constexpr int MI = THIN ? 2 : 4;       // 16-row output-channel fragments a wave multiplies
    auto compute = [&](int buf) {
        if constexpr (THIN) {
            if (wc != 0) return;           // wave-uniform: rows 64..127 of the tile do not exist
        }
        const unsigned char *s = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            frag a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag *>(s + offA + mi * 2048 + sw);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const frag *>(s + offB + ni * 2048 + sw);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = H::mfma(a[mi], b[ni], acc[mi][ni]);
        }
    };

    if constexpr (STATIC_TAPS) {
        // K-step ks = cc*NT + t.  Prologue: stages 0 and 1.
        auto issue_static = [&](int ks, int cc, auto tconst, int buf) {
            constexpr int t = decltype(tconst)::value;
            unsigned char *dst = wave_dst + buf * STAGE;
            issue_weights(ks, dst);
#pragma unroll
            for (int i = 0; i < PROWS; ++i) lds_dma16(rx, poff[i][t], (unsigned)cc * 128u, dst + TC * 128 + 8192 * i);
        };
        issue_static(0, 0, std::integral_constant<int, 0>{}, 0);
        if (AHEAD >= 2 && nk > 1) issue_static(1, NT > 1 ? 0 : 1, std::integral_constant<int, (NT > 1 ? 1 : 0)>{}, 1);
        int ks = 0, buf = 0;
        for (int cc = 0; cc < g.kchunks; ++cc) {
            // unrolled over the taps: t is a compile-time constant inside
            auto body = [&](auto tconst) {
                constexpr int t = decltype(tconst)::value;
                wait_stage(ks);
                constexpr int t2 = (t + AHEAD) % NT;
                const int cc2 = cc + (t + AHEAD) / NT;
                int b2 = buf + AHEAD; b2 = b2 >= NSTAGE ? b2 - NSTAGE : b2;
                // SIMD partners (waves w, w+4) run complementary phases: the older half stages first and
                // multiplies second, the younger half the other way round.
                if (!late_dma && ks + AHEAD < nk) issue_static(ks + AHEAD, cc2, std::integral_constant<int, t2>{}, b2);
                compute(buf);
                if (late_dma && ks + AHEAD < nk) issue_static(ks + AHEAD, cc2, std::integral_constant<int, t2>{}, b2);
                ++ks;
                buf = buf + 1 == NSTAGE ? 0 : buf + 1;
            };
            [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (body(std::integral_constant<int, Ts>{}), ...); }
            (std::make_integer_sequence<int, NT>{});
        }
    } else {
        issue_dynamic(0, 0);
        if (AHEAD >= 2 && nk > 1) issue_dynamic(1, 1);
        int buf = 0;
        for (int ks = 0; ks < nk; ++ks) {
            wait_stage(ks);
            int b2 = buf + AHEAD; b2 = b2 >= NSTAGE ? b2 - NSTAGE : b2;
            if (!late_dma && ks + AHEAD < nk) issue_dynamic(ks + AHEAD, b2);
            compute(buf);
            if (late_dma && ks + AHEAD < nk) issue_dynamic(ks + AHEAD, b2);
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
    }

    // ---------------- epilogue: bias, activation, BN statistics, NHWC store ----------------
    // acc[mi][ni][r]: cout = ct*TC + wc*64 + mi*16 + lq*4 + r ; pixel = pt*TP + wp*(TP/4) + ni*16 + l15
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's last fragment reads have returned, see wait_stage)
    __builtin_amdgcn_s_barrier();  // all waves are past their last LDS read: smem is reusable
    const int co_base = ct * TC + wc * 64 + lq * 4;
    long opix[NI];
    bool oval[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        unsigned p = p0 + wp * (TP / 4) + ni * 16 + l15;
        oval[ni] = p < plim;
        p = oval[ni] ? p : 0u;
        unsigned n, rem;
        pix_decode(p, n, rem);
        unsigned sy = fdiv(rem, g.div_w), sx = rem - sy * g.Wsub;
        opix[ni] = ((long)n * g.Hout + (sy * g.s_out_y + g.off_y)) * g.Wout + (sx * g.s_out_x + g.off_x);
    }
    // Half outputs with Cout % 8 == 0 leave through LDS: each lane parks its 4-cout groups in a
    // [TP pixels][128 couts] half tile (16-byte slot s of row r at s ^ (r & 15): conflict-free 8-byte
    // writes), then the workgroup streams the tile out as 16-byte stores, 16 lanes per 256-byte pixel row
    // -- full-line coalescing instead of 8-byte scattered stores.  `red` (BatchNorm partials) sits behind it.
    const bool staged = !g.out_f32 && (g.Cout & 7) == 0 && (g.ldy & 7) == 0 && (g.co_off & 7) == 0;
    unsigned char *otile = smem;                                   // TP * 256 B
    float *red = reinterpret_cast<float *>(smem + TP * 256);       // [4 wp][TC][2]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int co = co_base + mi * 16;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (bias != nullptr && co + r < g.Cout) ? bias[co + r] : 0.f;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mi][ni][r] + bv[r];
                if (g.act) v[r] = v[r] > 0.f ? v[r] : (g.act == 1 ? 0.2f : (g.act == 2 ? 0.1f : 0.f)) * v[r];
                if (oval[ni]) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
            }
            if (staged) {
                const int prow = wp * (TP / 4) + ni * 16 + l15;      // pixel row in the tile
                const int cl = wc * 64 + mi * 16 + lq * 4;           // cout in the tile (multiple of 4)
                uint2 pk;
                pk.x = (uint32_t)H::cvt(v[0]) | ((uint32_t)H::cvt(v[1]) << 16);
                pk.y = (uint32_t)H::cvt(v[2]) | ((uint32_t)H::cvt(v[3]) << 16);
                *reinterpret_cast<uint2 *>(otile + prow * 256 + (((cl >> 3) ^ (prow & 15)) << 4) + (cl & 4) * 2) = pk;
            } else if (oval[ni] && g.out_f32) {
                float *dst = reinterpret_cast<float *>(Y) + opix[ni] * g.ldy + g.co_off + co;
                if (co + 3 < g.Cout && (g.Cout & 3) == 0) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < g.Cout) dst[r] = v[r];
                }
            } else if (oval[ni]) {
                uint16_t *dst = Y + opix[ni] * g.ldy + g.co_off + co;
                if (co + 3 < g.Cout && (g.Cout & 3) == 0) {
                    uint2 pk;
                    pk.x = (uint32_t)H::cvt(v[0]) | ((uint32_t)H::cvt(v[1]) << 16);
                    pk.y = (uint32_t)H::cvt(v[2]) | ((uint32_t)H::cvt(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(dst) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < g.Cout) dst[r] = H::cvt(v[r]);
                }
            }
        }
        if (stats_partial != nullptr) {
            // sum over the 16 pixels held by lanes l15 = 0..15 of this quarter-wave
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int cl = wc * 64 + mi * 16 + lq * 4 + r;  // cout within the tile
                    red[(wp * TC + cl) * 2 + 0] = s1[r];
                    red[(wp * TC + cl) * 2 + 1] = s2[r];
                }
            }
        }
    }
    if (staged) {
        __syncthreads();
        // 512 threads: 16 lanes per pixel row, 32 rows per pass
        const int c16 = tid & 15;
        const bool cok = ct * TC + c16 * 8 < g.Cout;   // Cout % 8 == 0: a 16-byte chunk is all-in or all-out
#pragma unroll
        for (int pass = 0; pass < TP / 32; ++pass) {
            const int prow = pass * 32 + (tid >> 4);
            unsigned p = p0 + prow;
            if (p < plim && cok) {
                unsigned n, rem;
                pix_decode(p, n, rem);
                unsigned sy = fdiv(rem, g.div_w), sx = rem - sy * g.Wsub;
                long op = ((long)n * g.Hout + (sy * g.s_out_y + g.off_y)) * g.Wout + (sx * g.s_out_x + g.off_x);
                const uint4 v = *reinterpret_cast<const uint4 *>(otile + prow * 256 + ((c16 ^ (prow & 15)) << 4));
                *reinterpret_cast<uint4 *>(Y + op * g.ldy + g.co_off + ct * TC + c16 * 8) = v;
            }
        }
    }
    if (stats_partial != nullptr) {
        __syncthreads();
        if (tid < TC * 2) {
            const int cl = tid >> 1, which = tid & 1, co = ct * TC + cl;
            if (co < g.Cout) {
                float t = red[(0 * TC + cl) * 2 + which] + red[(1 * TC + cl) * 2 + which] +
                          red[(2 * TC + cl) * 2 + which] + red[(3 * TC + cl) * 2 + which];
                stats_partial[((long)(g.stats_row0 + pt) * 2 + which) * g.Cout + co] = t;
            }
        }
    }
}

template <int DT, int TP, int NTY, int NTX, bool THIN = false>
__global__ void __launch_bounds__(512, (TP == 256 ? 2 : 4))
conv_igemm_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias,
                  uint16_t *__restrict__ Y, float *__restrict__ stats_partial, const ConvGeom g) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[ConvTile<TP>::LDS];
    conv_igemm_body<DT, TP, NTY, NTX, THIN>(X, Wp, bias, Y, stats_partial, g, (int)blockIdx.x, smem);
}

// Dot-product convolution for ONE fp32 output channel (the PatchGAN logit layers, 512 -> 1, 4x4, zero padding;
// NLayerDiscriminator's last layer, networks.py:697-699): 67x131 pixels x 8192 MACs are 0.14 GFLOP, but on the
// GEMM tile they cost 135 workgroups x 128 latency-bound K-steps (37 us).  Here a wave owns an output pixel:
// lane = 8 consecutive input channels of each 512-channel slab, the 16-byte activation and weight loads of eight
// taps in flight at once (unconditional from clamped coordinates, zero padding by masking), fp32 accumulation,
// butterfly reduction, lane 0 stores.  Reads the same packed weights as the GEMM kernels.
template <int DT>
__device__ __forceinline__ float dot8(const uint4 &a, const uint4 &b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0, x1, y0, y1;
        if (DT == IR2RGB_BF16) {
            x0 = __uint_as_float(x[j] << 16); x1 = __uint_as_float(x[j] & 0xffff0000u);
            y0 = __uint_as_float(y[j] << 16); y1 = __uint_as_float(y[j] & 0xffff0000u);
        } else {
            x0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(x[j] & 0xffff)); x1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(x[j] >> 16));
            y0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(y[j] & 0xffff)); y1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(y[j] >> 16));
        }
        s += x0 * y0;
        s += x1 * y1;
    }
    return s;
}

template <int DT>
__global__ void __launch_bounds__(256)
conv_dot_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias,
                float *__restrict__ Y, const ConvGeom g) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    const long P = (long)g.N * g.Hout * g.Wout;
    const int HW = g.Hout * g.Wout, slabs = g.Cin >> 9, nty = g.ntaps / g.ntx;
    const float b0 = bias ? bias[0] : 0.f;
    for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += nwaves) {
        const int n = (int)(p / HW), rem = (int)(p - (long)n * HW);
        const int oy = rem / g.Wout, ox = rem - oy * g.Wout;
        float acc = 0.f;
        for (int s = 0; s < slabs; ++s) {
            const uint16_t *wbase = Wp + ((long)(s * 8 + (lane >> 3)) * g.ntaps) * 64 + (lane & 7) * 8;
            for (int t0 = 0; t0 < g.ntaps; t0 += 8) {
                uint4 xv[8], wv[8];
                bool in[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = min(t0 + u, g.ntaps - 1);
                    const int ty = t / g.ntx, tx = t - ty * g.ntx;
                    int iy = oy * g.s_in_y + g.dy0 + ty * g.dys, ix = ox * g.s_in_x + g.dx0 + tx * g.dxs;
                    in[u] = t0 + u < g.ntaps && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                    iy = max(0, min(iy, g.Hin - 1));
                    ix = max(0, min(ix, g.Win - 1));
                    xv[u] = *reinterpret_cast<const uint4 *>(X + (((long)n * g.Hin + iy) * g.Win + ix) * g.ldx + g.ci_off + s * 512 + lane * 8);
                    wv[u] = *reinterpret_cast<const uint4 *>(wbase + (long)t * 64);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float d = dot8<DT>(xv[u], wv[u]);
                    acc += in[u] ? d : 0.f;
                }
            }
        }
        (void)nty;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) {
            float v = acc + b0;
            if (g.act) v = v > 0.f ? v : (g.act == 1 ? 0.2f : (g.act == 2 ? 0.1f : 0.f)) * v;
            Y[p * g.ldy + g.co_off] = v;
        }
    }
}

// All sub-pixel classes of a stride-2 transposed convolution (or of the data gradient of a stride-2
// convolution) in ONE launch.  Launched class by class, each class fills only part of the chip (e.g.
// 128 workgroups for 1024->512 @64x128) and the four launches run back to back; here the classes'
// workgroups share the grid, longest K loop first.  Every class starts at a multiple of 8 workgroups
// so the XCD of a workgroup is still (class-local index) & 7.
struct ConvClasses {
    ConvGeom g[4];
    long w_off[4];     // element offset of the class's packed weights
    int first[5];      // first workgroup of class c (multiples of 8); first[n] = grid size
    int n;
};

template <int DT, int TP>
__global__ void __launch_bounds__(512, (TP == 256 ? 2 : 4))
conv_igemm_classes_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp,
                          const float *__restrict__ bias, uint16_t *__restrict__ Y,
                          float *__restrict__ stats_partial, const ConvClasses cs) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[ConvTile<TP>::LDS];
    int c = 0;
    while (c + 1 < cs.n && (int)blockIdx.x >= cs.first[c + 1]) ++c;   // workgroup-uniform
    const ConvGeom &g = cs.g[c];
    const uint16_t *wp = Wp + cs.w_off[c];
    const int b = (int)blockIdx.x - cs.first[c];
    const int nty = g.ntaps / g.ntx;
    if (nty == 2 && g.ntx == 2)      conv_igemm_body<DT, TP, 2, 2>(X, wp, bias, Y, stats_partial, g, b, smem);
    else if (nty == 2 && g.ntx == 1) conv_igemm_body<DT, TP, 2, 1>(X, wp, bias, Y, stats_partial, g, b, smem);
    else if (nty == 1 && g.ntx == 2) conv_igemm_body<DT, TP, 1, 2>(X, wp, bias, Y, stats_partial, g, b, smem);
    else if (nty == 1 && g.ntx == 1) conv_igemm_body<DT, TP, 1, 1>(X, wp, bias, Y, stats_partial, g, b, smem);
    else                             conv_igemm_body<DT, TP, 0, 0>(X, wp, bias, Y, stats_partial, g, b, smem);
}

// ----------------------------------------------------------------------------------------
// weight packing: torch layout (fp32) -> Wp[class][Cout][Cin/64][ntaps][64] (half)
//   Conv2d          weight [Cout][Cin][kh][kw]
//   ConvTranspose2d weight [Cin][Cout][kh][kw]
// ----------------------------------------------------------------------------------------
struct PackGeom {
    int Cout, Cin, kh, kw, transposed;  // transposed: source tensor is [Cin][Cout][kh][kw]
    int flip;                           // read tap (kh-1-ky, kw-1-kx): adjoint (data-gradient) weights
    int ntaps;
    signed char ky[IR2RGB_MAX_TAPS], kx[IR2RGB_MAX_TAPS];
};

template <int DT>
__device__ __forceinline__ void pack_weight_body(const float *__restrict__ w, uint16_t *__restrict__ wp, const PackGeom &g,
                                                 long total, long first, long stride) {
    const int kchunks = g.Cin / 64;
    for (long i = first; i < total; i += stride) {
        long r = i;
        int c64 = (int)(r & 63); r >>= 6;
        int tap = (int)(r % g.ntaps); r /= g.ntaps;
        int cc = (int)(r % kchunks);
        int co = (int)(r / kchunks);
        int ci = cc * 64 + c64;
        const int ky = g.flip ? g.kh - 1 - g.ky[tap] : g.ky[tap], kx = g.flip ? g.kw - 1 - g.kx[tap] : g.kx[tap];
        long src = g.transposed ? (((long)ci * g.Cout + co) * g.kh + ky) * g.kw + kx
                                : (((long)co * g.Cin + ci) * g.kh + ky) * g.kw + kx;
        wp[i] = Half<DT>::cvt(w[src]);
    }
}

template <int DT>
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float *__restrict__ w, uint16_t *__restrict__ wp, const PackGeom g, long total) {
    pack_weight_body<DT>(w, wp, g, total, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Tiled packing of a plain Conv2d weight [Cout][Cin][T] (T = kh*kw <= 9 taps in row-major order, one
// class).  A block stages a TCO x TCI x T tile through LDS so that both the fp32 reads (TCI*T contiguous
// floats per co) and the half writes (64 contiguous halfs per (row, tap)) are coalesced; the tile is
// long (64) along the axis that is contiguous in the OUTPUT and short (16) along the other, which
// gives 1024 blocks of 18 KB for a 1024 x 1024 weight (several per CU: loads of one overlap stores of
// another).
//   ADJ = 0 (16 co x 64 ci):  Wp[co][ci/64][t][ci%64]       = w[co][ci][t]   (forward operand)
//   ADJ = 1 (64 co x 16 ci):  Wp[ci][co/64][T-1-t][co%64]   = w[co][ci][t]   (data-gradient operand:
//             roles of the channel axes swapped, taps flipped)
#define PACK_MAX_T 9
#define PACK_TILE_HALFS (64 * (16 * PACK_MAX_T + 4))     // the larger of the two tile shapes
template <int DT, int ADJ>
__device__ __forceinline__ void pack_tile_body(const float *__restrict__ w, uint16_t *__restrict__ wp, int Cout, int Cin,
                                               int T, int bx, int by, uint16_t *tile) {
    constexpr int TCO = ADJ ? 64 : 16, TCI = ADJ ? 16 : 64;
    static_assert(TCO * (TCI * PACK_MAX_T + 4) <= PACK_TILE_HALFS, "tile buffer");
    const int row = TCI * T + 4;                          // halfs per co row (+4: spreads the banks)
    const int co0 = by * TCO, ci0 = bx * TCI;
    const int q_per_row = TCI * T / 4;                    // float4 per co row
    for (int i = threadIdx.x; i < TCO * q_per_row; i += 256) {
        const int co_l = i / q_per_row, q = i - co_l * q_per_row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co0 + co_l < Cout) v = *reinterpret_cast<const float4 *>(w + ((long)(co0 + co_l) * Cin + ci0) * T + 4 * q);
        uint2 h;
        h.x = (uint32_t)Half<DT>::cvt(v.x) | ((uint32_t)Half<DT>::cvt(v.y) << 16);
        h.y = (uint32_t)Half<DT>::cvt(v.z) | ((uint32_t)Half<DT>::cvt(v.w) << 16);
        *reinterpret_cast<uint2 *>(tile + co_l * row + 4 * q) = h;     // element ci_l*T + t of the row
    }
    __syncthreads();
    if (ADJ) {
        // rows r = ci_l (16), each (r, t) is 64 consecutive co: 8 lanes x 16 B
        const int kch = Cout / 64;
        for (int i = threadIdx.x; i < TCI * T * 8; i += 256) {
            const int o8 = i & 7, rt = i >> 3;
            const int r = rt / T, t = rt - r * T;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = tile[(o8 * 8 + 2 * j) * row + r * T + t];
                const uint32_t hi = tile[(o8 * 8 + 2 * j + 1) * row + r * T + t];
                o[j] = lo | (hi << 16);
            }
            const long dst = ((((long)(ci0 + r) * kch + by) * T + (T - 1 - t)) * 64 + o8 * 8);
            *reinterpret_cast<uint4 *>(wp + dst) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else {
        // rows r = co_l (16), each (r, t) is 64 consecutive ci: 8 lanes x 16 B
        const int kch = Cin / 64;
        for (int i = threadIdx.x; i < TCO * T * 8; i += 256) {
            const int o8 = i & 7, rt = i >> 3;
            const int r = rt / T, t = rt - r * T;
            if (co0 + r >= Cout) continue;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = tile[r * row + (o8 * 8 + 2 * j) * T + t];
                const uint32_t hi = tile[r * row + (o8 * 8 + 2 * j + 1) * T + t];
                o[j] = lo | (hi << 16);
            }
            const long dst = ((((long)(co0 + r) * kch + bx) * T + t) * 64 + o8 * 8);
            *reinterpret_cast<uint4 *>(wp + dst) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <int DT, int ADJ>
__global__ void __launch_bounds__(256)
pack_tile_kernel(const float *__restrict__ w, uint16_t *__restrict__ wp, int Cout, int Cin, int T) {
    __shared__ uint16_t tile[PACK_TILE_HALFS];
    pack_tile_body<DT, ADJ>(w, wp, Cout, Cin, T, blockIdx.x, blockIdx.y, tile);
}

// Batched packing: every weight of a network in ONE launch.  After an optimizer step ~280 packed copies
// (forward and data-gradient operands) are stale; most are small and their individual launches cost
// more than their bytes.  The table (device memory, built once by ir2rgb_conv2d_pack_batch_build) holds
// one PackEntry per (weight, class) followed by one PackBlock per workgroup: a block's whole job
// description is one uniform 48-byte load (scalar loads), no search.
struct PackEntry {
    const float *w;
    uint16_t *wp;
    long total;                 // kind 0: elements of this class
    int kind;                   // 0 = generic gather, 1 = tile forward, 2 = tile adjoint
    int first_block, nblocks;
    int Cout, Cin, T, gx;       // kinds 1/2: source dims, taps, blocks along ci
    PackGeom g;                 // kind 0
};
struct PackBlock {
    const float *w;
    uint16_t *wp;
    int Cout, Cin, T, kind;     // kinds 1/2
    int bx, by;                 // kinds 1/2: tile coordinates; kind 0: block index within the entry, blocks of the entry
    int entry, pad_;
};
static inline size_t pack_blocks_offset(int nentries) { return ((size_t)nentries * sizeof(PackEntry) + 15) & ~(size_t)15; }

template <int DT>
__global__ void __launch_bounds__(256)
pack_batch_kernel(const PackEntry *__restrict__ E, const PackBlock *__restrict__ B) {
    __shared__ uint16_t tile[PACK_TILE_HALFS];
    const PackBlock b = B[blockIdx.x];
    if (b.kind == 0) {
        const PackEntry &e = E[b.entry];
        pack_weight_body<DT>(b.w, b.wp, e.g, e.total, b.bx * 256L + threadIdx.x, b.by * 256L);
    } else if (b.kind == 1) {
        pack_tile_body<DT, 0>(b.w, b.wp, b.Cout, b.Cin, b.T, b.bx, b.by, tile);
    } else {
        pack_tile_body<DT, 1>(b.w, b.wp, b.Cout, b.Cin, b.T, b.bx, b.by, tile);
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
struct ClassPlan {
    ConvGeom geom;
    PackGeom pack;
    long w_offset;  // element offset of this class in the packed weight buffer
    int npt;        // pixel tiles (TP chosen at plan time)
    int tp;         // pixels per tile: 256 / 128 / 64
};

static bool merge_classes() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV_MERGE"); v = e ? atoi(e) : 1; }
    return v != 0;
}

static int conv_variant() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("IR2RGB_CONV_VARIANT");
        v = e ? atoi(e) : 1;  // default: staggered DMA (measured +6..7 % on the large shapes)
    }
    return v;
}

static int tile_pixels(long P, int Cout, int ksteps) {
    // TP = 256 (one resident workgroup per CU) pays off for long K loops with at least one tile per CU;
    // shorter loops / fewer tiles run two workgroups per CU with TP = 128 or 64.
    static int forced = -1;          // IR2RGB_CONV_TP=256|128|64: experiments only
    if (forced < 0) { const char *e = getenv("IR2RGB_CONV_TP"); forced = e ? atoi(e) : 0; }
    if (forced == 256 || forced == 128 || forced == 64) return forced;
    const long nct = (Cout + 127) / 128;
    const long tiles256 = ((P + 255) / 256) * nct;
    static int rule = -1;            // IR2RGB_CONV_TP_RULE=0: the earlier rule (TP = 256 also for short loops over >= 768 tiles)
    if (rule < 0) { const char *e = getenv("IR2RGB_CONV_TP_RULE"); rule = e ? atoi(e) : 1; }
    // long K loops: the deep 3-stage ring of the 256-pixel tile wins; short ones (k x 1 first layers and heads,
    // sub-pixel classes of the up-samplers: 7..36 steps) run 10-20 % faster as two 128-pixel workgroups per CU
    if (tiles256 >= 256 && (ksteps >= 40 || (rule == 0 && tiles256 >= 768))) return 256;
    if (((P + 127) / 128) * nct >= 256) return 128;
    return 64;
}

// Builds the launch plan(s) for one convolution; returns the number of classes or < 0.
static int make_plan(const ir2rgb_conv_desc *d, ClassPlan plans[4]) {
    if (!d || d->N < 1 || d->Cin < 64 || (d->Cin % 64) || d->Cout < 1 || d->kh < 1 || d->kw < 1 ||
        d->kh * d->kw > IR2RGB_MAX_TAPS || d->stride_h < 1 || d->stride_w < 1 || d->pad_h < 0 || d->pad_w < 0)
        return IR2RGB_EINVAL;
    if (d->dtype != IR2RGB_BF16 && d->dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (d->pad_mode != 0 && d->pad_mode != 1) return IR2RGB_ENOSUP;   // pad_mode 2 exists in conv3x3_patch_kernel only
    int ncls = 0;
    long woff = 0;
    int row0 = 0;
    if (!d->transposed) {
        int Ho = (d->Hin + 2 * d->pad_h - d->kh) / d->stride_h + 1, Wo = (d->Win + 2 * d->pad_w - d->kw) / d->stride_w + 1;
        if (Ho != d->Hout || Wo != d->Wout || Ho < 1 || Wo < 1) return IR2RGB_EINVAL;
        if (d->pad_mode == 1 && (d->pad_h >= d->Hin || d->pad_w >= d->Win)) return IR2RGB_EINVAL;
        ClassPlan &c = plans[0];
        ConvGeom &g = c.geom;
        g = ConvGeom{};
        g.N = d->N; g.Hin = d->Hin; g.Win = d->Win; g.Cin = d->Cin;
        g.Hsub = Ho; g.Wsub = Wo; g.Hout = Ho; g.Wout = Wo; g.Cout = d->Cout;
        g.s_in_y = d->stride_h; g.s_in_x = d->stride_w; g.s_out_y = g.s_out_x = 1; g.off_y = g.off_x = 0;
        g.ntaps = d->kh * d->kw; g.pad_mode = d->pad_mode; g.kchunks = d->Cin / 64; g.act = d->act;
        g.out_f32 = d->out_f32;
        c.pack = PackGeom{};
        c.pack.Cout = d->Cout; c.pack.Cin = d->Cin; c.pack.kh = d->kh; c.pack.kw = d->kw; c.pack.transposed = 0;
        c.pack.ntaps = g.ntaps;
        g.ntx = d->kw; g.dy0 = -d->pad_h; g.dys = 1; g.dx0 = -d->pad_w; g.dxs = 1;
        for (int ky = 0, t = 0; ky < d->kh; ++ky)
            for (int kx = 0; kx < d->kw; ++kx, ++t) {
                c.pack.ky[t] = (signed char)ky; c.pack.kx[t] = (signed char)kx;
            }
        c.w_offset = 0;
        ncls = 1;
    } else {
        // transposed convolution (stride 1 or 2 per axis) as stride_h*stride_w sub-pixel classes:
        //   oy = sh*iy - pad + ky  ->  for oy = sh*sy + a: ky = (a + pad) mod sh (+ sh*j), iy = sy + (a + pad - ky)/sh
        const int sh = d->stride_h, sw = d->stride_w;
        if (sh > 2 || sw > 2 || d->pad_mode != 0) return IR2RGB_ENOSUP;
        int Hfull = (d->Hin - 1) * sh - 2 * d->pad_h + d->kh, Wfull = (d->Win - 1) * sw - 2 * d->pad_w + d->kw;
        if (d->Hout < Hfull || d->Hout > Hfull + sh - 1 || d->Wout < Wfull || d->Wout > Wfull + sw - 1 || d->Hout < 1 ||
            d->Wout < 1)
            return IR2RGB_EINVAL;
        for (int a = 0; a < sh; ++a)
            for (int b = 0; b < sw; ++b) {
                const int Hsub = (d->Hout - a + sh - 1) / sh, Wsub = (d->Wout - b + sw - 1) / sw;
                if (Hsub < 1 || Wsub < 1) continue;
                ClassPlan &c = plans[ncls];
                ConvGeom &g = c.geom;
                g = ConvGeom{};
                g.N = d->N; g.Hin = d->Hin; g.Win = d->Win; g.Cin = d->Cin;
                g.Hsub = Hsub; g.Wsub = Wsub; g.Hout = d->Hout; g.Wout = d->Wout; g.Cout = d->Cout;
                g.s_in_y = g.s_in_x = 1; g.s_out_y = sh; g.s_out_x = sw; g.off_y = a; g.off_x = b;
                g.pad_mode = 0; g.kchunks = d->Cin / 64; g.act = d->act; g.out_f32 = d->out_f32;
                c.pack = PackGeom{};
                c.pack.Cout = d->Cout; c.pack.Cin = d->Cin; c.pack.kh = d->kh; c.pack.kw = d->kw; c.pack.transposed = 1;
                int t = 0, ntx = 0;
                const int ky0 = (a + d->pad_h) % sh, kx0 = (b + d->pad_w) % sw;
                for (int ky = ky0; ky < d->kh; ky += sh) {
                    ntx = 0;
                    for (int kx = kx0; kx < d->kw; kx += sw, ++t, ++ntx) {
                        c.pack.ky[t] = (signed char)ky; c.pack.kx[t] = (signed char)kx;
                    }
                }
                if (t == 0) return IR2RGB_ENOSUP;  // a class without taps would need a bias-only fill
                g.ntx = ntx; g.dy0 = (a + d->pad_h - ky0) / sh; g.dys = -1; g.dx0 = (b + d->pad_w - kx0) / sw; g.dxs = -1;
                g.ntaps = t; c.pack.ntaps = t;
                c.w_offset = woff;
                woff += (long)d->Cout * d->Cin * t;
                ++ncls;
            }
    }
    if ((long)d->N * d->Hin * d->Win >= (1L << 31) || (long)d->N * d->Hout * d->Wout >= (1L << 31) ||
        (long)d->N * d->Hin * d->Win * ((d->ldx > 0 ? d->ldx : d->Cin) / 8) >= 0xFFFFFFFFL)
        return IR2RGB_EINVAL;
    for (int i = 0; i < ncls; ++i) {
        ConvGeom &g = plans[i].geom;
        g.variant = conv_variant();
        g.div_hw = make_fastdiv((unsigned)(g.Hsub * g.Wsub)); g.div_w = make_fastdiv((unsigned)g.Wsub);
        g.ldx = d->ldx > 0 ? d->ldx : d->Cin; g.ci_off = d->ci_off;
        g.ldy = d->ldy > 0 ? d->ldy : d->Cout; g.co_off = d->co_off;
        if (g.ci_off < 0 || g.co_off < 0 || g.ci_off + d->Cin > g.ldx || g.co_off + d->Cout > g.ldy || (g.ldx & 7) || (g.ci_off & 7))
            return IR2RGB_EINVAL;
        // the epilogue's 4-channel vector stores (taken when Cout % 4 == 0) need 4-channel aligned rows
        if ((d->Cout & 3) == 0 && ((g.ldy & 3) || (g.co_off & 3))) return IR2RGB_EINVAL;
        const long xb = (long)g.N * g.Hin * g.Win * g.ldx * 2, wb = (long)g.Cout * g.Cin * g.ntaps * 2;
        if (xb >= (1L << 31) || wb >= (1L << 31)) return IR2RGB_EINVAL;  // 32-bit buffer offsets
        g.x_bytes = (unsigned)xb; g.w_bytes = (unsigned)wb;
        {
            static int force = -2;
            if (force == -2) { const char *e = getenv("IR2RGB_CONV_COUT_MAJOR"); force = e ? atoi(e) : -1; }
            const long x_used = (long)g.N * g.Hin * g.Win * g.Cin * 2;
            g.cout_major = force >= 0 ? force : (wb > x_used ? 1 : 0);
        }
    }
    // pixel-tile size: per class, or -- when the classes share one launch -- one size for all of them chosen
    // from the total tile count and the longest K loop
    int tp_all = 0;
    if (ncls > 1 && merge_classes()) {
        long p_all = 0, ks_sum = 0;
        for (int i = 0; i < ncls; ++i) {
            const ConvGeom &g = plans[i].geom;
            p_all += (long)g.N * g.Hsub * g.Wsub;
            ks_sum += (long)g.kchunks * g.ntaps;
        }
        tp_all = tile_pixels(p_all, plans[0].geom.Cout, (int)(ks_sum / ncls));     // mean K steps over the classes
        // Classes have K loops of different lengths (3x3 / stride 2: 1, 2, 2 and 4 taps) and a workgroup keeps its class
        // to the end: with 128-pixel tiles and at most ~one round of workgroups (the training sizes: 1024 -> 512 @32x64
        // has 256 of them) every CU runs ONE workgroup and the launch lasts as long as the 4-tap class.  64-pixel tiles give
        // two co-resident workgroups per CU, long and short ones mixed.  IR2RGB_CONV_CLASSES_SMALL=0: the earlier rule.
        static int small = -1;
        if (small < 0) { const char *e = getenv("IR2RGB_CONV_CLASSES_SMALL"); small = e ? atoi(e) : 1; }
        const long wg128 = ((p_all + 127) / 128) * ((plans[0].geom.Cout + 127) / 128);
        if (small && tp_all == 128 && wg128 <= 320) tp_all = 64;
    }
    for (int i = 0; i < ncls; ++i) {
        ConvGeom &g = plans[i].geom;
        long P = (long)g.N * g.Hsub * g.Wsub;
        int tp = tp_all ? tp_all : tile_pixels(P, g.Cout, g.kchunks * g.ntaps);
        plans[i].tp = tp;
        plans[i].npt = (int)((P + tp - 1) / tp);
        if (d->stats_per_sample) {          // pixel tiles cut per sample: a statistics row never mixes samples
            if (ncls > 1) return IR2RGB_ENOSUP;
            g.tpi = (int)(((long)g.Hsub * g.Wsub + tp - 1) / tp);
            plans[i].npt = g.N * g.tpi;
        }
        g.stats_row0 = row0;
        row0 += plans[i].npt;
    }
    return ncls;
}

extern "C" long ir2rgb_conv2d_packed_weight_elems(const ir2rgb_conv_desc *d) {
    ClassPlan plans[4];
    int n = make_plan(d, plans);
    if (n < 0) return n;
    long e = 0;
    for (int i = 0; i < n; ++i) e += (long)d->Cout * d->Cin * plans[i].geom.ntaps;
    return e;
}

// conv3x3_patch.hip: the patch-staged kernel for 3x3 / stride-1 layers with >= 256 input channels
struct P3Geom {
    int N, H, W, Ho, Wo, Cin, Cout;
    int pad, pad_mode, act;
    int ldx, ci_off, ldy, co_off;
    int stats_row0, nty, ntx;
    int cout_major, kchunks;
    unsigned x_bytes, w_bytes;
    int dbg;
};
int conv3x3p_plan(const ir2rgb_conv_desc *d, P3Geom *g, int *npt_out, bool allow_split = false);
long conv3x3p_workspace_bytes(int variant, const P3Geom &g);
int conv3x3p_launch(int variant, const P3Geom &g, int dtype, const void *x, const void *wp, const float *bias, void *y,
                    float *stats, hipStream_t s, void *workspace, long workspace_bytes);

// One fp32 output channel, zero padding, 512-channel slabs, <= 16 taps: conv_dot_kernel (IR2RGB_CONV_DOT=0: GEMM tile)
static bool conv_dot_ok(const ir2rgb_conv_desc *d) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV_DOT"); v = e ? atoi(e) : 1; }
    const int ldx = d->ldx ? d->ldx : d->Cin;
    return v != 0 && d->out_f32 && d->Cout == 1 && !d->transposed && d->pad_mode == 0 && d->Cin >= 512 && (d->Cin % 512) == 0 &&
           d->kh * d->kw <= 16 && (ldx % 8) == 0 && (d->ci_off % 8) == 0;
}

extern "C" const char *ir2rgb_conv2d_kernel_name(const ir2rgb_conv_desc *d) {
    if (!d) return "";
    P3Geom g3;
    int npt3 = 0;
    if (conv3x3p_plan(d, &g3, &npt3)) return "conv3x3_patch_kernel";
    {
        T7Geom g7;
        if (conv1x7_thin_plan(d, &g7)) return "conv1x7_thin_kernel";     // (launches without bias / statistics: the head pass)
        C7Geom gc;
        if (conv7x1_col_plan(d, &gc)) return "conv7x1_col_kernel";
    }
    ClassPlan plans[4];
    const int n = make_plan(d, plans);
    if (n < 0) return "";
    if (n == 1 && conv_dot_ok(d)) return "conv_dot_kernel";
    return (n > 1 && merge_classes()) ? "conv_igemm_classes_kernel" : "conv_igemm_kernel";
}

extern "C" int ir2rgb_conv2d_stats_rows(const ir2rgb_conv_desc *d) {
    {
        P3Geom g3;
        int npt3 = 0;
        if (d && conv3x3p_plan(d, &g3, &npt3)) return npt3;
        C7Geom gc;
        if (d && conv7x1_col_plan(d, &gc)) return conv7x1_col_tiles(gc);
    }
    ClassPlan plans[4];
    int n = make_plan(d, plans);
    if (n < 0) return n;
    int rows = 0;
    for (int i = 0; i < n; ++i) rows += plans[i].npt;
    return rows;
}

extern "C" int ir2rgb_conv2d_pack_weight_adjoint(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream);

static int pack_impl(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream, bool adjoint) {
    ClassPlan plans[4];
    int n = make_plan(d, plans);
    if (n < 0) return n;
    if (adjoint && d->transposed) return IR2RGB_ENOSUP;
    // fast path: plain convolution weight, one class, all taps in row-major order, <= 9 taps
    {
        const PackGeom &pg = plans[0].pack;
        const int T = d->kh * d->kw;
        bool natural = n == 1 && !d->transposed && !pg.transposed && !pg.flip && pg.ntaps == T && T <= PACK_MAX_T;
        for (int t = 0; natural && t < T; ++t) natural = pg.ky[t] == t / d->kw && pg.kx[t] == t % d->kw;
        // the packed buffer of the adjoint descriptor d (Cin' = d->Cin plays the source's Cout axis)
        const int srcCout = adjoint ? d->Cin : d->Cout, srcCin = adjoint ? d->Cout : d->Cin;
        if (natural && srcCin % 64 == 0 && (!adjoint || srcCout % 64 == 0) && (((uintptr_t)w | (uintptr_t)wpacked) & 15) == 0) {
            const int tco = adjoint ? 64 : 16, tci = adjoint ? 16 : 64;
            dim3 grid((unsigned)(srcCin / tci), (unsigned)((srcCout + tco - 1) / tco));
            uint16_t *dst = reinterpret_cast<uint16_t *>(wpacked);
            hipStream_t s = as_stream(stream);
            if (d->dtype == IR2RGB_BF16) {
                if (adjoint) pack_tile_kernel<IR2RGB_BF16, 1><<<grid, 256, 0, s>>>(w, dst, srcCout, srcCin, T);
                else pack_tile_kernel<IR2RGB_BF16, 0><<<grid, 256, 0, s>>>(w, dst, srcCout, srcCin, T);
            } else {
                if (adjoint) pack_tile_kernel<IR2RGB_F16, 1><<<grid, 256, 0, s>>>(w, dst, srcCout, srcCin, T);
                else pack_tile_kernel<IR2RGB_F16, 0><<<grid, 256, 0, s>>>(w, dst, srcCout, srcCin, T);
            }
            return ir2rgb_launch_status();
        }
    }
    for (int i = 0; i < n; ++i) {
        if (adjoint) { plans[i].pack.transposed = 1; plans[i].pack.flip = 1; }
        long total = (long)d->Cout * d->Cin * plans[i].geom.ntaps;
        uint16_t *dst = reinterpret_cast<uint16_t *>(wpacked) + plans[i].w_offset;
        int grid = stream_grid(total, 256);
        if (d->dtype == IR2RGB_BF16)
            pack_weight_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>(w, dst, plans[i].pack, total);
        else
            pack_weight_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>(w, dst, plans[i].pack, total);
    }
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_conv2d_pack_weight(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream) {
    return pack_impl(d, w, wpacked, stream, false);
}

// The entries pack_impl's launches correspond to (same decisions, same kernels' bodies).
static int pack_entries(const ir2rgb_conv_desc *d, const float *w, void *wpacked, bool adjoint, PackEntry *out) {
    ClassPlan plans[4];
    int n = make_plan(d, plans);
    if (n < 0) return n;
    if (adjoint && d->transposed) return IR2RGB_ENOSUP;
    if (!w || !wpacked) return IR2RGB_EINVAL;
    const PackGeom &pg = plans[0].pack;
    const int T = d->kh * d->kw;
    bool natural = n == 1 && !d->transposed && !pg.transposed && !pg.flip && pg.ntaps == T && T <= PACK_MAX_T;
    for (int t = 0; natural && t < T; ++t) natural = pg.ky[t] == t / d->kw && pg.kx[t] == t % d->kw;
    const int srcCout = adjoint ? d->Cin : d->Cout, srcCin = adjoint ? d->Cout : d->Cin;
    if (natural && srcCin % 64 == 0 && (!adjoint || srcCout % 64 == 0) && (((uintptr_t)w | (uintptr_t)wpacked) & 15) == 0) {
        const int tco = adjoint ? 64 : 16, tci = adjoint ? 16 : 64;
        PackEntry &e = out[0];
        memset(&e, 0, sizeof(e));
        e.w = w;
        e.wp = reinterpret_cast<uint16_t *>(wpacked);
        e.kind = adjoint ? 2 : 1;
        e.Cout = srcCout; e.Cin = srcCin; e.T = T;
        e.gx = srcCin / tci;
        e.nblocks = e.gx * ((srcCout + tco - 1) / tco);
        return 1;
    }
    for (int i = 0; i < n; ++i) {
        PackEntry &e = out[i];
        memset(&e, 0, sizeof(e));
        e.g = plans[i].pack;
        if (adjoint) { e.g.transposed = 1; e.g.flip = 1; }
        e.total = (long)d->Cout * d->Cin * plans[i].geom.ntaps;
        e.w = w;
        e.wp = reinterpret_cast<uint16_t *>(wpacked) + plans[i].w_offset;
        e.kind = 0;
        long nb = (e.total + 1023) / 1024;                     // four elements per thread
        e.nblocks = (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
    }
    return n;
}

static int pack_batch_expand(const ir2rgb_pack_job *jobs, int njobs, PackEntry *E, int max_entries, long *nblocks) {
    if (!jobs || njobs < 1) return IR2RGB_EINVAL;
    int n = 0;
    long blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        if (jobs[j].desc.dtype != jobs[0].desc.dtype) return IR2RGB_EINVAL;    // one element type per launch
        PackEntry tmp[4];
        const int k = pack_entries(&jobs[j].desc, (const float *)jobs[j].w, jobs[j].wpacked, jobs[j].adjoint != 0, tmp);
        if (k < 0) return k;
        for (int i = 0; i < k; ++i) {
            tmp[i].first_block = (int)blocks;
            blocks += tmp[i].nblocks;
            if (E) {
                if (n >= max_entries) return IR2RGB_EINVAL;
                E[n] = tmp[i];
            }
            ++n;
        }
    }
    if (blocks > 0x7fffffffL) return IR2RGB_EINVAL;
    *nblocks = blocks;
    return n;
}

extern "C" long ir2rgb_conv2d_pack_batch_table_bytes(const ir2rgb_pack_job *jobs, int njobs) {
    long blocks = 0;
    const int n = pack_batch_expand(jobs, njobs, nullptr, 0, &blocks);
    if (n < 0) return n;
    return (long)(pack_blocks_offset(n) + (size_t)blocks * sizeof(PackBlock));
}

extern "C" int ir2rgb_conv2d_pack_batch_build(const ir2rgb_pack_job *jobs, int njobs, void *table_host, long table_bytes,
                                              int *nblocks) {
    if (!table_host || !nblocks) return IR2RGB_EINVAL;
    const long need = ir2rgb_conv2d_pack_batch_table_bytes(jobs, njobs);
    if (need < 0) return (int)need;
    if (table_bytes < need) return IR2RGB_EINVAL;
    PackEntry *E = reinterpret_cast<PackEntry *>(table_host);
    long blocks = 0;
    const int n = pack_batch_expand(jobs, njobs, E, (int)(table_bytes / (long)sizeof(PackEntry)), &blocks);
    if (n < 0) return n;
    PackBlock *B = reinterpret_cast<PackBlock *>(reinterpret_cast<char *>(table_host) + pack_blocks_offset(n));
    for (int i = 0; i < n; ++i) {
        const PackEntry &e = E[i];
        for (int lb = 0; lb < e.nblocks; ++lb) {
            PackBlock &b = B[e.first_block + lb];
            memset(&b, 0, sizeof(b));
            b.w = e.w; b.wp = e.wp; b.Cout = e.Cout; b.Cin = e.Cin; b.T = e.T; b.kind = e.kind; b.entry = i;
            if (e.kind == 0) { b.bx = lb; b.by = e.nblocks; }
            else { b.bx = lb % e.gx; b.by = lb / e.gx; }
        }
    }
    *nblocks = (int)blocks;
    return n;
}

extern "C" int ir2rgb_conv2d_pack_batch_run(const void *table_dev, int nentries, int nblocks, int dtype, void *stream) {
    if (!table_dev || nentries < 1 || nblocks < 1) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if ((uintptr_t)table_dev & 15) return IR2RGB_EALIGN;
    const PackEntry *E = reinterpret_cast<const PackEntry *>(table_dev);
    const PackBlock *B = reinterpret_cast<const PackBlock *>(reinterpret_cast<const char *>(table_dev) + pack_blocks_offset(nentries));
    if (dtype == IR2RGB_BF16) pack_batch_kernel<IR2RGB_BF16><<<nblocks, 256, 0, as_stream(stream)>>>(E, B);
    else pack_batch_kernel<IR2RGB_F16><<<nblocks, 256, 0, as_stream(stream)>>>(E, B);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_conv2d_pack_weight_adjoint(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream) {
    return pack_impl(d, w, wpacked, stream, true);
}

template <int DT, int NTY, int NTX, bool THIN = false>
static void launch_conv_taps(const ClassPlan &c, int tp, unsigned grid, const uint16_t *x, const uint16_t *wp,
                             const float *bias, uint16_t *y, float *stats, hipStream_t s) {
    const ConvGeom &g = c.geom;
    switch (tp) {
        case 256: conv_igemm_kernel<DT, 256, NTY, NTX, THIN><<<grid, 512, 0, s>>>(x, wp, bias, y, stats, g); break;
        case 128: conv_igemm_kernel<DT, 128, NTY, NTX, THIN><<<grid, 512, 0, s>>>(x, wp, bias, y, stats, g); break;
        default:  conv_igemm_kernel<DT, 64, NTY, NTX, THIN><<<grid, 512, 0, s>>>(x, wp, bias, y, stats, g); break;
    }
}

static bool conv_thin() {      // IR2RGB_CONV_THIN=0: thin layers on the full 128-row tile (A/B measurements)
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV_THIN"); v = e ? atoi(e) : 1; }
    return v != 0;
}

template <int DT>
static void launch_conv(const ClassPlan &c, const uint16_t *x, const uint16_t *wp, const float *bias, uint16_t *y,
                        float *stats, hipStream_t s) {
    const ConvGeom &g = c.geom;
    const int tp = c.tp;
    const int nct = (g.Cout + 127) / 128;
    const unsigned grid = (unsigned)(c.npt * nct);
    const int nty = g.ntaps / g.ntx;
    const bool thin = g.Cout <= 32 && conv_thin();
    if (thin && nty == 4 && g.ntx == 4)      launch_conv_taps<DT, 4, 4, true>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (thin && nty == 1 && g.ntx == 7) launch_conv_taps<DT, 1, 7, true>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 3 && g.ntx == 3) launch_conv_taps<DT, 3, 3>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 4 && g.ntx == 4) launch_conv_taps<DT, 4, 4>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 7 && g.ntx == 1) launch_conv_taps<DT, 7, 1>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 1 && g.ntx == 7) launch_conv_taps<DT, 1, 7>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 4 && g.ntx == 1) launch_conv_taps<DT, 4, 1>(c, tp, grid, x, wp, bias, y, stats, s);
    // sub-pixel classes of the stride-2 transposed convolutions (3x3: 1x1, 1x2, 2x1, 2x2 taps; 4x4: 2x2)
    else if (nty == 2 && g.ntx == 2) launch_conv_taps<DT, 2, 2>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 2 && g.ntx == 1) launch_conv_taps<DT, 2, 1>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 1 && g.ntx == 2) launch_conv_taps<DT, 1, 2>(c, tp, grid, x, wp, bias, y, stats, s);
    else if (nty == 1 && g.ntx == 1) launch_conv_taps<DT, 1, 1>(c, tp, grid, x, wp, bias, y, stats, s);
    else                             launch_conv_taps<DT, 0, 0>(c, tp, grid, x, wp, bias, y, stats, s);
}

extern "C" long ir2rgb_conv2d_fwd_workspace_bytes(const ir2rgb_conv_desc *d) {
    if (!d) return IR2RGB_EINVAL;
    P3Geom g3;
    int npt3 = 0;
    return conv3x3p_workspace_bytes(conv3x3p_plan(d, &g3, &npt3, true), g3);
}

extern "C" int ir2rgb_conv2d_fwd_ws(const ir2rgb_conv_desc *d, const void *x, const void *wpacked, const float *bias,
                                    void *y, float *stats_partial, void *workspace, long workspace_bytes, void *stream);

extern "C" int ir2rgb_conv2d_fwd(const ir2rgb_conv_desc *d, const void *x, const void *wpacked, const float *bias,
                                 void *y, float *stats_partial, void *stream) {
    return ir2rgb_conv2d_fwd_ws(d, x, wpacked, bias, y, stats_partial, nullptr, 0, stream);
}

extern "C" int ir2rgb_conv2d_fwd_ws(const ir2rgb_conv_desc *d, const void *x, const void *wpacked, const float *bias,
                                    void *y, float *stats_partial, void *workspace, long workspace_bytes, void *stream) {
    if (!d) return IR2RGB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)wpacked | (uintptr_t)y) & 15) != 0) return IR2RGB_EALIGN;
    if (bias == nullptr && stats_partial == nullptr) {
        T7Geom g7;      // the 1x7 pass of the separable heads: row-segment staging, weights in registers (conv1x7_thin.hip)
        if (conv1x7_thin_plan(d, &g7)) return conv1x7_thin_launch(g7, d->dtype, d->Cin, x, wpacked, y, as_stream(stream));
    }
    {
        C7Geom gc;      // the 7x1 pass of the generators' first layers: column tiles staged once (conv7x1_col.hip)
        if (conv7x1_col_plan(d, &gc)) return conv7x1_col_launch(gc, d->dtype, x, wpacked, bias, y, stats_partial, as_stream(stream));
    }
    {
        P3Geom g3;
        int npt3 = 0;
        const int variant = conv3x3p_plan(d, &g3, &npt3, workspace != nullptr);
        if (variant)
            return conv3x3p_launch(variant, g3, d->dtype, x, wpacked, bias, y, stats_partial, as_stream(stream), workspace, workspace_bytes);
    }
    ClassPlan plans[4];
    int n = make_plan(d, plans);
    if (n < 0) return n;
    if (n == 1 && stats_partial == nullptr && conv_dot_ok(d)) {
        const ConvGeom &g = plans[0].geom;
        const long P = (long)g.N * g.Hout * g.Wout;
        const long waves = P < 8192 ? P : 8192;
        const unsigned grid = (unsigned)((waves + 3) / 4);
        if (d->dtype == IR2RGB_BF16)
            conv_dot_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t *)x, (const uint16_t *)wpacked, bias, (float *)y, g);
        else
            conv_dot_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t *)x, (const uint16_t *)wpacked, bias, (float *)y, g);
        return ir2rgb_launch_status();
    }
    if (n > 1 && merge_classes()) {
        // one launch for all classes: same pixel-tile size for all of them (plans were made with it)
        ConvClasses cs;
        int order[4] = {0, 1, 2, 3};
        for (int i = 0; i < n; ++i)           // longest K loop first
            for (int j = i + 1; j < n; ++j)
                if (plans[order[j]].geom.ntaps > plans[order[i]].geom.ntaps) { int t = order[i]; order[i] = order[j]; order[j] = t; }
        int first = 0;
        for (int i = 0; i < n; ++i) {
            const ClassPlan &c = plans[order[i]];
            cs.g[i] = c.geom;
            cs.w_off[i] = c.w_offset;
            cs.first[i] = first;
            const int nwg = c.npt * ((c.geom.Cout + 127) / 128);
            first += (nwg + 7) & ~7;
        }
        cs.first[n] = first;
        cs.n = n;
        const int tp = plans[0].tp;
        hipStream_t s = as_stream(stream);
        const uint16_t *X = (const uint16_t *)x, *W = (const uint16_t *)wpacked;
        uint16_t *Yp = (uint16_t *)y;
        if (d->dtype == IR2RGB_BF16) {
            if (tp == 256) conv_igemm_classes_kernel<IR2RGB_BF16, 256><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
            else if (tp == 128) conv_igemm_classes_kernel<IR2RGB_BF16, 128><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
            else conv_igemm_classes_kernel<IR2RGB_BF16, 64><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
        } else {
            if (tp == 256) conv_igemm_classes_kernel<IR2RGB_F16, 256><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
            else if (tp == 128) conv_igemm_classes_kernel<IR2RGB_F16, 128><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
            else conv_igemm_classes_kernel<IR2RGB_F16, 64><<<first, 512, 0, s>>>(X, W, bias, Yp, stats_partial, cs);
        }
        return ir2rgb_launch_status();
    }
    for (int i = 0; i < n; ++i) {
        const uint16_t *wp = reinterpret_cast<const uint16_t *>(wpacked) + plans[i].w_offset;
        if (d->dtype == IR2RGB_BF16)
            launch_conv<IR2RGB_BF16>(plans[i], (const uint16_t *)x, wp, bias, (uint16_t *)y, stats_partial, as_stream(stream));
        else
            launch_conv<IR2RGB_F16>(plans[i], (const uint16_t *)x, wp, bias, (uint16_t *)y, stats_partial, as_stream(stream));
    }
    return ir2rgb_launch_status();
}
