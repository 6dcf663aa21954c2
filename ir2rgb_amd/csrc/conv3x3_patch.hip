// conv3x3_patch.hip -- 3x3 / stride-1 convolutions with many input channels (the residual blocks:
// 84 % of the generator's forward flops) as an implicit GEMM whose activation operand is staged ONCE
// per channel slice instead of once per tap.
//
// conv_igemm_kernel (conv_mfma.hip) stages, for every (tap, 64-channel slice), a fresh im2col tile of the
// pixels next to the weight tile: 48 one-KB LDS-DMA instructions per 256 MFMAs at its largest tile, 24
// per 64 at its smallest -- the K loop of those layers is bound by DMA issue + barrier cadence, not by
// the matrix cores (profiles/r01_dominant_kernels_pmc_summary.txt).  The nine taps of a 3x3 kernel read
// the same pixels shifted by at most one row / column, so here a workgroup owns a 2 x TW pixel tile and
// keeps a haloed 4 x (TW+2) pixel patch of a 32-channel slice in LDS; the tap (ky,kx) operand of pixel
// (ty,tx) is simply patch entry (ty+ky, tx+kx).  Per 32-channel slice the patch is staged once (one
// row pair with the ky = 0 step, one further row with each of the ky = 1, 2 steps) and only the
// weights change per step: 42+33+33 DMA instructions per 288 MFMAs x 4 waves at the 2x128 px x 128
// cout tile (1 : 10.7 against 1 : 5.3).
//
// LDS rows are 64 B (32 channels); 16-byte position s of row R holds source chunk s ^ ((R>>1)&2), which makes
// the 16 lanes that a ds_read_b128 serves per LDS cycle hit 16 distinct bank windows for any row offset
// (without it: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles); every fragment address is a per-lane base + a
// compile-time immediate.  K-step = (32-channel slice, ky): three taps, MFMA K = 32.
// Waves: NCW multiply (MI = 4 x NI accumulator tiles each; two per SIMD at the large tile so that one
// wave's LDS latency and barrier wait hide under the other's MFMAs), NLW only stage (pure LDS-DMA issue: per-lane
// gather offsets are constants of the workgroup's tile, the slice offset rides in an SGPR); one
// s_barrier and one counted vmcnt per K-step; weights in an NSTW-stage ring, the patch double-buffered.
// Epilogue: bias, LeakyReLU, per-tile BatchNorm partial sums (same [tiles][2][Cout] contract as
// conv_igemm_kernel), half outputs staged through LDS and written as 16-byte stores.
#include <utility>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 p3_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 p3_f16x8;
typedef __attribute__((ext_vector_type(4))) float p3_f32x4;
typedef __attribute__((address_space(3))) void *p3_lptr_t;
typedef __amdgpu_buffer_rsrc_t p3_rsrc_t;
#define P3_OOB 0x80000000u

template <int DT> struct P3Half;
template <> struct P3Half<IR2RGB_BF16> {
    typedef p3_bf16x8 frag;
    static __device__ __forceinline__ p3_f32x4 mfma(frag a, frag b, p3_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t cvt(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
};
template <> struct P3Half<IR2RGB_F16> {
    typedef p3_f16x8 frag;
    static __device__ __forceinline__ p3_f32x4 mfma(frag a, frag b, p3_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t cvt(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
};

struct P3Geom {
    int N, H, W, Ho, Wo, Cin, Cout;   // input / output extents
    int pad, pad_mode, act;
    int ldx, ci_off, ldy, co_off;
    int stats_row0, nty, ntx;         // pixel tiles per image: nty x ntx
    int cout_major, kchunks;
    unsigned x_bytes, w_bytes;
    int dbg;   // ablation switches for timing experiments, honoured only by -DIR2RGB_ABLATION builds (results are garbage): 1 = no staging after the prologue, 2 = no fragment reads
};

__device__ __forceinline__ int p3_reflect(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * n - 2 - v : v;
}

// SPS = 32-channel slices per K-step (1 or 2): with 2, a step covers a whole 64-channel chunk at one ky --
// twice the MFMA work between barriers for the small tile, whose steps are otherwise only 24 MFMAs long.
template <int TW, int TCO, int NCW, int NLW, int ADJ, int SPS, int TR_ = 2> struct P3Cfg {
    static constexpr int TR = TR_, NPX = TR * TW;            // pixel tile: TR rows x TW columns
    static constexpr int PWP = ((TW + 2 + 15) / 16) * 16;   // patch row pitch in entries (80 | 144)
    static constexpr int PROWI = PWP / 16;                   // DMA instructions per patch row
    static constexpr int NW1 = 3 * TCO / 16, NW = SPS * NW1; // DMA instructions for the weights of one K-step
    static constexpr int WST1 = 3 * TCO * 64, WST = SPS * WST1;   // bytes per weight stage (per slice, per step)
    static constexpr int PBUF = (TR + 2) * PWP * 64;         // bytes per patch buffer (one slice): TR + 2 haloed rows
    static constexpr int NSTW = (SPS == 2 || TCO == 128 && (TW == 128 || TR > 2)) ? 3 : 4;
    static constexpr int AHEAD = NSTW - 1;
    static constexpr int WM = TCO / 64, WN = NCW / WM, PXW = NPX / WN, NI = PXW / 16;
    static constexpr int LDS = NSTW * WST + 2 * SPS * PBUF + 1024;
    // patch rows staged with the step of phase ky: rows {0..TR-1} | {TR} | {TR+1}; reflect-adjoint mode needs every row
    // from the first step on (its border terms read row 2 at ky = 0), so it stages all of them with ky = 0
    static constexpr int np1(int ky) { return ADJ ? (ky == 0 ? (TR + 2) * PROWI : 0) : (ky == 0 ? TR * PROWI : PROWI); }
    static constexpr int np(int ky) { return SPS * np1(ky); }
    static constexpr int nl(int ky) { return (NW + np(ky) + NLW - 1) / NLW; }     // DMA instructions per loader wave
    static constexpr int NLMAX = nl(0);
    // DMA instructions a loader wave may leave in flight while step (phase ky) is consumed: the steps staged after it
    static constexpr int out(int ky) {
        int s = 0;
        for (int d = 1; d < AHEAD; ++d) s += nl((ky + d) % 3);
        return s;
    }
};

template <int DT, int TW, int TCO, int NCW, int NLW, int PIPE, int ADJ, int SPS, int SPLIT = 1, int TR = 2>
__global__ void __launch_bounds__((NCW + NLW) * 64, 1)
conv3x3_patch_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias,
                     uint16_t *__restrict__ Y, float *__restrict__ stats_partial, const P3Geom g,
                     unsigned *tickets = nullptr, float *partials = nullptr) {
    static_assert(SPLIT == 1 || SPLIT == 2, "the hand-over adds two partial tiles (order-independent)");
    typedef P3Cfg<TW, TCO, NCW, NLW, ADJ, SPS, TR> C;
    static_assert(SPS == 1 || (PIPE == 1 && !ADJ), "two slices per step: pipelined forward form only");
    static_assert(TR == 2 || C::PXW == TW, "taller tiles: one pixel row per multiplying wave");
    static_assert(!(ADJ && PIPE) || C::NI <= 2 || C::PXW % TW == 0,
                  "pipelined reflect-adjoint form: border operands must be prefetched (registers: small tile, or whole pixel rows per wave)");
    typedef P3Half<DT> Hf;
    typedef typename Hf::frag frag;
    constexpr int NI = C::NI, MI = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C::LDS];
    unsigned char *const wring = smem, *const pbufs = smem + C::NSTW * C::WST, *const dummy = pbufs + 2 * SPS * C::PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= NCW;

    // ---- tile of this workgroup (XCD-aware ids as in conv_igemm_kernel) ----
    const int npt = g.N * g.nty * g.ntx, nct = g.Cout / TCO;
    int tile;
    {
        const int nwg = npt * nct * SPLIT, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    // the SPLIT workgroups of a tile have consecutive ids: same XCD, their partial sums meet in its L2
    const int ksplit = SPLIT > 1 ? tile % SPLIT : 0;
    if (SPLIT > 1) tile /= SPLIT;
    const int kcl = g.kchunks / SPLIT, kc0 = ksplit * kcl;   // this workgroup's 64-channel chunks: [kc0, kc0 + kcl)
    int pt, ct;
    if (g.cout_major) { ct = tile / npt; pt = tile - ct * npt; }
    else              { pt = tile / nct; ct = tile - pt * nct; }
    const int txi = pt % g.ntx, tyi = (pt / g.ntx) % g.nty, n = pt / (g.ntx * g.nty);
    const int y0 = tyi * C::TR, x0 = txi * TW;
    const int NK = kcl * (2 / SPS) * 3;         // K-steps: (64-channel chunk, [half,] ky)

    p3_f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (p3_f32x4){0.f, 0.f, 0.f, 0.f};

    if (loader) {
        // =============================== staging waves ===============================
        const int lw = wave - NCW;
        const p3_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(Wp), 0, (int)g.w_bytes, 0x00020000);
        const p3_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(X), 0, (int)g.x_bytes, 0x00020000);
        const int row16 = lane >> 2;
        // per (phase ky, slot j): instruction id = lw + NLW*j -> weights (id < NW), patch, or padding
        unsigned voff[3][C::NLMAX];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < C::NLMAX; ++j) {
                const int id = lw + NLW * j;
                unsigned v = P3_OOB;
                if (j < C::nl(ky)) {
                    // LDS slot (row R, 16-byte position s) holds source chunk s ^ ((R >> 1) & 2).  ds_read_b128 is
                    // served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS):
                    // with this XOR the 16 lanes of every group hit 16 distinct 16-byte bank windows for ANY row
                    // offset (exhaustive check over offsets 0..15), which the tap shifts kx = 0..2 need.
                    // R & 15 == row16 in every region.
                    const int chunk = (lane & 3) ^ ((row16 >> 1) & 2);
                    if (id < C::NW) {
                        const int row = (id % C::NW1) * 16 + row16, kx = row / TCO, col = row - kx * TCO;
                        const int co = ct * TCO + col;
                        v = (unsigned)((((long)co * g.kchunks) * 9 + kx) * 128 + chunk * 16);
                    } else if (id < C::NW + C::np(ky)) {
                        const int q = (id - C::NW) % (C::np1(ky) > 0 ? C::np1(ky) : 1);
                        const int pr = ky == 0 ? q / C::PROWI : TR + ky - 1;   // (ADJ: ky == 0 covers every row)
                        const int pc = (q % C::PROWI) * 16 + row16;
                        int iy = y0 - g.pad + pr, ix = x0 - g.pad + pc;
                        bool ok = pc < TW + 2;
                        if (g.pad_mode == 1) { iy = p3_reflect(iy, g.H); ix = p3_reflect(ix, g.W); ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W; }
                        else ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                        if (ok) v = (unsigned)((((long)n * g.H + iy) * g.W + ix) * g.ldx * 2 + g.ci_off * 2 + chunk * 16);
                    }
                }
                voff[ky][j] = v;
            }
        int is = 0;   // next K-step to stage
        auto issue = [&]<int KY>(std::integral_constant<int, KY>) {
            const bool live = is < NK;
            const int hs = is / 3;                       // slice (SPS == 1) or 64-channel chunk (SPS == 2) of this step
            const int cc = kc0 + (SPS == 2 ? hs : hs >> 1), half0 = SPS == 2 ? 0 : hs & 1;
            const unsigned w_soff = (unsigned)((cc * 9 + KY * 3) * 128 + half0 * 64);
            const unsigned x_soff = (unsigned)((cc * 64 + half0 * 32) * 2);
            unsigned char *wdst = wring + (is % C::NSTW) * C::WST;
            unsigned char *pdst = pbufs + (hs & 1) * SPS * C::PBUF;
            if ((g.dbg & 1) && is >= C::AHEAD) { ++is; return; }
#pragma unroll
            for (int j = 0; j < C::nl(KY); ++j) {
                const int id = lw + NLW * j;             // wave-uniform
                const unsigned v = live ? voff[KY][j] : P3_OOB;
                if (id < C::NW) {
                    const unsigned hoff = (unsigned)(id / C::NW1) * 64u;       // second slice of the step: +32 channels
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (p3_lptr_t)(wdst + id * 1024), 16, v, w_soff + hoff, 0, 0);
                } else if (id < C::NW + C::np(KY)) {
                    constexpr int NP1 = C::np1(KY) > 0 ? C::np1(KY) : 1;
                    const int qq = id - C::NW, h = qq / NP1, q = qq - h * NP1;
                    const int slot = KY == 0 ? q : (TR + KY - 1) * C::PROWI + q;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (p3_lptr_t)(pdst + h * C::PBUF + slot * 1024), 16, v,
                                                             x_soff + (unsigned)h * 64u, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (p3_lptr_t)dummy, 16, P3_OOB, 0, 0, 0);
                }
            }
            ++is;
        };
        // prologue: AHEAD steps (phases 0 .. AHEAD-1)
        issue(std::integral_constant<int, 0>{});
        if constexpr (C::AHEAD >= 2) issue(std::integral_constant<int, 1>{});
        if constexpr (C::AHEAD >= 3) issue(std::integral_constant<int, 2>{});
        for (int hs = 0; hs < NK / 3; ++hs) {
            auto step = [&]<int KY>(std::integral_constant<int, KY>) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::out(KY)) : "memory");   // this step's operands have landed
                __builtin_amdgcn_s_barrier();                                         // ... and the previous step is consumed
                issue(std::integral_constant<int, (KY + C::AHEAD) % 3>{});
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PIPE == 2) __builtin_amdgcn_s_barrier();   // (the multiplying waves' unconditional barrier of the last step)
    } else {
        // =============================== multiplying waves ===============================
        const int wm = wave / C::WN, wn = wave - wm * C::WN;
        const int l15 = lane & 15, grp = lane >> 4;
        unsigned aofs[MI], bofs[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aofs[mi] = (unsigned)((wm * 64 + mi * 16 + l15) * 64 + ((grp ^ ((l15 >> 1) & 2)) * 16));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int p = wn * C::PXW + ni * 16, ty = p / TW, tx = p - ty * TW;
            // entry (ty+ky)*PWP + tx + l15 + kx; PWP and tx are multiples of 16, so the swizzle bit of tap kx is that
            // of l15 + kx: the kx = 0 address plus kx*64, with address bit 5 flipped when l15 + kx carries into bit 2
            bofs[ni] = (unsigned)((ty * C::PWP + tx + l15) * 64 + ((grp ^ ((l15 >> 1) & 2)) * 16));
        }
        const unsigned flip1 = ((l15 & 3) + 1 >= 4) ? 32u : 0u, flip2 = ((l15 & 3) + 2 >= 4) ? 32u : 0u;
        // Software pipeline across the K-step barrier: the fragments of tap kx+1 are fetched while tap kx is
        // multiplied, and the barrier that opens step ks+1 sits between the LAST fragment fetch of step ks and
        // its MFMAs (the fetch is complete: lgkmcnt(0)), so those MFMAs cover the barrier wait and the first
        // fetch of the next step.  (With the barrier at the top of a step every wave of the CU idled through
        // its first LDS round trip: MFMA-busy 58 %.)
        frag fa[PIPE ? 2 : 1][MI], fb[PIPE == 1 ? 2 : 1][NI];
        auto fetch = [&](int buf, const unsigned char *wst, const unsigned char *patch, int kyoff, int kx) {
            if (g.dbg & 2) return;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[buf][mi] = *reinterpret_cast<const frag *>(wst + aofs[mi] + kx * TCO * 64);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                fb[buf][ni] = *reinterpret_cast<const frag *>(patch + (bofs[ni] ^ (kx == 0 ? 0u : (kx == 1 ? flip1 : flip2))) + kx * 64 + kyoff);
        };
        auto mma = [&](int buf) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Hf::mfma(fa[buf][mi], fb[buf][ni], acc[mi][ni]);
        };
        // Reflect-adjoint mode (pad_mode 2): this launch is the data gradient of a reflection-padded 3x3 convolution,
        //   dx(q) = sum over padded positions u that reflect onto q of F(u),  F = full correlation of gy with the
        //   flipped kernel (zero outside).  Besides u = q, row 1 also receives F(-1, .), row H-2 F(H, .), column 1
        //   F(., -1), column W-2 F(., W), and the four pixels next to the corners the corner values.  Each of these is
        //   a tap of the same weights applied to a patch entry of this tile at another offset:
        //     F(-1, x)  = sum_sx Wf(+1, sx) gy(0, x+sx)    -> pixel row ty = 1 of a top tile, weights ky = 2, entries of ky = 0
        //     F(H, x)   = sum_sx Wf(-1, sx) gy(H-1, x+sx)  -> pixel row ty = TR-2 of a bottom tile, weights ky = 0, entries of ky = 2
        //     F(y, -1)  = sum_sy Wf(sy, +1) gy(y+sy, 0)    -> pixel tx = 1, weights kx = 2, the entry of kx = 0
        //     F(y, W)   = sum_sy Wf(sy, -1) gy(y+sy, W-1)  -> pixel tx = TW-2, weights kx = 0, the entry of kx = 2
        //   so the padded 34 x 66 output grid and the fold pass disappear.  (H even, W % TW == 0: whole tiles.)
        const bool top = y0 == 0, bot = y0 == g.H - TR, left = x0 == 0, right = x0 + TW == g.W;
        // The border operands of a step are fetched right after its barrier (their LDS latency then hides under the
        // regular taps; fetched at the point of use they cost a full LDS round trip per term on a wave that has its
        // SIMD to itself: 67.6 -> see profiles) and multiplied after the tap whose weights they use.
        // (the large tile has two waves per SIMD to hide the latency and no registers to spare)
        constexpr bool PRE = NI <= 2 || (PIPE && C::PXW % TW == 0);
        frag eb_row[3][PRE ? NI : 1], eb_l[PRE ? NI : 1], eb_r[PRE ? NI : 1];
        auto entry = [&](const unsigned char *patch, int ni, int kyoff, int kxb) -> frag {
            return *reinterpret_cast<const frag *>(patch + (bofs[ni] ^ (kxb == 0 ? 0u : (kxb == 1 ? flip1 : flip2))) + kxb * 64 + kyoff);
        };
        auto prefetch_terms = [&]<int KY>(const unsigned char *patch, std::integral_constant<int, KY>) {
            if constexpr (PRE) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    // wave-uniform; tx is a compile-time constant where a wave owns whole pixel rows (unused operands then vanish)
                    const int ty = (wn * C::PXW + ni * 16) / TW;
                    const int tx = (C::PXW % TW == 0) ? (ni * 16) % TW : (wn * C::PXW + ni * 16) % TW;
                    const bool rowterm = (KY == 2 && top && ty == 1) || (KY == 0 && bot && ty == TR - 2);
                    if (rowterm) {
                        const int kyoff = KY == 2 ? 0 : 2 * C::PWP * 64;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) eb_row[kx][ni] = entry(patch, ni, kyoff, kx);
                    }
                    if (left && tx == 0) eb_l[ni] = entry(patch, ni, KY * C::PWP * 64, 0);
                    if (right && tx == TW - 16) eb_r[ni] = entry(patch, ni, KY * C::PWP * 64, 2);
                }
            }
        };
        auto border_terms = [&]<int KY>(const unsigned char *patch, std::integral_constant<int, KY>, int kx, int abuf) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                // wave-uniform; tx is a compile-time constant where a wave owns whole pixel rows (unused operands then vanish)
                    const int ty = (wn * C::PXW + ni * 16) / TW;
                    const int tx = (C::PXW % TW == 0) ? (ni * 16) % TW : (wn * C::PXW + ni * 16) % TW;
                auto term = [&](frag b, int only_lane) {
                    if (only_lane >= 0 && l15 != only_lane) b = frag{};
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = Hf::mfma(fa[abuf][mi], b, acc[mi][ni]);
                };
                const bool rowterm = (KY == 2 && top && ty == 1) || (KY == 0 && bot && ty == TR - 2);
                const int rowoff = KY == 2 ? 0 : 2 * C::PWP * 64;
                constexpr int pi = PRE ? 1 : 0;   // index helper: prefetched arrays are per ni only when PRE
                if (rowterm) term(PRE ? eb_row[kx][ni * pi] : entry(patch, ni, rowoff, kx), -1);
                if (kx == 2 && left && tx == 0) term(PRE ? eb_l[ni * pi] : entry(patch, ni, KY * C::PWP * 64, 0), 1);
                if (kx == 0 && right && tx == TW - 16) term(PRE ? eb_r[ni * pi] : entry(patch, ni, KY * C::PWP * 64, 2), 14);
                // corners: the row-term entry of the other tap, one lane only
                if (rowterm && kx == 2 && left && tx == 0) term(PRE ? eb_row[0][ni * pi] : entry(patch, ni, rowoff, 0), 1);
                if (rowterm && kx == 0 && right && tx == TW - 16) term(PRE ? eb_row[2][ni * pi] : entry(patch, ni, rowoff, 2), 14);
            }
        };
        if constexpr (PIPE == 1 && SPS == 2) {
            // a step = one ky of a whole 64-channel chunk: six taps (slice, kx); tap t lives in fragment buffer t & 1,
            // the barrier that opens the next step sits before the MFMAs of the last tap (see below)
            int ks = 0;
            __builtin_amdgcn_s_barrier();                       // step 0 has landed
            fetch(0, wring, pbufs, 0, 0);
            for (int cc = 0; cc < kcl; ++cc) {
                const unsigned char *patch = pbufs + (cc & 1) * 2 * C::PBUF;
                const unsigned char *patch_next = pbufs + ((cc + 1) & 1) * 2 * C::PBUF;
                auto step = [&]<int KY>(std::integral_constant<int, KY>) {
                    const unsigned char *wst = wring + (ks % C::NSTW) * C::WST;
                    const unsigned char *wst_next = wring + ((ks + 1) % C::NSTW) * C::WST;
                    constexpr int KO = KY * C::PWP * 64;
                    fetch(1, wst, patch, KO, 1);
                    mma(0);
                    fetch(0, wst, patch, KO, 2);
                    mma(1);
                    fetch(1, wst + C::WST1, patch + C::PBUF, KO, 0);
                    mma(0);
                    fetch(0, wst + C::WST1, patch + C::PBUF, KO, 1);
                    mma(1);
                    fetch(1, wst + C::WST1, patch + C::PBUF, KO, 2);
                    mma(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of step ks is complete
                    if (ks + 1 < NK) {
                        __builtin_amdgcn_s_barrier();            // step ks+1 has landed; stage ks may be overwritten
                        if constexpr (KY < 2) fetch(0, wst_next, patch, (KY + 1) * C::PWP * 64, 0);
                        else fetch(0, wst_next, patch_next, 0, 0);
                    }
                    mma(1);
                    ++ks;
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
            }
        } else if constexpr (PIPE == 2) {
            // Large tile (NI = 4, two multiplying waves per SIMD, 168 registers): the full pipelined form below needs two
            // complete fragment sets (+32 registers: spills).  Here only the weight fragments are double-buffered (+16);
            // the pixel fragments are refilled column by column -- fb[ni] is reloaded for the NEXT tap right after the four
            // MFMAs that read it -- so the next tap's operands arrive while the current tap multiplies, within one set.
            // The barrier that opens step ks+1 sits in front of the last tap's MFMAs (its fragments are complete:
            // lgkmcnt(0), which is also what makes handing the stage back safe), and that tap refills from step ks+1.
            static_assert(!ADJ && SPS == 1, "column-refill form: forward, one slice per step");
            // The reads are inline asm and the waits are counted by hand: left to hipcc, the reads sink down to their uses
            // (it schedules for register pressure at this occupancy) and every MFMA group ends up behind a full
            // s_waitcnt of the read issued just before it.  A wait names the fragments it makes valid ("+v"), so the MFMAs
            // that read them cannot be scheduled above it; __builtin_amdgcn_sched_barrier(0) keeps the groups in order.
            // Addresses are a per-lane base register + a compile-time immediate (the weight stage of a step is its ky:
            // six steps per 64-channel chunk, three stages); the third stage lies beyond the 16-bit immediate's reach and
            // has base registers of its own.
            static_assert(C::NSTW == 3 && C::PBUF + 2 * 64 + 2 * C::PWP * 64 < 65536 && C::WST + 2 * TCO * 64 < 65536, "immediates");
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem;
            unsigned abase[2][MI], bbase[3][NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { abase[0][mi] = lds0 + aofs[mi]; abase[1][mi] = lds0 + aofs[mi] + 2 * C::WST; }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bbase[0][ni] = lds0 + C::NSTW * C::WST + bofs[ni];
                bbase[1][ni] = lds0 + C::NSTW * C::WST + (bofs[ni] ^ flip1);
                bbase[2][ni] = lds0 + C::NSTW * C::WST + (bofs[ni] ^ flip2);
            }
            auto rd = [&]<int IMM>(frag &dst, unsigned addr, std::integral_constant<int, IMM>) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
            };
            // operands of tap (stage ST, column kx; patch buffer HB, row offset ky): weights -> fa[buf], one pixel column -> fb[0][ni]
            auto fetchA = [&]<int ST, int KX>(int buf, std::integral_constant<int, ST>, std::integral_constant<int, KX>) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    rd(fa[buf][mi], abase[ST == 2][mi], std::integral_constant<int, (ST == 2 ? 0 : ST * C::WST) + KX * TCO * 64>{});
            };
            auto fetchB = [&]<int HB, int KYR, int KX>(int ni, std::integral_constant<int, HB>, std::integral_constant<int, KYR>,
                                                       std::integral_constant<int, KX>) {
                rd(fb[0][ni], bbase[KX][ni], std::integral_constant<int, HB * C::PBUF + KX * 64 + KYR * C::PWP * 64>{});
            };
            // one tap: weights in fa[cur], pixels in fb[0]; meanwhile the next tap's weights go to fa[cur ^ 1] and its pixel
            // fragments replace fb[0] column by column.  LDS operations in flight, oldest first, when column ni is due:
            // [this tap's operands up to fb[ni]] fb[ni+1..3] | the four fa[cur ^ 1] reads | the refills fb[0..ni-1]: seven
            // younger ones in every column, so the wait is lgkmcnt(7) throughout (after the step's lgkmcnt(0): a no-op).
            // (no condition around the fetches or the barrier: with a branch in the loop hipcc's s_waitcnt insertion merges
            // the two paths' LDS scoreboards; the last step therefore fetches a tap nobody multiplies, from stages that
            // exist, and the staging waves run one barrier more to match.)
            auto tap = [&]<int ST, int HB, int KYR, int KX>(int cur, std::integral_constant<int, ST> st, std::integral_constant<int, HB> hb,
                                                             std::integral_constant<int, KYR> kyr, std::integral_constant<int, KX> kx) {
                fetchA(cur ^ 1, st, kx);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (ni == 0)
                        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fa[cur][0]), "+v"(fa[cur][1]), "+v"(fa[cur][2]), "+v"(fa[cur][3]), "+v"(fb[0][0]) :: "memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fb[0][ni]) :: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = Hf::mfma(fa[cur][mi], fb[0][ni], acc[mi][ni]);
                    __builtin_amdgcn_sched_barrier(0);
                    fetchB(ni, hb, kyr, kx);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            __builtin_amdgcn_s_barrier();                       // step 0 has landed
            fetchA(0, I0{}, I0{});
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fetchB(ni, I0{}, I0{}, I0{});
            for (int cc = 0; cc < kcl; ++cc) {
                // step (ky = KY of slice HALF): weight stage KY, patch buffer HALF; P = fragment set of its first tap
                auto step = [&]<int KY, int P, int HALF>(std::integral_constant<int, KY> ky, std::integral_constant<int, P>,
                                                         std::integral_constant<int, HALF> half) {
                    tap(P, ky, half, ky, I1{});                  // tap kx = 0; fetches kx = 1
                    tap(P ^ 1, ky, half, ky, I2{});              // tap kx = 1; fetches kx = 2
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this step is complete
                    __builtin_amdgcn_s_barrier();                // the next step has landed; this step's stage may be overwritten
                    __builtin_amdgcn_sched_barrier(0);
                    // tap kx = 2; fetches tap 0 of the next step: next ky of this slice, or ky = 0 of the next slice
                    if constexpr (KY < 2) tap(P, std::integral_constant<int, KY + 1>{}, half, std::integral_constant<int, KY + 1>{}, I0{});
                    else tap(P, I0{}, std::integral_constant<int, HALF ^ 1>{}, I0{}, I0{});
                };
                step(I0{}, I0{}, I0{});
                step(I1{}, I1{}, I0{});
                step(I2{}, I0{}, I0{});
                step(I0{}, I1{}, I1{});
                step(I1{}, I0{}, I1{});
                step(I2{}, I1{}, I1{});
            }
        } else if constexpr (PIPE) {
            int ks = 0;
            __builtin_amdgcn_s_barrier();                       // step 0 has landed
            fetch(0, wring, pbufs, 0, 0);
            // reflect-adjoint mode: the border operands of a step are prefetched into registers as soon as the barrier
            // that opens the step has passed (for step 0: here), and consumed right after the tap whose weights they
            // use -- no LDS read of a step's data is left after the barrier that hands its weight stage back to the loaders
            if constexpr (ADJ) prefetch_terms(pbufs, std::integral_constant<int, 0>{});
            // six steps per 64-channel chunk (two 32-channel slices x ky); the fragment buffers alternate with a
            // compile-time parity P (three taps per step flip it once per step)
            for (int cc = 0; cc < kcl; ++cc) {
                auto step = [&]<int KY, int P, int HALF>(std::integral_constant<int, KY>, std::integral_constant<int, P>,
                                                         std::integral_constant<int, HALF>) {
                    const unsigned char *patch = pbufs + HALF * C::PBUF;          // slice 2*cc + HALF
                    const unsigned char *patch_next = pbufs + (HALF ^ 1) * C::PBUF;
                    const unsigned char *wst = wring + (ks % C::NSTW) * C::WST;
                    const unsigned char *wst_next = wring + ((ks + 1) % C::NSTW) * C::WST;
                    using IKY = std::integral_constant<int, KY>;
                    fetch(P ^ 1, wst, patch, KY * C::PWP * 64, 1);
                    mma(P);                                      // tap kx = 0 (fetched during the previous step)
                    if constexpr (ADJ) border_terms(patch, IKY{}, 0, P);
                    fetch(P, wst, patch, KY * C::PWP * 64, 2);
                    mma(P ^ 1);                                  // tap kx = 1
                    if constexpr (ADJ) border_terms(patch, IKY{}, 1, P ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of step ks is complete
                    if (ks + 1 < NK) {
                        __builtin_amdgcn_s_barrier();            // step ks+1 has landed; stage ks may be overwritten
                        // tap 0 of the next step: next ky of this slice, or ky = 0 of the next slice (other patch buffer)
                        if constexpr (KY < 2) fetch(P ^ 1, wst_next, patch, (KY + 1) * C::PWP * 64, 0);
                        else fetch(P ^ 1, wst_next, patch_next, 0, 0);
                    }
                    mma(P);                                      // tap kx = 2
                    if constexpr (ADJ) {
                        border_terms(patch, IKY{}, 2, P);
                        if (ks + 1 < NK) {
                            if constexpr (KY < 2) prefetch_terms(patch, std::integral_constant<int, KY + 1>{});
                            else prefetch_terms(patch_next, std::integral_constant<int, 0>{});
                        }
                    }
                    ++ks;
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                step(I0{}, I0{}, I0{});
                step(I1{}, I1{}, I0{});
                step(I2{}, I0{}, I0{});
                step(I0{}, I1{}, I1{});
                step(I1{}, I0{}, I1{});
                step(I2{}, I1{}, I1{});
            }
        } else {
            // plain form: barrier at the top of every step (two multiplying waves per SIMD hide each other's LDS
            // latency; the pipelined form above needs 2 x (MI + NI) more fragment registers than that occupancy allows)
            int ks = 0;
            for (int hs = 0; hs < kcl * 2; ++hs) {
                const unsigned char *patch = pbufs + (hs & 1) * C::PBUF;
                auto step = [&]<int KY>(std::integral_constant<int, KY>) {
                    // the previous step's fragment reads have returned before the barrier that hands its weight stage
                    // (and, every third step, its patch buffer) back to the loaders: a raw s_barrier does not keep the
                    // compiler from sinking their wait below it (conv_mfma.hip wait_stage; tools/check_lds_war.py)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    const unsigned char *wst = wring + (ks % C::NSTW) * C::WST;
                    if constexpr (ADJ) prefetch_terms(patch, std::integral_constant<int, KY>{});
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        fetch(0, wst, patch, KY * C::PWP * 64, kx);
                        mma(0);
                        if constexpr (ADJ) border_terms(patch, std::integral_constant<int, KY>{}, kx, 0);
                    }
                    ++ks;
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
            }
        }
    }

    // ---------------- epilogue ----------------
    // bias + activation + BatchNorm partial sums from the accumulators; the half outputs go through LDS
    // ([pixel][TCO couts], 16-byte chunk c of pixel row p at position c ^ (p & (TCO/8 - 1)): conflict-free both ways)
    // so that global memory sees 16-byte stores, TCO/8 lanes per contiguous TCO*2-byte pixel row, instead of 8-byte
    // pieces 2 KB apart.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is past its last LDS read / its last DMA has landed: smem is reusable
    if constexpr (SPLIT > 1) {
        // Hand-over between the two workgroups of this tile: [tile][split][wave][mi][ni][lane] float4, 1 KB per store.
        // The partial sums travel as 16-byte sc1 (write-through) stores and sc1 loads around an agent-scope ticket, with every
        // storing wave's s_waitcnt vmcnt(0) and the workgroup barrier between the stores and the ticket
        // (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms: one lane adds to one counter, the workgroup whose
        // add came last consumes; hipMalloc memory, one workgroup per CU).  A release / acquire fence pair (__threadfence)
        // would be the textbook form, but on this chip it is buffer_wbl2 + buffer_inv: a walk over the XCD's whole L2 per
        // workgroup -- measured 104 us instead of 49 for the launch.
        typedef __attribute__((ext_vector_type(4))) unsigned p3_u32x4;
        constexpr int PER = NCW * MI * NI * 64 * 16;       // bytes per partial tile
        // exact extent of the partial tiles ([tiles][SPLIT] x PER bytes; the host keeps it under 1 GB): an offset past it
        // is dropped / reads zeros instead of touching a neighbour allocation
        const p3_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(partials, 0, npt * nct * SPLIT * PER, 0x00020000);
        const unsigned mine = (unsigned)((tile * SPLIT + ksplit) * PER), other = (unsigned)((tile * SPLIT + (ksplit ^ 1)) * PER);
        if (!loader) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, acc[mi][ni]), rp,
                                                           mine + (unsigned)((((wave * MI + mi) * NI + ni) * 64 + lane) * 16), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial sums have reached the coherence point ...
        __builtin_amdgcn_s_barrier();                      // ... every wave's have, before the ticket is drawn
        unsigned *flag = reinterpret_cast<unsigned *>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_barrier();
        const unsigned ticket = *flag;
        if (ticket % SPLIT != SPLIT - 1) return;           // not the last arrival: done (uniform over the workgroup)
        __builtin_amdgcn_s_barrier();                      // (the flag word is reused by the output staging below)
        if (!loader) {
            // all 16 loads are issued before the first is consumed (they are served by the memory side, ~2 us: issued in
            // dependent groups they took longer than the K loop); own + other: commutative, either arrival order gives the
            // same bits
            p3_u32x4 tmp[MI][NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    tmp[mi][ni] = __builtin_amdgcn_raw_buffer_load_b128(rp, other + (unsigned)((((wave * MI + mi) * NI + ni) * 64 + lane) * 16), 0, 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += __builtin_bit_cast(p3_f32x4, tmp[mi][ni]);
        }
    }
    constexpr int NCH = TCO / 8;                                    // 16-byte chunks per staged pixel row
    unsigned char *otile = smem;                                    // NPX * TCO * 2 bytes
    float *red = reinterpret_cast<float *>(smem + C::NPX * TCO * 2);   // [WN][TCO][2]
    static_assert(C::NPX * TCO * 2 + C::WN * TCO * 2 * 4 <= C::LDS, "epilogue staging exceeds the ring");
    if (!loader) {
        const int wm = wave / C::WN, wn = wave - wm * C::WN;
        const int l15 = lane & 15, grp = lane >> 4;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int cl = wm * 64 + mi * 16 + grp * 4;      // cout within the tile (multiple of 4)
            const int co = ct * TCO + cl;
            float bv[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias ? bias[co + r] : 0.f;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int p = wn * C::PXW + ni * 16 + l15, ty = p / TW, tx = p - ty * TW;
                const int oy = y0 + ty, ox = x0 + tx;
                const bool valid = oy < g.Ho && ox < g.Wo;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mi][ni][r] + bv[r];
                    if (g.act) v[r] = v[r] > 0.f ? v[r] : (g.act == 1 ? 0.2f : (g.act == 2 ? 0.1f : 0.f)) * v[r];
                    if (valid) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                }
                uint2 pk;
                pk.x = (uint32_t)Hf::cvt(v[0]) | ((uint32_t)Hf::cvt(v[1]) << 16);
                pk.y = (uint32_t)Hf::cvt(v[2]) | ((uint32_t)Hf::cvt(v[3]) << 16);
                *reinterpret_cast<uint2 *>(otile + p * (TCO * 2) + (((cl >> 3) ^ (p & (NCH - 1))) << 4) + (cl & 4) * 2) = pk;
            }
            if (stats_partial != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
                if (l15 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        red[(wn * TCO + cl + r) * 2 + 0] = s1[r];
                        red[(wn * TCO + cl + r) * 2 + 1] = s2[r];
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    {
        // every thread of the workgroup (loader waves included) copies 16-byte chunks: NCH lanes per pixel row
        constexpr int NT = (NCW + NLW) * 64;
        const int c16 = tid % NCH;
        for (int p = tid / NCH; p < C::NPX; p += NT / NCH) {
            const int ty = p / TW, tx = p - ty * TW;
            const int oy = y0 + ty, ox = x0 + tx;
            if (oy < g.Ho && ox < g.Wo) {
                const uint4 v = *reinterpret_cast<const uint4 *>(otile + p * (TCO * 2) + ((c16 ^ (p & (NCH - 1))) << 4));
                *reinterpret_cast<uint4 *>(Y + (((long)n * g.Ho + oy) * g.Wo + ox) * g.ldy + g.co_off + ct * TCO + c16 * 8) = v;
            }
        }
        if (stats_partial != nullptr && tid < TCO * 2) {
            const int cl = tid >> 1, which = tid & 1;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < C::WN; ++w) t += red[(w * TCO + cl) * 2 + which];
            if constexpr (TR == 2) {
                stats_partial[((long)(g.stats_row0 + pt) * 2 + which) * g.Cout + ct * TCO + cl] = t;
            } else {
                // the statistics buffer keeps the rows of the 2-row tiling (ir2rgb_conv2d_stats_rows does not depend on the
                // variant): this tile's sums go to the first of the TR/2 rows it covers, zeros to the others
                const int nty2 = (g.Ho + 1) / 2;
#pragma unroll
                for (int j = 0; j < TR / 2; ++j) {
                    const int ty2 = tyi * (TR / 2) + j;
                    if (ty2 < nty2)
                        stats_partial[((long)(g.stats_row0 + (n * nty2 + ty2) * g.ntx + txi) * 2 + which) * g.Cout + ct * TCO + cl] = j == 0 ? t : 0.f;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side (called from conv_mfma.hip)
// ----------------------------------------------------------------------------------------
static bool p3_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV3X3P"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// Small tile with one 64-channel chunk per K-step instead of one 32-channel slice (half the barriers).  Measured
// neutral (51.7 / 54.0 us against 53.7 / 53.9 us): the 2x64 px x 64 cout tile is bound by LDS bandwidth, not by the
// barrier cadence -- four waves each read all 64 couts' weights for only 32 pixels: 0.75 KB of ds_read_b128 per MFMA
// = 192 B/clk of the 256 B/clk LDS, plus the DMA writes.  Kept selectable (IR2RGB_CONV3X3P_SPS2=1), off by default.
static bool p3_sps2() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV3X3P_SPS2"); v = e ? atoi(e) : 0; }
    return v != 0;
}

static long p3_min_tiles() {   // fewer tiles than this leave too much of the chip idle: the general kernel runs instead
    static long v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV3X3P_MIN_TILES"); v = e ? atol(e) : 200; }
    return v;
}

static int p3_min_cin() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV3X3P_MIN_CIN"); v = e ? atoi(e) : 256; }
    return v;
}

// Split-K forms (variants 3, 4; only where the caller supplies a workspace).  IR2RGB_CONV3X3P_SPLIT=0 switches them off,
// =3 / =4 forces one for A/B runs; default: 4 where it applies, else 3.  Measured on 1024 -> 1024 @32x64 (forward / reflect
// adjoint, us): unsplit 47.9 / 67.2, variant 3 47.8 / 60.2, variant 4 46.3 / 56.9.  Four workgroups per tile with 8
// multiplying waves (4x64 px x 128 cout, 8x64 px x 64 cout: the lowest L2 -> LDS traffic per flop) ran 70 us: with 24
// K-steps per workgroup the pipeline fill and the hand-over (each way ~2 us to the memory side) outweigh the saving.
struct P3SplitCfg { int tr, tco, ncw, split; };
static constexpr P3SplitCfg P3_SPLIT_CFGS[2] = {
    {2, 128, 4, 2},   // 3: 2x64 px x 128 cout, 4 multiplying waves (pipelined), two workgroups per tile
    {4, 64, 4, 2},    // 4: 4x64 px x  64 cout, 4 multiplying waves (pipelined), two workgroups per tile
};
static int p3_split() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_CONV3X3P_SPLIT"); v = e ? atoi(e) : 1; }
    return v;
}

// variant: 0 = not applicable, 1 = 2x64 px x 64 cout (8 loader waves), 2 = 2x128 px x 128 cout (4 loader waves),
// 3, 4 (only with allow_split: the caller supplies a workspace) = P3_SPLIT_CFGS: several workgroups per tile splitting the
// input channels -- 64 x 64-per-wave tiles for layers that have too few such tiles to fill the chip
int conv3x3p_plan(const ir2rgb_conv_desc *d, P3Geom *g, int *npt_out, bool allow_split) {
    if (!p3_enabled() || d->transposed || d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1) return 0;
    if (d->pad_h != d->pad_w || d->pad_h < 0 || d->pad_h > 2 || d->out_f32) return 0;
    if (d->pad_mode < 0 || d->pad_mode > 2) return 0;
    const bool adj = d->pad_mode == 2;   // data gradient of a reflection-padded convolution (see border_terms)
    if (adj && (d->pad_h != 1 || (d->Hin & 1) || (d->Win % 64))) return 0;
    if (d->Hout != d->Hin + 2 * d->pad_h - 2 || d->Wout != d->Win + 2 * d->pad_w - 2) return 0;
    // (128-channel layers: the general kernel is faster forward (53 vs 56 us at 256x512), but the in-place reflection adjoint
    // beats its zero-padded convolution + fold pass: 54 us against 62 + 19)
    const int min_cin = adj && p3_min_cin() > 128 ? 128 : p3_min_cin();
    if ((d->Cin % 64) || d->Cin < min_cin || (d->Cout % 64) || d->Hin < 4 || d->Win < 4 || d->N < 1) return 0;
    if (d->dtype != IR2RGB_BF16 && d->dtype != IR2RGB_F16) return 0;
    const int ldx = d->ldx > 0 ? d->ldx : d->Cin, ldy = d->ldy > 0 ? d->ldy : d->Cout;
    if ((ldx & 7) || (d->ci_off & 7) || (ldy & 7) || (d->co_off & 7)) return 0;   // 16-byte loads and stores
    const long xb = (long)d->N * d->Hin * d->Win * ldx * 2, wb = (long)d->Cout * d->Cin * 9 * 2;
    if (xb >= (1L << 31) || wb >= (1L << 31)) return 0;
    auto waste = [](int n, int t) { return (double)(((n + t - 1) / t) * t) / n; };
    int variant = 0;
    const double wr = waste(d->Hout, 2);
    const long t128 = (long)d->N * ((d->Hout + 1) / 2) * ((d->Wout + 127) / 128) * (d->Cout / 128);
    const long t64 = (long)d->N * ((d->Hout + 1) / 2) * ((d->Wout + 63) / 64) * (d->Cout / 64);
    const long tmin = p3_min_tiles();
    if ((d->Cout % 128) == 0 && wr * waste(d->Wout, 128) <= 1.13 && t128 >= tmin && (!adj || d->Win % 128 == 0)) variant = 2;
    else if (wr * waste(d->Wout, 64) <= 1.13 && t64 >= tmin) variant = 1;
    if (!variant) return 0;
    // (split forms only where variant 1 would run: ir2rgb_conv2d_stats_rows() keeps that variant's rows, see the kernel)
    if (variant == 1 && allow_split && p3_split() && d->Cin >= 512) {
        static const int order[2] = {4, 3};
        for (int i = 0; i < 2; ++i) {
            const int v = p3_split() >= 3 ? p3_split() : order[i];
            if (v > 4) break;
            const P3SplitCfg &c = P3_SPLIT_CFGS[v - 3];
            const int kch = d->Cin / 64;
            const long wgs = (long)d->N * ((d->Hout + c.tr - 1) / c.tr) * ((d->Wout + 63) / 64) * (d->Cout / c.tco) * c.split;
            const bool ok = (d->Cout % c.tco) == 0 && (kch % c.split) == 0 && kch / c.split >= 2 && (!adj || (d->Hin % c.tr) == 0) &&
                            waste(d->Hout, c.tr) * waste(d->Wout, 64) <= 1.13 && wgs >= tmin &&
                            wgs * (c.ncw * 4096L * 4) < (1L << 30);   // (partial tiles addressed through one buffer resource)
            if (ok) { variant = v; break; }
            if (p3_split() >= 3) break;
        }
    }
    const int tr = variant >= 3 ? P3_SPLIT_CFGS[variant - 3].tr : 2;
    const int tw = variant == 2 ? 128 : 64;
    *g = P3Geom{};
    g->N = d->N; g->H = d->Hin; g->W = d->Win; g->Ho = d->Hout; g->Wo = d->Wout; g->Cin = d->Cin; g->Cout = d->Cout;
    g->pad = d->pad_h; g->pad_mode = d->pad_mode; g->act = d->act;
    g->ldx = ldx; g->ci_off = d->ci_off; g->ldy = ldy; g->co_off = d->co_off;
    g->stats_row0 = 0; g->nty = (d->Hout + tr - 1) / tr; g->ntx = (d->Wout + tw - 1) / tw;
    g->kchunks = d->Cin / 64;
    g->x_bytes = (unsigned)xb; g->w_bytes = (unsigned)wb;
    g->cout_major = wb > (long)d->N * d->Hin * d->Win * d->Cin * 2 ? 1 : 0;
#ifdef IR2RGB_ABLATION   // timing experiments only (tools/conv_ablate.py; build with -DIR2RGB_ABLATION): results are garbage
    { static int dbg = -1; if (dbg < 0) { const char *e = getenv("IR2RGB_CONV3X3P_DBG"); dbg = e ? atoi(e) : 0; } g->dbg = dbg; }
#endif
    if (d->pad_mode == 1 && (d->pad_h >= d->Hin || d->pad_w >= d->Win)) return 0;
    *npt_out = d->N * ((d->Hout + 1) / 2) * g->ntx;    // rows of the statistics buffer: 2-row tiles in every variant
    return variant;
}

// workspace of the split forms: [tiles] tickets (padded to 4 KB) + [tiles][split][waves][4][4][64] float4 partial accumulators
long conv3x3p_workspace_bytes(int variant, const P3Geom &g) {
    if (variant < 3) return 0;
    const P3SplitCfg &c = P3_SPLIT_CFGS[variant - 3];
    const long tiles = (long)g.N * g.nty * g.ntx * (g.Cout / c.tco);
    return ((tiles * 4 + 4095) & ~4095L) + tiles * c.split * (c.ncw * 4L * 4 * 4 * 64) * 4;
}

template <int DT, int TCO, int NCW, int PIPE, int SPLIT, int TR, int NLWF = 4>
static void p3_launch_split(bool adj, unsigned grid, const uint16_t *X, const uint16_t *W, const float *bias, uint16_t *Y,
                            float *stats, const P3Geom &g, unsigned *tickets, float *partials, hipStream_t s) {
    if (adj) conv3x3_patch_kernel<DT, 64, TCO, NCW, 4, PIPE, 1, 1, SPLIT, TR><<<grid, (NCW + 4) * 64, 0, s>>>(X, W, bias, Y, stats, g, tickets, partials);
    else conv3x3_patch_kernel<DT, 64, TCO, NCW, NLWF, PIPE, 0, 1, SPLIT, TR><<<grid, (NCW + NLWF) * 64, 0, s>>>(X, W, bias, Y, stats, g, tickets, partials);
}

template <int DT>
static void p3_launch_split_variant(int variant, bool adj, unsigned grid, const uint16_t *X, const uint16_t *W, const float *bias,
                                    uint16_t *Y, float *stats, const P3Geom &g, unsigned *tickets, float *partials, hipStream_t s) {
    if (variant == 3) p3_launch_split<DT, 128, 4, 1, 2, 2>(adj, grid, X, W, bias, Y, stats, g, tickets, partials, s);
    else {
        static int nlw8 = -1;
        if (nlw8 < 0) { const char *e = getenv("IR2RGB_CONV3X3P_NLW8"); nlw8 = e ? atoi(e) : 0; }
        if (nlw8) p3_launch_split<DT, 64, 4, 1, 2, 4, 8>(adj, grid, X, W, bias, Y, stats, g, tickets, partials, s);
        else p3_launch_split<DT, 64, 4, 1, 2, 4>(adj, grid, X, W, bias, Y, stats, g, tickets, partials, s);
    }
}

int conv3x3p_launch(int variant, const P3Geom &g, int dtype, const void *x, const void *wp, const float *bias, void *y,
                    float *stats, hipStream_t s, void *workspace, long workspace_bytes) {
    const uint16_t *X = (const uint16_t *)x, *W = (const uint16_t *)wp;
    uint16_t *Yp = (uint16_t *)y;
    const int npt = g.N * g.nty * g.ntx;
    const bool adj = g.pad_mode == 2;
    if (variant >= 3) {
        if (!workspace || workspace_bytes < conv3x3p_workspace_bytes(variant, g) || ((uintptr_t)workspace & 15)) return IR2RGB_EINVAL;
        const P3SplitCfg &c = P3_SPLIT_CFGS[variant - 3];
        const long tiles = (long)npt * (g.Cout / c.tco);
        unsigned *tickets = (unsigned *)workspace;
        float *partials = (float *)((char *)workspace + ((tiles * 4 + 4095) & ~4095L));
        const unsigned grid = (unsigned)(tiles * c.split);
        if (dtype == IR2RGB_BF16) p3_launch_split_variant<IR2RGB_BF16>(variant, adj, grid, X, W, bias, Yp, stats, g, tickets, partials, s);
        else p3_launch_split_variant<IR2RGB_F16>(variant, adj, grid, X, W, bias, Yp, stats, g, tickets, partials, s);
        return ir2rgb_launch_status();
    }
    if (variant == 2) {
        const unsigned grid = (unsigned)(npt * (g.Cout / 128));
        if (adj) {
            if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 128, 128, 8, 4, 0, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            else conv3x3_patch_kernel<IR2RGB_F16, 128, 128, 8, 4, 0, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
        } else {
            // (the fully pipelined form at this tile needs 2 x (MI + NI) fragment registers more than the 168 the 12-wave
            // workgroup leaves a wave: measured in round 2, 8 VGPRs spilled.  PIPE = 2, round 3: weights double-buffered,
            // pixel fragments refilled column by column; IR2RGB_CONV3X3P_LARGE_PIPE=0 selects the plain loop for A/B runs)
            static int lp = -1;
            if (lp < 0) { const char *e = getenv("IR2RGB_CONV3X3P_LARGE_PIPE"); lp = e ? atoi(e) : 1; }
            if (lp) {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 128, 128, 8, 4, 2, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 128, 128, 8, 4, 2, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            } else {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 128, 128, 8, 4, 0, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 128, 128, 8, 4, 0, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            }
        }
    } else {
        const unsigned grid = (unsigned)(npt * (g.Cout / 64));
        if (adj) {
            // software-pipelined like the forward twin (border operands prefetched into registers); IR2RGB_CONV3X3P_ADJ_PIPE=0
            // selects the plain loop (barrier at the top of every step) for A/B runs
            static int adj_pipe = -1;
            if (adj_pipe < 0) { const char *e = getenv("IR2RGB_CONV3X3P_ADJ_PIPE"); adj_pipe = e ? atoi(e) : 1; }
            if (adj_pipe) {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 64, 64, 4, 8, 1, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 64, 64, 4, 8, 1, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            } else {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 64, 64, 4, 8, 0, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 64, 64, 4, 8, 0, 1, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            }
        } else {
            if (p3_sps2()) {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 64, 64, 4, 8, 1, 0, 2><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 64, 64, 4, 8, 1, 0, 2><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            } else {
                if (dtype == IR2RGB_BF16) conv3x3_patch_kernel<IR2RGB_BF16, 64, 64, 4, 8, 1, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
                else conv3x3_patch_kernel<IR2RGB_F16, 64, 64, 4, 8, 1, 0, 1><<<grid, 768, 0, s>>>(X, W, bias, Yp, stats, g);
            }
        }
    }
    return ir2rgb_launch_status();
}
