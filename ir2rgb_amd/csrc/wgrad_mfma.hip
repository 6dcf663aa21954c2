// wgrad_mfma.hip -- weight gradient of the generator / discriminator convolutions on the gfx950
// matrix cores (what cuDNN's backward-filter does for the reference's loss.backward()).
//
//   D[tap][a][b] = sum_q U[q][a] * V[g(q,tap)][b]
//
// U [Q pixels][Ca] is the ungathered operand, V [.][Cb] the gathered one, both NHWC half:
//   Conv2d          : U = grad wrt conv output (a = cout), V = conv input (b = cin),
//                     g(q,tap) = (qy*stride + ky - pad, qx*stride + kx - pad)   -> dW[cout][cin][ky][kx]
//   ConvTranspose2d : U = conv input (a = cin), V = grad wrt conv output (b = cout),
//                     g(q,tap) = (qy*stride + ky - pad, qx*stride + kx - pad)   -> dW[cin][cout][ky][kx]
// The contraction index (pixels) is the SLOW index of both operands in memory, so both MFMA
// operands need a transpose: tiles are staged pixel-major into LDS with 16-byte LDS-DMA and the
// fragments are read with ds_read_b64_tr_b16 (hardware 4x16 transpose).  LDS rows are 256 B
// (128 channels); 32-byte channel pair P of row r is stored at pair P ^ f(r),
// f(r) = (r & 3) | ((r >> 3) & 1) << 2, which makes the eight rows touched by one transposed read
// (two 16-lane groups x 4 rows) hit 8 disjoint 32-byte bank windows.
//
// Workgroup = 256 threads (4 waves, 2x2), tile 128 (a) x 128 (b) for ONE tap and one K-split,
// K-step = 64 pixels, two LDS stages (64 KB -> 2 workgroups per CU).  Split-K partial tiles go to
// per-split fp32 slabs [split][tap][a][b] with plain stores (no float atomics: they run at ~1.3 TB/s
// chip-wide and are order-dependent); wgrad_finish sums the slabs and permutes into the torch weight
// layout -- deterministic.  Algorithmic flops = 2 * Q * Ca * Cb * ntaps; bound: MFMA.
#include <utility>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct WgradGeom {
    int N, Hq, Wq;          // pixel space of U (Q = N*Hq*Wq)
    int Hv, Wv;             // pixel space of V
    int Ca, Cb;
    int stride_y, stride_x, pad_mode;
    int nty, ntx, dy0, dx0; // tap (ty,tx): V coordinate = q*stride + (dy0+ty, dx0+tx)
    int ksplit, ksteps;     // K-steps (64 pixels each) in total and number of splits
    int use_atomics;
    int tpb;                // taps per workgroup: 2 when Cb <= 64 (the two halves of the 128-column V tile hold two taps)
    FastDiv div_hw, div_w;  // exact division by Hq*Wq and by Wq
    unsigned u_bytes, v_bytes;  // extents of U and V (buffer resources, < 2^31)
};

typedef __attribute__((address_space(3))) void *lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define WG_OOB 0x80000000u  // per-lane offset past any extent: the LDS-DMA delivers zeros

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// buffer-addressed LDS-DMA: 16 B per lane from base + 32-bit per-lane byte offset (range-checked, zeros
// when out of range) to LDS at wave-uniform base + lane * 16
__device__ __forceinline__ void dma16(rsrc_t r, unsigned voff, unsigned char *dst_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)dst_wave_base, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ int reflect1(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * n - 2 - v : v;
}

template <int DT> struct Mfma;
template <> struct Mfma<IR2RGB_BF16> {
    static __device__ __forceinline__ f32x4 run(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<IR2RGB_F16> {
    static __device__ __forceinline__ f32x4 run(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ s16x4 lds_tr(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
}

// SPLIT = 1: 512 threads, waves 0..3 multiply and waves 4..7 only stage (as in conv_wgrad3x3_kernel): the eight LDS-DMA
// issues of a K-step and their gather arithmetic leave the multiplying waves' instruction streams.  Measured on
// 1024x1024x3x3 @32x64: 119.9 -> 102.4 us; training window 42.6 -> 42.2 ms.  Default (IR2RGB_WGRAD_SPLIT=0: one role).
template <int DT, int SPLIT>
__global__ void __launch_bounds__(SPLIT ? 512 : 256, 2)
conv_wgrad_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, float *__restrict__ D,
                  const WgradGeom g) {
    constexpr int STAGE = 2 * 64 * 256;  // U tile + V tile, 64 pixel rows x 256 B each
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid0 = threadIdx.x, lane = tid0 & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const bool loader = SPLIT && wave8 >= 4;
    const int wave = wave8 & 3;            // role-local wave index
    const int tid = tid0 & 255;            // role-local thread index
    const int nta = (g.Ca + 127) >> 7, ntb = (g.Cb + 127) >> 7;
    // XCD-aware bijective remap (blocks b, b+8, ... share an L2): each XCD gets a contiguous run of
    // (split, tap, tile_a, tile_b) ids, i.e. few distinct U / V panels per private L2.
    int bid;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    // Thin layers (Cb <= 64: the full-resolution first / last layers, the 64-channel strided ones): the 128-column V tile
    // holds TWO taps' 64 channels, so the U tile is streamed once per tap pair instead of once per tap and no half of the
    // MFMA tile multiplies padding.  (One tap per workgroup re-read both operands for every tap: a 7x1 layer at
    // 64 channels, 512x1024, moved 7 x 134 MB.)
    const int ntaps = g.nty * g.ntx, ntg = (ntaps + g.tpb - 1) / g.tpb;
    const int tb = bid % ntb; bid /= ntb;
    const int ta = bid % nta; bid /= nta;
    const int tapg = bid % ntg;
    const int split = bid / ntg;
    const int a0 = ta * 128, b0 = tb * 128;
    const int per = (g.ksteps + g.ksplit - 1) / g.ksplit;
    const int kbeg = split * per, kend = min(g.ksteps, kbeg + per);
    if (kbeg >= kend) return;  // workgroup-uniform

    // ---------------- staging roles: 16 lanes per 256-B pixel row, 4 rows per wave instruction ----------------
    const int slot = tid & 15, r0 = tid >> 4;                   // rows r0 + 16*i, i = 0..3
    const int f = (r0 & 3) | (((r0 >> 3) & 1) << 2);            // same for r0 + 16*i
    const int chunk = ((((slot >> 1) ^ f) << 1) | (slot & 1));  // source 16-B chunk held by this LDS slot
    // this thread's V chunk: tap and channel offset (tpb == 2: chunks 0..7 = first tap of the pair, 8..15 = second)
    const int vtap = g.tpb == 2 ? tapg * 2 + (chunk >> 3) : tapg;
    const int vch = g.tpb == 2 ? (chunk & 7) * 8 : b0 + chunk * 8;
    const int ty = vtap / g.ntx, tx = vtap - ty * g.ntx;
    const int dy = g.dy0 + ty, dx = g.dx0 + tx;
    const bool a_ok = a0 + chunk * 8 < g.Ca, b_ok = vch < g.Cb && vtap < ntaps;
    const rsrc_t ru = make_rsrc(U, g.u_bytes), rv = make_rsrc(V, g.v_bytes);
    const unsigned ubase = a_ok ? (unsigned)(a0 + chunk * 8) * 2u : WG_OOB;  // byte offset inside a pixel row
    const unsigned vbase = b_ok ? (unsigned)vch * 2u : WG_OOB;
    const unsigned Q = (unsigned)g.N * g.Hq * g.Wq, HWq = (unsigned)g.Hq * g.Wq;
    const unsigned ca2 = (unsigned)g.Ca * 2u, cb2 = (unsigned)g.Cb * 2u;
    unsigned char *const wave_dst = smem + wave * 1024;  // + buf*STAGE + 4096*i (+16384 for V)
    const bool wide = g.Wq >= 16;  // rows r0+16*i of one K-step then wrap at most once per step of 16

    // one DMA pair (U row, gathered V row) for pixel q = (n, qy, qx); all selects, no branches
    auto issue_row = [&](unsigned q, unsigned n, unsigned qy, unsigned qx, unsigned char *dst) {
        const bool v = q < Q;
        dma16(ru, v ? ubase + q * ca2 : WG_OOB, dst);
        int iy = (int)qy * g.stride_y + dy, ix = (int)qx * g.stride_x + dx;
        const bool inb = ((unsigned)iy < (unsigned)g.Hv) & ((unsigned)ix < (unsigned)g.Wv);
        iy = g.pad_mode ? reflect1(iy, g.Hv) : iy;
        ix = g.pad_mode ? reflect1(ix, g.Wv) : ix;
        const unsigned vp = (n * (unsigned)g.Hv + (unsigned)iy) * (unsigned)g.Wv + (unsigned)ix;
        dma16(rv, (v && (g.pad_mode || inb)) ? vbase + vp * cb2 : WG_OOB, dst + 16384);
    };
    auto issue = [&](int ks, int buf) {
        unsigned char *dst = wave_dst + buf * STAGE;
        // pixel coordinates of this thread's first row by exact division ...
        unsigned q = (unsigned)ks * 64u + r0;
        unsigned qq = q < Q ? q : 0u;
        unsigned n = fdiv(qq, g.div_hw), rem = qq - n * HWq;
        unsigned qy = fdiv(rem, g.div_w), qx = rem - qy * g.Wq;
        if (wide) {  // ... the other three incrementally (a step of 16 pixels wraps at most one image row)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                issue_row(q, n, qy, qx, dst + 4096 * i);
                q += 16;
                qx += 16;
                const bool wr_ = qx >= (unsigned)g.Wq;
                qx = wr_ ? qx - g.Wq : qx;
                qy += wr_ ? 1u : 0u;
                const bool wy_ = qy >= (unsigned)g.Hq;
                qy = wy_ ? 0u : qy;
                n += wy_ ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                issue_row(q, n, qy, qx, dst + 4096 * i);
                q += 16;
                qq = q < Q ? q : 0u;
                n = fdiv(qq, g.div_hw); rem = qq - n * HWq;
                qy = fdiv(rem, g.div_w); qx = rem - qy * g.Wq;
            }
        }
    };

    // ---------------- compute roles ----------------
    const int wr = wave >> 1, wc = wave & 1;       // wave tile: rows a = wr*64.., cols b = wc*64..
    const int grp = lane >> 4, l15 = lane & 15;
    const int qd = l15 >> 2, pp = l15 & 3;          // transposed read: lane 4q+p -> LDS row q, channels 4p..4p+3
    // Row read by this lane in K32-step kk, half h: r = kk*32 + 4*h + (8*grp + qd).  Its swizzle term
    // f(r) = (r&3) | ((r>>3)&1)<<2 = qd | (grp&1)<<2 does not depend on (kk, h): a per-lane constant.  So
    //   address = [ (8*grp+qd)*256 + (pp>>1)*16 + (pp&1)*8 + ((col0>>4) ^ fr)*32 ]   per lane and per m
    //           + (kk*32 + 4*h)*256                                                  instruction immediate
    //           + tile base                                                          one add per K-step and m
    const int fr = qd | ((grp & 1) << 2);
    const int lane_off = (8 * grp + qd) * 256 + (pp >> 1) * 16 + (pp & 1) * 8;
    int aoff[4], boff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        aoff[m] = lane_off + ((((wr * 64 + m * 16) >> 4) ^ fr) << 5);
        boff[m] = lane_off + ((((wc * 64 + m * 16) >> 4) ^ fr) << 5) + 16384;
    }
    auto frag = [&](const unsigned char *p, int kk) -> s16x8 {
        const s16x4 lo = lds_tr(p + (kk * 32) * 256);
        const s16x4 hi = lds_tr(p + (kk * 32 + 4) * 256);
        return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (SPLIT) {
        if (loader) {
            issue(kbeg, 0);
            int lb = 0;
            for (int ks = kbeg; ks < kend; ++ks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step ks has landed
                __builtin_amdgcn_s_barrier();                         // ... and step ks-1 has been consumed
                if (ks + 1 < kend) issue(ks + 1, lb ^ 1);
                lb ^= 1;
            }
            return;
        }
    } else {
        issue(kbeg, 0);
    }
    int buf = 0;
    for (int ks = kbeg; ks < kend; ++ks) {
        if constexpr (!SPLIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (!SPLIT) {
            if (ks + 1 < kend) issue(ks + 1, buf ^ 1);
        }
        const unsigned char *tile = smem + buf * STAGE;
        const unsigned char *pa[4], *pb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { pa[m] = tile + aoff[m]; pb[m] = tile + boff[m]; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            s16x8 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = frag(pa[mi], kk);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = frag(pb[ni], kk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Mfma<DT>::run(a[mi], b[ni], acc[mi][ni]);
        }
        buf ^= 1;
    }

    // ---------------- epilogue: D[split][tap][a][b] (fp32 slab of this K-split) ----------------
    // (tpb == 2: the wave column wc is the tap of the pair, the channel is the column inside its 64)
    const int tap = g.tpb == 2 ? tapg * 2 + wc : tapg;
    if (tap >= ntaps) return;
    float *Dt = D + ((long)split * ntaps + tap) * g.Ca * g.Cb;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int b = g.tpb == 2 ? ni * 16 + l15 : b0 + wc * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + wr * 64 + mi * 16 + grp * 4 + r;
                if (a < g.Ca && b < g.Cb) {
                    Dt[(long)a * g.Cb + b] = acc[mi][ni][r];
                }
            }
        }
}

// ----------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (the residual blocks: most of the generator's weights): all nine
// taps of a 64 (a = cout) x 64 (b = cin) tile in ONE workgroup.
//
// The one-tap kernel above streams 32 KB of operands per 2 MFLOP and is bound by the LDS fill rate
// (a CU takes in ~55-90 GB/s by LDS-DMA).  The nine taps of a 3x3 kernel read the same gradient
// pixels and input pixels that are shifted by at most one row / one column, so a K-step here is one
// 64-pixel row segment: the gradient tile (64 px x 64 ch) plus a 3 x 66 pixel input patch with halo
// (33 KB) feed 9 x 64 x 64 x 64 MACs = 4.7 MFLOP.  The tap (ky,kx) operand is the patch read at row
// offset ky*66 + kx -- no extra staging.  Four waves split b (16 channels each) and share a (64):
// 36 accumulator tiles per wave, 26 transposed LDS reads per 36 MFMAs.
//
// LDS rows are 128 B (64 channels = four 32-byte pairs); pair P of row r is stored at P ^ f(r),
// f(r) = ((r>>1)&1) | ((r>>3)&1)<<1: the 8 rows {r0+0..3, r0+8..11} touched by two 16-lane groups of a
// transposed read then fall into 8 distinct 32-byte bank windows for ANY r0, which is what lets the
// tap offsets shift the read rows freely.
// 512 threads: four multiplying waves and four loader waves (one of each per SIMD).
// Four LDS stages of 36 KB (one workgroup per CU), three K-steps in flight; every loader wave issues exactly
// nine 1-KB LDS-DMA instructions per K-step (the last three of the 36 are padding into a dummy area)
// so the counted s_waitcnt immediates are uniform.
// With ksplit == 1 the tile is written straight into the torch layout [a][b][3][3] (each lane owns 9
// consecutive floats); otherwise into per-split slabs of that layout summed by wgrad_sum_kernel.
// ----------------------------------------------------------------------------------------
struct Wgrad9Geom {
    int N, H, W, Ca, Cb, pad_mode;
    int ksteps, ksplit, segs;    // segs = W / 64
    unsigned u_bytes, v_bytes;
};

#define W9_STAGE 36864
#define W9_NST 4

template <int IMM>
__device__ __forceinline__ s16x4 tr_read(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(IMM) : "memory");
    return v;
}

template <int DT>
__global__ void __launch_bounds__(512, 1)
conv_wgrad3x3_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, float *__restrict__ D,
                     const Wgrad9Geom g) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[W9_NST * W9_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0..3 multiply, waves 4..7 only stage: with one wave per SIMD the nine LDS-DMA issues (and their
    // address arithmetic) of a K-step would sit in front of that wave's MFMAs; a loader partner on the
    // same SIMD hides them.
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int ntb = g.Cb >> 6, nta = g.Ca >> 6;
    int bid = blockIdx.x;
    const int tb = bid % ntb; bid /= ntb;
    const int ta = bid % nta;
    const int split = bid / nta;
    const int a0 = ta * 64, b0 = tb * 64;
    const int per = (g.ksteps + g.ksplit - 1) / g.ksplit;
    const int kbeg = split * per, kend = min(g.ksteps, kbeg + per);

    // ---------------- staging: 9 LDS-DMA instructions per wave and K-step; instruction id = wave + 4*j ----------
    // id 0..7: gradient rows id*8 + (lane>>3); id 8..32: patch rows (id-8)*8 + (lane>>3) (< 198 valid);
    // id 33..35: padding.  Lane slot s = lane & 7 of a 128-B row holds source chunk ((s>>1) ^ f(r))*2 + (s&1).
    const rsrc_t ru = make_rsrc(U, g.u_bytes), rv = make_rsrc(V, g.v_bytes);
    const int rin = lane >> 3, sl = lane & 7;
    auto fsw = [](int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); };
    unsigned u_off[2];           // byte offset inside the 64-pixel segment of U, j = 0, 1
    // Patch row R -> (dy, dx) relative to (y, x0).  The gather offset of a lane is
    //   base(row, x0) + L0 + [edge fixes], L0 = (dy*W + dx)*Cb*2 + channel bytes,
    // where the fixes only apply on the image border (uniform per K-step: top / bottom row, first / last
    // segment): reflection moves dy = -1 -> +1 (top), +1 -> -1 (bottom), dx = -1 -> +1, 64 -> 62; zero
    // padding turns the lane off.  So a K-step costs a handful of VALU per DMA, no divisions.
    int v_L0[7], v_fixU[7], v_fixD[7], v_fixL[7], v_fixR[7], v_edge[7];
    bool v_dead[7];
    const int cb2i = g.Cb * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 4 * j) * 8 + rin;
        const int chunk = (((sl >> 1) ^ fsw(r)) << 1) | (sl & 1);
        u_off[j] = (unsigned)r * (unsigned)g.Ca * 2u + (unsigned)(a0 + chunk * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int id = wave + 4 * (j + 2);
        const int R = (id - 8) * 8 + rin;
        const bool real = id < 33 && R < 198;
        const int pr = R / 66, pxl = R - pr * 66;
        const int chunk = (((sl >> 1) ^ fsw(R)) << 1) | (sl & 1);
        const int dy = pr - 1, dx = pxl - 1;
        v_dead[j] = !real;
        v_L0[j] = (dy * g.W + dx) * cb2i + (b0 + chunk * 8) * 2;
        v_fixU[j] = dy == -1 ? 2 * g.W * cb2i : 0;
        v_fixD[j] = dy == 1 ? -2 * g.W * cb2i : 0;
        v_fixL[j] = dx == -1 ? 2 * cb2i : 0;
        v_fixR[j] = dx == 64 ? -2 * cb2i : 0;
        v_edge[j] = (dy == -1 ? 1 : 0) | (dy == 1 ? 2 : 0) | (dx == -1 ? 4 : 0) | (dx == 64 ? 8 : 0);
    }
    const unsigned ca2 = (unsigned)g.Ca * 2u;
    const bool refl = g.pad_mode != 0;
    // (row, seg) of the next K-step to issue, advanced incrementally; row = n*H + y
    int is_row = kbeg / g.segs, is_seg = kbeg - is_row * g.segs, is_y = is_row % g.H, is_ks = kbeg;
    auto issue = [&](int st) {
        unsigned char *dst = smem + st * W9_STAGE + wave * 1024;
        const bool live = is_ks < kend;                    // uniform; dead steps still issue (zeros) to keep counts
        const int x0 = is_seg * 64;
        const unsigned pix = (unsigned)is_row * (unsigned)g.W + (unsigned)x0;
        const unsigned ubase = pix * ca2;
        const int vbase = (int)(pix * (unsigned)cb2i);
        const bool top = is_y == 0, bot = is_y == g.H - 1, first = is_seg == 0, last = is_seg == g.segs - 1;
        const int edge = (top ? 1 : 0) | (bot ? 2 : 0) | (first ? 4 : 0) | (last ? 8 : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(ru, live ? ubase + u_off[j] : WG_OOB, dst + 4096 * j);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int o = vbase + v_L0[j] + (top ? v_fixU[j] : 0) + (bot ? v_fixD[j] : 0) + (first ? v_fixL[j] : 0) +
                          (last ? v_fixR[j] : 0);
            const bool off = v_dead[j] | !live | (!refl & ((v_edge[j] & edge) != 0));
            dma16(rv, off ? WG_OOB : (unsigned)o, dst + 4096 * (j + 2));
        }
        ++is_ks;
        if (++is_seg == g.segs) {
            is_seg = 0;
            ++is_row;
            if (++is_y == g.H) is_y = 0;
        }
    };

    // ---------------- compute roles ----------------
    const int grp = lane >> 4, l15 = lane & 15;
    const int qd = l15 >> 2, pp = l15 & 3;
    const int sub = (pp >> 1) * 16 + (pp & 1) * 8;
    const int rl = 8 * grp + qd;                       // row of this lane inside a 32-row K block (h = 0)
    int aoff[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = rl * 128 + ((mi ^ fsw(rl)) << 5) + sub;
    int boff[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int R = (t / 3) * 66 + (t % 3) + rl + 4 * h;
            boff[t][h] = 8192 + R * 128 + ((wave ^ fsw(R)) << 5) + sub;
        }

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) unsigned char *)smem);

    if (kbeg < kend) {
        if (loader) {
            issue(0);
            issue(1);
            issue(2);
            int st = 0;
            for (int ks = kbeg; ks < kend; ++ks) {
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // step ks has landed (ks+1, ks+2 may be in flight)
                __builtin_amdgcn_s_barrier();                         // ... and step ks-1 has been consumed
                issue((st + 3) & 3);
                st = (st + 1) & 3;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the padding DMAs before LDS goes away
        } else {
            int st = 0;
            for (int ks = kbeg; ks < kend; ++ks) {
                __builtin_amdgcn_s_barrier();
                // Fragment reads are inline asm: the compiler models the ds_read_tr builtin as a possible LDS
                // write and would put s_waitcnt vmcnt(0) in front of it (draining the three K-steps of LDS-DMA
                // in flight).  Hence the explicit lgkmcnt waits below; each names the registers it makes valid.
                const unsigned tbase = lds_base + (unsigned)st * W9_STAGE;
                unsigned av[4], bv[9][2];
    #pragma unroll
                for (int mi = 0; mi < 4; ++mi) av[mi] = tbase + (unsigned)aoff[mi];
    #pragma unroll
                for (int t = 0; t < 9; ++t) { bv[t][0] = tbase + (unsigned)boff[t][0]; bv[t][1] = tbase + (unsigned)boff[t][1]; }
                // 18 fragment steps s = kk*9 + tap, B fragments fetched three steps ahead (LDS latency is
                // ~2 steps of 4 MFMAs), the A fragments of the second half fetched during step 5.
                s16x4 alo[2][4], ahi[2][4], blo[4], bhi[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    alo[0][mi] = tr_read<0>(av[mi]);
                    ahi[0][mi] = tr_read<512>(av[mi]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    blo[q] = tr_read<0>(bv[q][0]);
                    bhi[q] = tr_read<0>(bv[q][1]);
                }
                auto step = [&]<int S>(std::integral_constant<int, S>) {
                    constexpr int KK = S / 9, T = S % 9, cur = S & 3;
                    if constexpr (S == 5) {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            alo[1][mi] = tr_read<4096>(av[mi]);
                            ahi[1][mi] = tr_read<4096 + 512>(av[mi]);
                        }
                    }
                    if constexpr (S + 3 < 18) {
                        constexpr int S3 = S + 3;
                        blo[S3 & 3] = tr_read<(S3 / 9) * 4096>(bv[S3 % 9][0]);
                        bhi[S3 & 3] = tr_read<(S3 / 9) * 4096>(bv[S3 % 9][1]);
                    }
                    // LDS reads issued after B(S): B(S+1..S+3), plus the 8 reads of the second A set in steps 5..8
                    constexpr int after = 2 * ((S + 3 < 18 ? 3 : 17 - S)) + ((S >= 5 && S <= 8) ? 8 : 0);
                    if constexpr (T == 0)
                        asm volatile("s_waitcnt lgkmcnt(%10)"
                                     : "+v"(alo[KK][0]), "+v"(ahi[KK][0]), "+v"(alo[KK][1]), "+v"(ahi[KK][1]), "+v"(alo[KK][2]),
                                       "+v"(ahi[KK][2]), "+v"(alo[KK][3]), "+v"(ahi[KK][3]), "+v"(blo[cur]), "+v"(bhi[cur])
                                     : "n"(after)
                                     : "memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blo[cur]), "+v"(bhi[cur]) : "n"(after) : "memory");
                    const s16x8 b = (s16x8){blo[cur][0], blo[cur][1], blo[cur][2], blo[cur][3],
                                            bhi[cur][0], bhi[cur][1], bhi[cur][2], bhi[cur][3]};
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const s16x8 a = (s16x8){alo[KK][mi][0], alo[KK][mi][1], alo[KK][mi][2], alo[KK][mi][3],
                                                ahi[KK][mi][0], ahi[KK][mi][1], ahi[KK][mi][2], ahi[KK][mi][3]};
                        acc[T][mi] = Mfma<DT>::run(a, b, acc[T][mi]);
                    }
                };
                [&]<int... Ss>(std::integer_sequence<int, Ss...>) { (step(std::integral_constant<int, Ss>{}), ...); }
                (std::make_integer_sequence<int, 18>{});
                st = (st + 1) & 3;
            }
        }
    }
    if (loader) return;

    // ---------------- epilogue: [a][b][tap], 9 consecutive floats per (a, b) ----------------
    float *Dt = D + (long)split * 9 * g.Ca * g.Cb;
    const int b = b0 + wave * 16 + l15;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = a0 + mi * 16 + grp * 4 + r;
            float *o = Dt + ((long)a * g.Cb + b) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) o[t] = acc[t][mi][r];
        }
}

// ----------------------------------------------------------------------------------------
// k x 1 / 1 x k convolutions at 64-channel tiles (the generators' separable 7x7 first layers and heads, the
// discriminators' 4x1 stride-2 first layers: full-resolution tensors, tiny weights): ALL taps of a 64 (a) x 64 (b) tile
// in one workgroup, as conv_wgrad3x3_kernel does for 3x3.  The one-tap kernel (even with two taps per workgroup)
// streams the gradient tensor once per tap pair and multiplies a half-empty 128-row tile: 202 us for a 7x1 layer at
// 64 channels and 512x1024, where both tensors are 27 us of HBM time and the products 12 us of MFMA time.
//
// A K-step = one 64-pixel segment of one output row: the gradient tile (64 px x 64 ch, rows 0..63 of the stage) and
// the NTY x (64 + NTX - 1) pixel input patch behind it (rows 64..); tap (ty, tx) is the patch read at row offset
// ty * PW + tx.  LDS layout, swizzle and the transposed fragment reads are conv_wgrad3x3_kernel's (128-byte rows).
// 512 threads: waves 0..3 multiply (16 b-channels each, all 64 a-channels: NT x 4 accumulator tiles), waves 4..7
// stage the next K-step by LDS-DMA into the other of two stages (the multipliers never issue a DMA, so the
// compiler's conservative vmcnt waits around the transposed reads cost nothing).  Output: per-split slabs
// [split][tap][a][b], summed by the finish pass in a fixed order -- no atomics.
// ----------------------------------------------------------------------------------------
struct WgradLineGeom {
    int N, Hq, Wq, Hv, Wv, Ca, Cb;
    int stride_y, pad_mode, dy0, dx0;
    int segs, ksteps, ksplit, per;     // K-steps = N * Hq * segs, `per` of them per split
    unsigned u_bytes, v_bytes;
};

// SX = 2 (3x3 stride-2 layers, plain and transposed: the down- / up-samplers): output pixel i of a segment reads input
// columns 2i + tx + const, i.e. every other pixel -- but the transposed fragment reads want the 32 pixels of a K block in
// consecutive LDS rows.  So the patch is staged split by column parity: run (ty, parity) holds the columns
// cb + 2 * idx + parity, idx = 0 .. PWp - 1, and tap (ty, tx) is run (ty, tx & 1) read from idx offset tx >> 1.
template <int DT, int NTY, int NTX, int SX = 1>
__global__ void __launch_bounds__(512, 1)
conv_wgrad_line_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, float *__restrict__ D,
                       const WgradLineGeom g) {
    constexpr int NT = NTY * NTX;
    constexpr int PW = SX == 2 ? 64 + ((NTX - 1) >> 1) : 64 + NTX - 1;     // pixels per patch run
    constexpr int PR = NTY * (SX == 2 ? 2 : 1) * PW;
    constexpr int NID = 8 + (PR + 7) / 8;          // 1-KB DMA instructions (8 rows each) per K-step: 8 gradient, the rest patch
    constexpr int J = (NID + 3) / 4;               // ... per loader wave
    constexpr int STAGE = NID * 1024;
    static_assert(2 * STAGE <= 160 * 1024 - 1024, "two stages must fit the LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int ntb = g.Cb >> 6, nta = g.Ca >> 6;
    int bid = blockIdx.x;
    const int tb = bid % ntb; bid /= ntb;
    const int ta = bid % nta;
    const int split = bid / nta;
    const int a0 = ta * 64, b0 = tb * 64;
    const int kbeg = split * g.per, kend = min(g.ksteps, kbeg + g.per);   // (the host leaves no split empty)
    auto fsw = [](int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); };

    if (loader) {
        const rsrc_t ru = make_rsrc(U, g.u_bytes), rv = make_rsrc(V, g.v_bytes);
        const int rin = lane >> 3, sl = lane & 7;
        const unsigned ca2 = (unsigned)g.Ca * 2u, cb2 = (unsigned)g.Cb * 2u;
        // (row, seg) of the next K-step to issue, advanced incrementally; row = n * Hq + qy
        int is_row = kbeg / g.segs, is_seg = kbeg - is_row * g.segs, is_n = is_row / g.Hq, is_y = is_row - is_n * g.Hq;
        auto issue = [&](int buf) {
            unsigned char *dst = smem + buf * STAGE + wave * 1024;
            const int x0 = is_seg * 64;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int id = wave + 4 * j;           // wave-uniform
                if (id >= NID) break;
                unsigned off = WG_OOB;
                if (id < 8) {                           // gradient rows: output pixels x0 + r of this row
                    const int r = id * 8 + rin;
                    const int chunk = (((sl >> 1) ^ fsw(r)) << 1) | (sl & 1);
                    if (x0 + r < g.Wq)
                        off = ((unsigned)is_row * (unsigned)g.Wq + (unsigned)(x0 + r)) * ca2 + (unsigned)(a0 + chunk * 8) * 2u;
                    dma16(ru, off, dst + 4096 * j);
                } else {                                // patch rows: tap row pr, column pxl
                    const int R = (id - 8) * 8 + rin;
                    const int run = R / PW, pxl = R - run * PW;
                    const int pr = SX == 2 ? run >> 1 : run;
                    const int chunk = (((sl >> 1) ^ fsw(R)) << 1) | (sl & 1);
                    int iy = is_y * g.stride_y + g.dy0 + pr;
                    int ix = SX == 2 ? 2 * x0 + g.dx0 + 2 * pxl + (run & 1) : x0 + pxl + g.dx0;
                    const bool inb = ((unsigned)iy < (unsigned)g.Hv) & ((unsigned)ix < (unsigned)g.Wv);
                    iy = g.pad_mode ? reflect1(iy, g.Hv) : iy;
                    ix = g.pad_mode ? reflect1(ix, g.Wv) : ix;
                    // (reflection of a column far right of a ragged last segment can still leave the row: such patch
                    // columns only ever meet gradient rows that are zero, any in-range address will do)
                    ix = min(max(ix, 0), g.Wv - 1);
                    iy = min(max(iy, 0), g.Hv - 1);
                    if (R < PR && (g.pad_mode || inb))
                        off = (((unsigned)is_n * (unsigned)g.Hv + (unsigned)iy) * (unsigned)g.Wv + (unsigned)ix) * cb2 +
                              (unsigned)(b0 + chunk * 8) * 2u;
                    dma16(rv, off, dst + 4096 * j);
                }
            }
            if (++is_seg == g.segs) {
                is_seg = 0;
                ++is_row;
                if (++is_y == g.Hq) { is_y = 0; ++is_n; }
            }
        };
        issue(0);
        int lb = 0;
        for (int ks = kbeg; ks < kend; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step ks has landed
            __builtin_amdgcn_s_barrier();                         // ... and step ks-1 has been consumed
            if (ks + 1 < kend) issue(lb ^ 1);
            lb ^= 1;
        }
        return;
    }

    // ---------------- multiplying waves ----------------
    const int grp = lane >> 4, l15 = lane & 15;
    const int qd = l15 >> 2, pp = l15 & 3;
    const int sub = (pp >> 1) * 16 + (pp & 1) * 8;
    const int rl = 8 * grp + qd;                       // row of this lane inside a 32-row K block (h = 0)
    int aoff[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = rl * 128 + ((mi ^ fsw(rl)) << 5) + sub;
    int boff[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ty = t / NTX, tx = t % NTX;
            const int R = (SX == 2 ? (ty * 2 + (tx & 1)) * PW + (tx >> 1) : ty * PW + tx) + rl + 4 * h;
            boff[t][h] = 8192 + R * 128 + ((wave ^ fsw(R)) << 5) + sub;
        }
    f32x4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int ks = kbeg; ks < kend; ++ks) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *tile = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            s16x8 a[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const s16x4 lo = lds_tr(tile + aoff[mi] + kk * 4096), hi = lds_tr(tile + aoff[mi] + kk * 4096 + 512);
                a[mi] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const s16x4 lo = lds_tr(tile + boff[t][0] + kk * 4096), hi = lds_tr(tile + boff[t][1] + kk * 4096);
                const s16x8 b = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[t][mi] = Mfma<DT>::run(a[mi], b, acc[t][mi]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage have returned before it is handed back
        buf ^= 1;
    }
    // ---------------- epilogue: D[split][tap][a][b] ----------------
    const int b = b0 + wave * 16 + l15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float *Dt = D + ((long)split * NT + t) * g.Ca * g.Cb;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) Dt[(long)(a0 + mi * 16 + grp * 4 + r) * g.Cb + b] = acc[t][mi][r];
    }
}

// The k x 1 layers again, walking DOWN the image: consecutive K-steps are the same 64-pixel column segment of consecutive
// output rows, whose NTY-row input patches overlap in all but SY rows.  The patch lives in a ring of NTY + SY row slots
// (64 pixels x 128 B each) per column parity; a K-step stages its gradient segment and only the SY input rows that are
// new (all NTY at the top of a column or of a split, into the OTHER parity's ring, so that the rows the multiplying waves
// are reading are never touched).  conv_wgrad_line_kernel restaged the whole patch per step and ran at the LDS fill
// rate: 7x1 at 64 channels, 512x1024: 85 us; here the operands enter the LDS once.
template <int DT, int NTY, int SY>
__global__ void __launch_bounds__(512, 1)
conv_wgrad_col_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, float *__restrict__ D,
                      const WgradLineGeom g) {
    constexpr int RC = NTY + SY;                   // ring slots per column parity
    constexpr int PATCH0 = 16384;                  // two 8-KB gradient buffers in front
    constexpr int LDS = PATCH0 + 2 * RC * 8192;
    constexpr int JMAX = (8 + NTY * 8 + 3) / 4;    // DMA instructions per loader wave of a priming step
    static_assert(LDS <= 160 * 1024 - 1024, "ring must fit the LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int ntb = g.Cb >> 6, nta = g.Ca >> 6;
    int bid = blockIdx.x;
    const int tb = bid % ntb; bid /= ntb;
    const int ta = bid % nta;
    const int split = bid / nta;
    const int a0 = ta * 64, b0 = tb * 64;
    const int kbeg = split * g.per, kend = min(g.ksteps, kbeg + g.per);   // K-step ks = (column = n * segs + seg, row y), y fastest
    auto fsw = [](int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); };

    if (loader) {
        const rsrc_t ru = make_rsrc(U, g.u_bytes), rv = make_rsrc(V, g.v_bytes);
        const int rin = lane >> 3, sl = lane & 7;
        const unsigned ca2 = (unsigned)g.Ca * 2u, cb2 = (unsigned)g.Cb * 2u;
        int col = kbeg / g.Hq, y = kbeg - col * g.Hq, n = col / g.segs, seg = col - n * g.segs, ub = 0;
        bool first = true;
        auto issue = [&]() {
            const bool prime = first || y == 0;
            const int x0 = seg * 64, half = col & 1;
            const int total = 8 + (prime ? NTY : SY) * 8;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int id = wave + 4 * j;           // wave-uniform
                if (id >= total) break;
                unsigned off = WG_OOB;
                if (id < 8) {                           // gradient pixels x0 + r of output row y
                    const int r = id * 8 + rin;
                    const int chunk = (((sl >> 1) ^ fsw(r)) << 1) | (sl & 1);
                    if (x0 + r < g.Wq)
                        off = (((unsigned)n * (unsigned)g.Hq + (unsigned)y) * (unsigned)g.Wq + (unsigned)(x0 + r)) * ca2 +
                              (unsigned)(a0 + chunk * 8) * 2u;
                    dma16(ru, off, smem + ub * 8192 + id * 1024);
                } else {                                // input row y * SY + jt (virtual, before padding), pixels x0 + r
                    const int q = id - 8;
                    const int jt = (prime ? 0 : NTY - SY) + (q >> 3);
                    const int r = (q & 7) * 8 + rin;
                    const int chunk = (((sl >> 1) ^ fsw(r)) << 1) | (sl & 1);
                    const int v = y * SY + jt;
                    int iy = v + g.dy0;
                    const int ix = x0 + r;
                    const bool inb = ((unsigned)iy < (unsigned)g.Hv) & (ix < g.Wv);
                    iy = g.pad_mode ? reflect1(iy, g.Hv) : iy;
                    iy = min(max(iy, 0), g.Hv - 1);
                    if (ix < g.Wv && (g.pad_mode || inb))
                        off = (((unsigned)n * (unsigned)g.Hv + (unsigned)iy) * (unsigned)g.Wv + (unsigned)ix) * cb2 +
                              (unsigned)(b0 + chunk * 8) * 2u;
                    const int slot = half * RC + v % RC;
                    dma16(rv, off, smem + PATCH0 + slot * 8192 + (q & 7) * 1024);
                }
            }
            first = false;
            ub ^= 1;
            if (++y == g.Hq) {
                y = 0;
                ++col;
                if (++seg == g.segs) { seg = 0; ++n; }
            }
        };
        issue();
        for (int ks = kbeg; ks < kend; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step ks has landed
            __builtin_amdgcn_s_barrier();                         // ... and step ks-1 has been consumed
            if (ks + 1 < kend) issue();
        }
        return;
    }

    // ---------------- multiplying waves ----------------
    const int grp = lane >> 4, l15 = lane & 15;
    const int qd = l15 >> 2, pp = l15 & 3;
    const int sub = (pp >> 1) * 16 + (pp & 1) * 8;
    const int rl = 8 * grp + qd;
    int aoff[4], boffl[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = rl * 128 + ((mi ^ fsw(rl)) << 5) + sub;
#pragma unroll
    for (int h = 0; h < 2; ++h) boffl[h] = PATCH0 + (rl + 4 * h) * 128 + ((wave ^ fsw(rl + 4 * h)) << 5) + sub;
    f32x4 acc[NTY][4];
#pragma unroll
    for (int t = 0; t < NTY; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int col = kbeg / g.Hq, y = kbeg - col * g.Hq, ub = 0;
    for (int ks = kbeg; ks < kend; ++ks) {
        __builtin_amdgcn_s_barrier();
        const unsigned char *ut = smem + ub * 8192;
        const int base = (col & 1) * RC, v0 = (y * SY) % RC;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            s16x8 a[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const s16x4 lo = lds_tr(ut + aoff[mi] + kk * 4096), hi = lds_tr(ut + aoff[mi] + kk * 4096 + 512);
                a[mi] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int t = 0; t < NTY; ++t) {
                int slot = v0 + t;
                slot = slot >= RC ? slot - RC : slot;
                const unsigned char *pt = smem + (base + slot) * 8192 + kk * 4096;
                const s16x4 lo = lds_tr(pt + boffl[0]), hi = lds_tr(pt + boffl[1]);
                const s16x8 b = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[t][mi] = Mfma<DT>::run(a[mi], b, acc[t][mi]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads have returned before the buffers are handed back
        ub ^= 1;
        if (++y == g.Hq) { y = 0; ++col; }
    }
    const int b = b0 + wave * 16 + l15;
#pragma unroll
    for (int t = 0; t < NTY; ++t) {
        float *Dt = D + ((long)split * NTY + t) * g.Ca * g.Cb;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) Dt[(long)(a0 + mi * 16 + grp * 4 + r) * g.Cb + b] = acc[t][mi][r];
    }
}

// out[i] = sum_split D[split][i]   (slabs already in the torch layout)
__global__ void __launch_bounds__(256)
wgrad_sum_kernel(const float4 *__restrict__ D, float4 *__restrict__ out, long n4, int nsplit) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = D[i];
#pragma unroll 4
        for (int k = 1; k < nsplit; ++k) {
            const float4 v = D[k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[i] = s;
    }
}

static bool plan9(const ir2rgb_conv_desc *d, Wgrad9Geom *g) {
    if (d->transposed || d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1)
        return false;
    if (d->Hout != d->Hin || d->Wout != d->Win || (d->Win % 64) || (d->Cin % 64) || (d->Cout % 64) || d->Hin < 2) return false;
    const long Q = (long)d->N * d->Hin * d->Win;
    if (Q * d->Cout * 2 >= (1L << 31) || Q * d->Cin * 2 >= (1L << 31)) return false;
    *g = Wgrad9Geom{};
    g->N = d->N; g->H = d->Hin; g->W = d->Win; g->Ca = d->Cout; g->Cb = d->Cin; g->pad_mode = d->pad_mode;
    g->segs = d->Win / 64;
    g->ksteps = (int)(Q / 64);
    g->u_bytes = (unsigned)(Q * d->Cout * 2); g->v_bytes = (unsigned)(Q * d->Cin * 2);
    // one workgroup per CU: split K until ~256 workgroups exist, keeping >= 8 K-steps per split
    const long tiles = (long)(d->Cout / 64) * (d->Cin / 64);
    long ks = (256 + tiles - 1) / tiles;
    if (ks > g->ksteps / 8) ks = g->ksteps / 8;
    if (ks < 1) ks = 1;
    if (ks > 256) ks = 256;
    g->ksplit = (int)ks;
    return true;
}

// out[a][b][tap] = sum_split D[split][tap][a][b]   (torch weight layout [Ca][Cb][kh][kw], fp32)
__global__ void __launch_bounds__(256)
wgrad_finish_kernel(const float *__restrict__ D, float *__restrict__ out, int Ca, int Cb, int ntaps, int nsplit,
                    long total, int acc) {
    const long slab = (long)ntaps * Ca * Cb;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % ntaps);
        const long ab = i / ntaps;
        const float *p = D + (long)tap * Ca * Cb + ab;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += p[k * slab];
        out[i] = acc ? out[i] + s : s;
    }
}

// The same sums (same order over the splits: bit-identical) with coalesced reads: a block owns 64 consecutive
// (a, b) pairs and all taps; wave w reads the taps w, w+4, ... of its 64 pairs (256 contiguous bytes per
// split), the sums are transposed through LDS and leave as one contiguous run of 64 * ntaps floats.  The
// gather form above reads one 4-byte element per lane from ntaps different planes.
#define WF_MAX_TAPS 16
__global__ void __launch_bounds__(256)
wgrad_finish_tiled_kernel(const float *__restrict__ D, float *__restrict__ out, long AB, int ntaps, int nsplit, int acc) {
    __shared__ float t[64 * (WF_MAX_TAPS + 1)];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long ab0 = blockIdx.x * 64L, ab = ab0 + l;
    const long slab = (long)ntaps * AB;
    for (int tap = w; tap < ntaps; tap += 4) {
        float s = 0.f;
        if (ab < AB) {
            const float *p = D + (long)tap * AB + ab;
#pragma unroll 8
            for (int k = 0; k < nsplit; ++k) s += p[k * slab];
        }
        t[l * (ntaps + 1) + tap] = s;
    }
    __syncthreads();
    const long left = AB - ab0;
    const int n = (int)(left < 64 ? left : 64) * ntaps;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float v = t[(j / ntaps) * (ntaps + 1) + j % ntaps];
        out[ab0 * ntaps + j] = acc ? out[ab0 * ntaps + j] + v : v;      // acc: dw += (a later use of the same parameter)
    }
}

// Many splits (the line kernel: up to 256): one workgroup per (64 (a, b) pairs, tap); wave w sums the splits w, w+4, ...
// (64 coalesced loads each instead of 256 dependent ones per thread), the four partial sums are added in wave order.
__global__ void __launch_bounds__(256)
wgrad_finish_wide_kernel(const float *__restrict__ D, float *__restrict__ out, long AB, int ntaps, int nsplit, int acc) {
    __shared__ float part[4][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, tap = blockIdx.y;
    const long ab = blockIdx.x * 64L + l;
    const long slab = (long)ntaps * AB;
    float s = 0.f;
    if (ab < AB) {
        const float *p = D + (long)tap * AB + ab;
#pragma unroll 8
        for (int k = w; k < nsplit; k += 4) s += p[k * slab];
    }
    part[w][l] = s;
    __syncthreads();
    if (w == 0 && ab < AB) {
        const float v = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
        const long o = ab * ntaps + tap;
        out[o] = acc ? out[o] + v : v;
    }
}

static void launch_wgrad_finish(const float *D, float *out, int Ca, int Cb, int ntaps, int nsplit, hipStream_t s, int acc = 0) {
    const long AB = (long)Ca * Cb, elems = AB * ntaps;
    if (nsplit > 64 && (AB + 63) / 64 <= 0x7fffffffL && ntaps <= 65535)
        wgrad_finish_wide_kernel<<<dim3((unsigned)((AB + 63) / 64), (unsigned)ntaps), 256, 0, s>>>(D, out, AB, ntaps, nsplit, acc);
    else if (ntaps <= WF_MAX_TAPS && (AB + 63) / 64 <= 0x7fffffffL)
        wgrad_finish_tiled_kernel<<<(unsigned)((AB + 63) / 64), 256, 0, s>>>(D, out, AB, ntaps, nsplit, acc);
    else
        wgrad_finish_kernel<<<stream_grid(elems, 256), 256, 0, s>>>(D, out, Ca, Cb, ntaps, nsplit, elems, acc);
}

static int plan(const ir2rgb_conv_desc *d, WgradGeom *g) {
    if (!d || d->N < 1 || d->Cin < 8 || d->Cout < 8 || (d->Cin % 8) || (d->Cout % 8) || d->kh < 1 || d->kw < 1 ||
        d->stride_h < 1 || d->stride_w < 1 || d->pad_h < 0 || d->pad_w < 0)
        return IR2RGB_EINVAL;
    if (d->dtype != IR2RGB_BF16 && d->dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (d->transposed && d->pad_mode != 0) return IR2RGB_ENOSUP;
    *g = WgradGeom{};
    g->N = d->N;
    if (!d->transposed) {  // U = grad output [N,Hout,Wout,Cout], V = input [N,Hin,Win,Cin]
        g->Hq = d->Hout; g->Wq = d->Wout; g->Hv = d->Hin; g->Wv = d->Win; g->Ca = d->Cout; g->Cb = d->Cin;
    } else {               // U = input [N,Hin,Win,Cin], V = grad output [N,Hout,Wout,Cout]
        g->Hq = d->Hin; g->Wq = d->Win; g->Hv = d->Hout; g->Wv = d->Wout; g->Ca = d->Cin; g->Cb = d->Cout;
    }
    g->stride_y = d->stride_h; g->stride_x = d->stride_w; g->pad_mode = d->pad_mode;
    g->nty = d->kh; g->ntx = d->kw; g->dy0 = -d->pad_h; g->dx0 = -d->pad_w;
    long Q = (long)g->N * g->Hq * g->Wq;
    if (Q >= (1L << 31) || (long)g->N * g->Hv * g->Wv >= (1L << 31)) return IR2RGB_EINVAL;
    g->ksteps = (int)((Q + 63) / 64);
    // K-split by a cost model: rounds of 512 resident workgroups (2 per CU) x (K-steps per split + pipeline
    // fill) against the extra slab traffic of the finish pass (split x elems x 8 bytes at ~4 TB/s).
    {
        static int pairs = -1;
        if (pairs < 0) { const char *e = getenv("IR2RGB_WGRAD_TAP_PAIRS"); pairs = e ? atoi(e) : 1; }
        g->tpb = (pairs && g->Cb <= 64 && d->kh * d->kw > 1) ? 2 : 1;
    }
    const long tiles = (long)((d->kh * d->kw + g->tpb - 1) / g->tpb) * ((g->Ca + 127) / 128) * ((g->Cb + 127) / 128);
    const double elems = (double)d->kh * d->kw * g->Ca * g->Cb;
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= 64; ++ks) {
        if (ks > 1 && g->ksteps / ks < 4) break;               // at least 4 K-steps per split
        const double rounds = (double)((tiles * ks + 511) / 512);
        const double cost = rounds * ((g->ksteps + ks - 1) / ks + 3) * 1.2 + ks * elems * 8.0 / 4e6;  // microseconds
        if (cost < best_cost) { best_cost = cost; best = ks; }
    }
    g->ksplit = best;
    g->use_atomics = 0;
    g->div_hw = make_fastdiv((unsigned)(g->Hq * g->Wq)); g->div_w = make_fastdiv((unsigned)g->Wq);
    {
        const long ub = Q * g->Ca * 2, vb = (long)g->N * g->Hv * g->Wv * g->Cb * 2;
        if (ub >= (1L << 31) || vb >= (1L << 31)) return IR2RGB_EINVAL;  // 32-bit buffer offsets
        g->u_bytes = (unsigned)ub; g->v_bytes = (unsigned)vb;
    }
    return IR2RGB_OK;
}

// k x 1 / 1 x k layers on conv_wgrad_line_kernel: 7x1, 1x7 (stride 1 along the taps' axis is not required: the taps run
// along y for k x 1, where the stride enters the row index) and the stride-2 4x1; 64-multiples on both channel axes.
static bool plan_line(const ir2rgb_conv_desc *d, WgradLineGeom *g) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("IR2RGB_WGRAD_LINE"); on = e ? atoi(e) : 1; }
    if (!on || (d->Cin % 64) || (d->Cout % 64)) return false;
    static int s2 = -1;             // IR2RGB_WGRAD_S2=0: the stride-2 3x3 layers on the one-tap kernel (A/B measurements)
    if (s2 < 0) { const char *e = getenv("IR2RGB_WGRAD_S2"); s2 = e ? atoi(e) : 1; }
    // (measured against the one-tap kernel, us at the training sizes: 64->128 @512x1024 49 vs 60, 512->1024 @64x128 54 vs 60,
    // 1024->512 transposed 53 vs 58, 128->64 transposed 49 vs 56 -- but 128->256 52 vs 43, 256->512 48 vs 43: every 64 x 64
    // tile restages the 57-KB patch, which only pays at the two ends of the channel range; s2 = 2 forces it everywhere)
    const long tl = (long)(d->Cout / 64) * (d->Cin / 64);
    const bool k3s2 = s2 && d->kh == 3 && d->kw == 3 && d->stride_h == 2 && d->stride_w == 2 && d->pad_mode == 0 &&
                      (s2 == 2 || tl <= 2 || tl >= 128);
    const bool k7x1 = d->kh == 7 && d->kw == 1, k1x7 = d->kh == 1 && d->kw == 7, k4x1 = d->kh == 4 && d->kw == 1;
    if (!k3s2 && (d->transposed || d->stride_w != 1)) return false;
    if (!(k7x1 || k1x7 || k4x1 || k3s2)) return false;
    if (k7x1 && d->pad_w != 0) return false;
    if (k4x1 && d->pad_w != 0) return false;
    if (d->pad_mode != 0 && d->pad_mode != 1) return false;
    if (d->pad_mode == 1 && (d->pad_h >= d->Hin || d->pad_w >= d->Win)) return false;
    *g = WgradLineGeom{};
    g->N = d->N;
    if (!d->transposed) {   // U = gradient of the output (a = cout), V = input (b = cin)
        g->Hq = d->Hout; g->Wq = d->Wout; g->Hv = d->Hin; g->Wv = d->Win; g->Ca = d->Cout; g->Cb = d->Cin;
    } else {                // ConvTranspose2d: U = input (a = cin), V = gradient of the output (b = cout)
        g->Hq = d->Hin; g->Wq = d->Win; g->Hv = d->Hout; g->Wv = d->Wout; g->Ca = d->Cin; g->Cb = d->Cout;
    }
    const long Q = (long)d->N * g->Hq * g->Wq, Pv = (long)d->N * g->Hv * g->Wv;
    if (Q * g->Ca * 2 >= (1L << 31) || Pv * g->Cb * 2 >= (1L << 31)) return false;
    g->stride_y = d->stride_h; g->pad_mode = d->pad_mode; g->dy0 = -d->pad_h; g->dx0 = -d->pad_w;
    g->segs = (g->Wq + 63) / 64;
    const long ksteps = (long)d->N * g->Hq * g->segs;
    if (ksteps >= (1L << 30)) return false;
    g->ksteps = (int)ksteps;
    // one workgroup per CU: ~256 workgroups, at least 8 K-steps each, no empty split
    const long tiles = (long)(g->Ca / 64) * (g->Cb / 64);
    long ks = (256 + tiles - 1) / tiles;
    if (ks > ksteps / 8) ks = ksteps / 8;
    if (ks < 1) ks = 1;
    if (ks > 256) ks = 256;
    g->per = (int)((ksteps + ks - 1) / ks);
    g->ksplit = (int)((ksteps + g->per - 1) / g->per);
    g->u_bytes = (unsigned)(Q * g->Ca * 2); g->v_bytes = (unsigned)(Pv * g->Cb * 2);
    return true;
}

template <int DT>
static void launch_line(const ir2rgb_conv_desc *d, const WgradLineGeom &g, const uint16_t *U, const uint16_t *V, float *D, hipStream_t s) {
    const unsigned grid = (unsigned)((long)g.ksplit * (g.Ca / 64) * (g.Cb / 64));
    static int ring = -1;           // IR2RGB_WGRAD_RING=0: the k x 1 layers on the row-major line kernel (A/B measurements)
    if (ring < 0) { const char *e = getenv("IR2RGB_WGRAD_RING"); ring = e ? atoi(e) : 1; }
    if (d->kh == 3 && d->kw == 3) conv_wgrad_line_kernel<DT, 3, 3, 2><<<grid, 512, 0, s>>>(U, V, D, g);
    else if (d->kh == 7 && ring && d->stride_h == 1) conv_wgrad_col_kernel<DT, 7, 1><<<grid, 512, 0, s>>>(U, V, D, g);
    else if (d->kh == 4 && ring && d->stride_h == 2) conv_wgrad_col_kernel<DT, 4, 2><<<grid, 512, 0, s>>>(U, V, D, g);
    else if (d->kh == 7) conv_wgrad_line_kernel<DT, 7, 1><<<grid, 512, 0, s>>>(U, V, D, g);
    else if (d->kw == 7) conv_wgrad_line_kernel<DT, 1, 7><<<grid, 512, 0, s>>>(U, V, D, g);
    else conv_wgrad_line_kernel<DT, 4, 1><<<grid, 512, 0, s>>>(U, V, D, g);
}

static bool use_wgrad9() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("IR2RGB_WGRAD9"); v = e ? atoi(e) : 1; }
    return v != 0;
}

extern "C" long ir2rgb_conv2d_wgrad_workspace_elems(const ir2rgb_conv_desc *d) {
    WgradGeom g;
    int rc = plan(d, &g);
    if (rc) return rc;
    Wgrad9Geom g9;
    if (use_wgrad9() && plan9(d, &g9)) return g9.ksplit > 1 ? (long)g9.ksplit * 9 * g9.Ca * g9.Cb : 4;
    WgradLineGeom gl;
    if (plan_line(d, &gl)) return (long)gl.ksplit * d->kh * d->kw * gl.Ca * gl.Cb;
    return (long)g.ksplit * d->kh * d->kw * g.Ca * g.Cb;
}

static int wgrad_impl(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw, float *workspace, void *stream,
                      int acc);

extern "C" int ir2rgb_conv2d_wgrad(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw,
                                   float *workspace, void *stream) {
    return wgrad_impl(d, x, gy, dw, workspace, stream, 0);
}

extern "C" int ir2rgb_conv2d_wgrad_acc(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw,
                                       float *workspace, void *stream) {
    return wgrad_impl(d, x, gy, dw, workspace, stream, 1);
}

extern "C" long ir2rgb_conv2d_wgrad_acc_workspace_elems(const ir2rgb_conv_desc *d) {
    WgradGeom g;
    int rc = plan(d, &g);
    if (rc) return rc;
    WgradLineGeom gl;
    if (plan_line(d, &gl)) return (long)gl.ksplit * d->kh * d->kw * gl.Ca * gl.Cb;
    return (long)g.ksplit * d->kh * d->kw * g.Ca * g.Cb;       // else always the one-tap kernel's slabs (see wgrad_impl)
}

static int wgrad_impl(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw, float *workspace, void *stream,
                      int acc) {
    WgradGeom g;
    int rc = plan(d, &g);
    if (rc) return rc;
    if ((((uintptr_t)x | (uintptr_t)gy) & 15) || !dw || !workspace) return IR2RGB_EALIGN;
    hipStream_t s = as_stream(stream);
    const int ntaps = d->kh * d->kw;
    const long elems = (long)ntaps * g.Ca * g.Cb;
    Wgrad9Geom g9;
    // accumulate mode (a parameter used several times per pass: the discriminators) sums in the finish pass of the
    // one-tap kernel; the nine-tap kernel's direct-write form has no such pass, so those layers take the one-tap kernel
    if (!acc && use_wgrad9() && plan9(d, &g9)) {
        if (((uintptr_t)dw | (uintptr_t)workspace) & 15) return IR2RGB_EALIGN;
        float *dst = g9.ksplit > 1 ? workspace : dw;
        const unsigned grid9 = (unsigned)((long)g9.ksplit * (g9.Ca / 64) * (g9.Cb / 64));
        if (d->dtype == IR2RGB_BF16) conv_wgrad3x3_kernel<IR2RGB_BF16><<<grid9, 512, 0, s>>>((const uint16_t *)gy, (const uint16_t *)x, dst, g9);
        else conv_wgrad3x3_kernel<IR2RGB_F16><<<grid9, 512, 0, s>>>((const uint16_t *)gy, (const uint16_t *)x, dst, g9);
        if (g9.ksplit > 1)
            wgrad_sum_kernel<<<stream_grid(elems / 4, 256), 256, 0, s>>>((const float4 *)workspace, (float4 *)dw, elems / 4, g9.ksplit);
        return ir2rgb_launch_status();
    }
    const uint16_t *U = (const uint16_t *)(d->transposed ? x : gy), *V = (const uint16_t *)(d->transposed ? gy : x);
    WgradLineGeom gl;
    if (plan_line(d, &gl)) {        // k x 1 / 1 x k layers: all taps of a 64 x 64 tile per workgroup
        if (d->dtype == IR2RGB_BF16) launch_line<IR2RGB_BF16>(d, gl, U, V, workspace, s);
        else launch_line<IR2RGB_F16>(d, gl, U, V, workspace, s);
        launch_wgrad_finish(workspace, dw, gl.Ca, gl.Cb, ntaps, gl.ksplit, s, acc);
        return ir2rgb_launch_status();
    }
    const unsigned grid = (unsigned)((long)g.ksplit * ((ntaps + g.tpb - 1) / g.tpb) * ((g.Ca + 127) / 128) * ((g.Cb + 127) / 128));
    static int split = -1;
    if (split < 0) { const char *e = getenv("IR2RGB_WGRAD_SPLIT"); split = e ? atoi(e) : 1; }
    if (split) {
        if (d->dtype == IR2RGB_BF16) conv_wgrad_kernel<IR2RGB_BF16, 1><<<grid, 512, 0, s>>>(U, V, workspace, g);
        else conv_wgrad_kernel<IR2RGB_F16, 1><<<grid, 512, 0, s>>>(U, V, workspace, g);
    } else {
        if (d->dtype == IR2RGB_BF16) conv_wgrad_kernel<IR2RGB_BF16, 0><<<grid, 256, 0, s>>>(U, V, workspace, g);
        else conv_wgrad_kernel<IR2RGB_F16, 0><<<grid, 256, 0, s>>>(U, V, workspace, g);
    }
    launch_wgrad_finish(workspace, dw, g.Ca, g.Cb, ntaps, g.ksplit, s, acc);
    return ir2rgb_launch_status();
}
