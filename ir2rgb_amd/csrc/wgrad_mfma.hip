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

template <int DT>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, float *__restrict__ D,
                  const WgradGeom g) {
    constexpr int STAGE = 2 * 64 * 256;  // U tile + V tile, 64 pixel rows x 256 B each
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nta = (g.Ca + 127) >> 7, ntb = (g.Cb + 127) >> 7;
    // XCD-aware bijective remap (blocks b, b+8, ... share an L2): each XCD gets a contiguous run of
    // (split, tap, tile_a, tile_b) ids, i.e. few distinct U / V panels per private L2.
    int bid;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tb = bid % ntb; bid /= ntb;
    const int ta = bid % nta; bid /= nta;
    const int tap = bid % (g.nty * g.ntx);
    const int split = bid / (g.nty * g.ntx);
    const int ty = tap / g.ntx, tx = tap - ty * g.ntx;
    const int dy = g.dy0 + ty, dx = g.dx0 + tx;
    const int a0 = ta * 128, b0 = tb * 128;
    const int per = (g.ksteps + g.ksplit - 1) / g.ksplit;
    const int kbeg = split * per, kend = min(g.ksteps, kbeg + per);
    if (kbeg >= kend) return;  // workgroup-uniform

    // ---------------- staging roles: 16 lanes per 256-B pixel row, 4 rows per wave instruction ----------------
    const int slot = tid & 15, r0 = tid >> 4;                   // rows r0 + 16*i, i = 0..3
    const int f = (r0 & 3) | (((r0 >> 3) & 1) << 2);            // same for r0 + 16*i
    const int chunk = ((((slot >> 1) ^ f) << 1) | (slot & 1));  // source 16-B chunk held by this LDS slot
    const bool a_ok = a0 + chunk * 8 < g.Ca, b_ok = b0 + chunk * 8 < g.Cb;
    const rsrc_t ru = make_rsrc(U, g.u_bytes), rv = make_rsrc(V, g.v_bytes);
    const unsigned ubase = a_ok ? (unsigned)(a0 + chunk * 8) * 2u : WG_OOB;  // byte offset inside a pixel row
    const unsigned vbase = b_ok ? (unsigned)(b0 + chunk * 8) * 2u : WG_OOB;
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

    issue(kbeg, 0);
    int buf = 0;
    for (int ks = kbeg; ks < kend; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ks + 1 < kend) issue(ks + 1, buf ^ 1);
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
    float *Dt = D + ((long)split * (g.nty * g.ntx) + tap) * g.Ca * g.Cb;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int b = b0 + wc * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + wr * 64 + mi * 16 + grp * 4 + r;
                if (a < g.Ca && b < g.Cb) {
                    Dt[(long)a * g.Cb + b] = acc[mi][ni][r];
                }
            }
        }
}

// out[a][b][tap] = sum_split D[split][tap][a][b]   (torch weight layout [Ca][Cb][kh][kw], fp32)
__global__ void __launch_bounds__(256)
wgrad_finish_kernel(const float *__restrict__ D, float *__restrict__ out, int Ca, int Cb, int ntaps, int nsplit,
                    long total) {
    const long slab = (long)ntaps * Ca * Cb;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % ntaps);
        const long ab = i / ntaps;
        const float *p = D + (long)tap * Ca * Cb + ab;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += p[k * slab];
        out[i] = s;
    }
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
    const long tiles = (long)d->kh * d->kw * ((g->Ca + 127) / 128) * ((g->Cb + 127) / 128);
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

extern "C" long ir2rgb_conv2d_wgrad_workspace_elems(const ir2rgb_conv_desc *d) {
    WgradGeom g;
    int rc = plan(d, &g);
    if (rc) return rc;
    return (long)g.ksplit * d->kh * d->kw * g.Ca * g.Cb;
}

extern "C" int ir2rgb_conv2d_wgrad(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw,
                                   float *workspace, void *stream) {
    WgradGeom g;
    int rc = plan(d, &g);
    if (rc) return rc;
    if ((((uintptr_t)x | (uintptr_t)gy) & 15) || !dw || !workspace) return IR2RGB_EALIGN;
    hipStream_t s = as_stream(stream);
    const int ntaps = d->kh * d->kw;
    const long elems = (long)ntaps * g.Ca * g.Cb;
    const uint16_t *U = (const uint16_t *)(d->transposed ? x : gy), *V = (const uint16_t *)(d->transposed ? gy : x);
    const unsigned grid = (unsigned)((long)g.ksplit * ntaps * ((g.Ca + 127) / 128) * ((g.Cb + 127) / 128));
    if (d->dtype == IR2RGB_BF16) conv_wgrad_kernel<IR2RGB_BF16><<<grid, 256, 0, s>>>(U, V, workspace, g);
    else conv_wgrad_kernel<IR2RGB_F16><<<grid, 256, 0, s>>>(U, V, workspace, g);
    wgrad_finish_kernel<<<stream_grid(elems, 256), 256, 0, s>>>(workspace, dw, g.Ca, g.Cb, ntaps, g.ksplit, elems);
    return ir2rgb_launch_status();
}
