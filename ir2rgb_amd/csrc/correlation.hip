// correlation.hip -- FlowNet2 cost volume (fwd/bwd) for gfx950.
// Semantics follow the reference's correlation_cuda_kernel.cu:46-334 (see include/ir2rgb_hip.h).
//
// Reference structure (for contrast): two extra passes write zero-padded channels-last
// copies of both inputs to HBM, then one 32-thread block per OUTPUT PIXEL walks the 441
// displacements serially, each ending in a warp-shuffle tree whose lane 0 stores 4 bytes.
//
// This design (forward fast path, the FlowNetC configuration k=1, stride1=1, pad=max_disp):
//   * reads the NCHW inputs directly; zero padding is a predicate on 16-byte loads, so the
//     algorithmic traffic is 2 inputs + 1 output and no scratch tensor exists;
//   * one WORKGROUP owns (n, y, tj, 128-wide x chunk); its 4 waves split the 21 x-displacements in
//     two halves and the channels in two halves.  In a wave 16 lanes tile x in runs of 8 pixels and
//     the 4 lane-quarters interleave the channels; a lane keeps a (<=12) x 8 register tile of partial
//     dot products: per channel it loads 8 floats of f1 and <=30 floats of f2 as float4s and issues
//     up to 96 FMAs (2.5 FMA per loaded dword) at ~160 VGPRs (3 waves per SIMD);
//   * the lane quarters are combined with two wave64 xor-shuffles (lanes^16, ^32), the channel
//     halves through LDS; every lane quarter then stores its share of the rows as 32-byte runs
//     (512 B contiguous per row per wave).
// corr_fwd_lds (below) is that design with its operands staged through LDS by LDS-DMA and packed FMAs.
// A generic one-lane-per-output kernel covers every other parameter set.
// Backward (corr_bwd_kernel): one lane per input element, API completeness only -- FlowNet2 runs under no_grad on
// this path (models/flownet.py:21), so it is NOT a performance kernel (2.8 ms at the config-3 size).
//
// Algorithmic bytes (forward) = 4*N*(2*C*H*W + outC*outH*outW); flops = 2*N*outC*outH*outW*k*k*C.
#include <type_traits>
#include <utility>

#include "common.h"

static void out_shape(int H, int W, int pad, int ksize, int md, int s1, int s2, int *oc, int *oh, int *ow) {
    int krad = (ksize - 1) / 2, border = krad + md;
    int pH = H + 2 * pad, pW = W + 2 * pad, drad = md / s2;
    *oc = (2 * drad + 1) * (2 * drad + 1);
    *oh = (int)ceilf((float)(pH - 2 * border) / (float)s1);
    *ow = (int)ceilf((float)(pW - 2 * border) / (float)s1);
}

extern "C" int ir2rgb_correlation_out_shape(int C, int H, int W, int pad_size, int kernel_size,
                                            int max_displacement, int stride1, int stride2, int *outC, int *outH,
                                            int *outW) {
    (void)C;
    if (stride1 < 1 || stride2 < 1 || kernel_size < 1 || !(kernel_size & 1) || pad_size < 0 || max_displacement < 0)
        return IR2RGB_EINVAL;
    out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, outC, outH, outW);
    return IR2RGB_OK;
}

// ----------------------------------------------------------------------------------------
// fast forward path
//   workgroup (256 threads) = one unit (n, y, tj, 128-wide x chunk); its 4 waves split the unit into
//   2 displacement halves (ti 0..T0-1 and T0..D-1; T0 = 12 keeps both f2 windows float4 aligned) x
//   2 channel halves.  Within a wave: 16 lanes tile x in runs of 8 pixels, the 4 lane-quarters
//   interleave the wave's channels.  A lane keeps a (<= 12) x 8 register tile of partial dot
//   products (<= 96 accumulators, ~160 VGPRs -> 3 waves per SIMD): 4x the waves in flight of a
//   one-wave-per-unit layout, which is what hides the L2/HBM latency of the operand loads.
//   Reduction: two wave64 xor-shuffles combine the lane quarters, the channel halves meet in LDS.
// ----------------------------------------------------------------------------------------
template <int DR, int S2, int T0>
__global__ void __launch_bounds__(256)
corr_fwd_tile(const float *__restrict__ f1, const float *__restrict__ f2, float *__restrict__ out, int C, int H,
              int W, int xchunks, float inv_nelems) {
    constexpr int D = 2 * DR + 1;
    constexpr int TMAX = T0 > D - T0 ? T0 : D - T0;  // displacements per wave (first half is the larger)
    constexpr int HALO = S2 * DR;
    constexpr int NB = 8 + S2 * (TMAX - 1);           // f2 floats needed per lane per channel
    constexpr int NB4 = (NB + 3) / 4;
    static_assert(HALO % 4 == 0 && (S2 * T0) % 4 == 0, "f2 windows must keep float4 alignment");
    __shared__ float red[2][TMAX * 8][64];            // [ti half][acc index][lane]: partial sums of channel half 1

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int th = wave & 1, ch = wave >> 1;          // displacement half, channel half
    long unit = blockIdx.x;
    const int xc = (int)(unit % xchunks); unit /= xchunks;
    const int tj = (int)(unit % D) - DR;  unit /= D;
    const int y = (int)(unit % H);
    const int n = (int)(unit / H);

    const int t0 = th ? T0 : 0, nt = th ? D - T0 : T0;   // this wave's ti range [t0, t0+nt)
    const int xo = lane & 15, cs = lane >> 4;
    const int x0 = xc * 128 + xo * 8;
    const int y2 = y + tj * S2;
    const bool row_ok = (y2 >= 0) && (y2 < H);       // workgroup-uniform
    const bool lane_ok = x0 < W;
    const long hw = (long)H * W;
    const int xb = x0 - HALO + S2 * t0;              // first f2 column of this wave's window (multiple of 4)

    float acc[TMAX][8];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[t][m] = 0.f;

    if (row_ok && lane_ok) {
        const float *p1 = f1 + ((long)n * C) * hw + (long)y * W + x0;
        const float *p2 = f2 + ((long)n * C) * hw + (long)y2 * W + xb;
        bool inb[NB4];
#pragma unroll
        for (int j = 0; j < NB4; ++j) {
            int xs = xb + 4 * j;
            inb[j] = (xs >= 0) && (xs + 3 < W);
        }
        // channels of this wave: c = 2*(4*i + cs) + ch  -> halves interleave, quarters interleave
        for (int c = 2 * cs + ch; c < C; c += 8) {
            const float4 *a4 = reinterpret_cast<const float4 *>(p1 + (long)c * hw);
            const float4 *b4 = reinterpret_cast<const float4 *>(p2 + (long)c * hw);
            float a[8], b[NB4 * 4];
            float4 u0 = a4[0], u1 = a4[1];
            a[0] = u0.x; a[1] = u0.y; a[2] = u0.z; a[3] = u0.w;
            a[4] = u1.x; a[5] = u1.y; a[6] = u1.z; a[7] = u1.w;
#pragma unroll
            for (int j = 0; j < NB4; ++j) {
                float4 v = inb[j] ? b4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (t < nt) acc[t][m] = fmaf(a[m], b[m + S2 * t], acc[t][m]);
        }
    }

    // lane quarters -> every lane holds the wave's sum
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            float v = acc[t][m];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][m] = v;
        }
    // channel halves meet in LDS: waves with ch == 1 publish, waves with ch == 0 add and store
    if (ch == 1) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int m = 0; m < 8; ++m) red[th][t * 8 + m][lane] = acc[t][m];
    }
    __syncthreads();
    if (ch == 0 && lane_ok) {
        float *o = out + (((long)n * (D * D) + (long)(tj + DR) * D + t0) * H + y) * (long)W + x0;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < nt && (t & 3) == cs) {   // lane quarter cs stores rows t = cs, cs+4, ...
                float r[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) r[m] = (acc[t][m] + red[th][t * 8 + m][lane]) * inv_nelems;
                float4 *q = reinterpret_cast<float4 *>(o + (long)t * hw);
                q[0] = make_float4(r[0], r[1], r[2], r[3]);
                q[1] = make_float4(r[4], r[5], r[6], r[7]);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// LDS-staged forward path (the default for the FlowNetC configuration; corr_fwd_tile above stays as the
// fallback for channel counts that are not multiples of 8 and as the A/B reference, IR2RGB_CORR_LDS=0).
//
// corr_fwd_tile feeds its FMAs straight from L2: per channel a lane issues 10 global_load_dwordx4 for 96 FMAs,
// and the texture path hands a CU 64 B per clock -- 16 clocks per 1 KB wave-load, ~39 us of load issue chip-wide
// for the [1,256,64,128] volume: that, not the 12 us of FMAs, set its 74 us.  Here the same register tile (a lane
// keeps 8 pixels x <= 12 displacements, the 4 waves of a workgroup = 2 displacement halves x 2 channel halves,
// the 4 lane quarters interleave channels) is fed from LDS, which serves a 1 KB ds_read_b128 in 4 clocks:
//   * operands arrive by LDS-DMA (buffer_load ... lds: no VGPR round trip, per-lane 32-bit source offsets computed
//     once, the channel-group offset rides in an SGPR; out-of-row columns are out-of-range buffer offsets = zeros,
//     which IS the zero padding): one stage = 8 channels of the f1 row chunk (8 x 128 floats) and of the f2 row
//     window (8 x 176 floats, halo 20 left / 28 right), 11 one-KB DMA instructions, three stages in flight;
//   * rows of odd lane quarters are stored one 16-byte slot to the right: the 16 lanes a ds_read_b128 serves per
//     clock ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... : two quarters at a time, MI355X_MICROARCH.md LDS) then
//     touch 16 distinct 16-byte bank windows (even ones for one quarter, odd ones for the other);
//   * the FMAs are packed: a pixel pair times a pixel pair of the shifted window (stride2 = 2 shifts by whole
//     pairs), v_pk_fma_f32 -- 48 instructions per channel instead of 96.
// One s_barrier + one counted vmcnt per 8-channel stage.  Reduction and stores as in corr_fwd_tile.
// ----------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) float corr_f2;
typedef __attribute__((address_space(3))) void *corr_lptr_t;
#define CORR_OOB 0x80000000u

template <int DR, int S2, int T0>
__global__ void __launch_bounds__(256)
corr_fwd_lds(const float *__restrict__ f1, const float *__restrict__ f2, float *__restrict__ out, int C, int H, int W,
             int xchunks, float inv_nelems, unsigned in_bytes) {
    static_assert(S2 == 2, "packed form: a displacement step shifts the window by one pixel pair");
    constexpr int D = 2 * DR + 1;
    constexpr int TMAX = T0 > D - T0 ? T0 : D - T0;
    constexpr int HALO = S2 * DR;                       // 20
    constexpr int NB = 8 + S2 * (TMAX - 1), NB4 = (NB + 3) / 4;   // 30 floats -> 8 chunks per lane and channel
    constexpr int F1P = 40, F2P = 48;                   // row pitches in 16-byte slots (32 / 44 data chunks + shift + pad)
    constexpr int F2C = 44;                             // data chunks of an f2 row: x in [x_chunk - 20, x_chunk + 156)
    constexpr int F1S = 8 * F1P, SLOTS = F1S + 8 * F2P; // 320 + 384 slots per stage: f1 ends on an instruction boundary
    constexpr int NDMA = SLOTS / 64, NF1 = F1S / 64;    // 11 DMA instructions per stage, the first 5 read f1
    constexpr int PERW = (NDMA + 3) / 4;                // 3 per wave (wave 3: its third is a dummy)
    constexpr int STAGE = NDMA * 1024, NST = 4;         // four stages: three in flight while one is consumed
    static_assert(HALO % 4 == 0 && (S2 * T0) % 4 == 0, "f2 windows must keep float4 alignment");
    static_assert(F1S % 64 == 0 && SLOTS % 64 == 0, "regions must end on DMA instruction boundaries");
    constexpr int RED = 2 * TMAX * 8 * 64 * 4;          // reduction scratch (re-uses the ring after the last stage)
    constexpr int RING = NST * STAGE > RED ? NST * STAGE : RED;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + 1024];
    unsigned char *const dummy = smem + RING;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int th = wave & 1, ch = wave >> 1;
    // XCD-aware unit ids: workgroup b runs on XCD b % 8, and the eight L2s (4 MB each) do not share.  Units are
    // numbered (n, y, tj, x chunk) with y slowest; giving XCD k the k-th contiguous eighth of them makes the units that
    // read the same f1 row (all 21 tj of a y) and the same f2 row ((y, tj) and (y + 2, tj - 1)) neighbours in ONE L2:
    // with round-robin ids every XCD streamed all 16.8 MB of both tensors through its own 4 MB.
    long unit = blockIdx.x;
    if ((gridDim.x & 7) == 0) unit = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int xc = (int)(unit % xchunks); unit /= xchunks;
    const int tj = (int)(unit % D) - DR;  unit /= D;
    const int y = (int)(unit % H);
    const int n = (int)(unit / H);

    const int t0 = th ? T0 : 0, nt = th ? D - T0 : T0;
    const int xo = lane & 15, cs = lane >> 4;
    const int x0 = xc * 128 + xo * 8;
    const int y2 = y + tj * S2;
    const bool row_ok = (y2 >= 0) && (y2 < H);       // workgroup-uniform
    const bool lane_ok = x0 < W;
    const long hw = (long)H * W;

    corr_f2 acc[TMAX][4];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t][k] = (corr_f2){0.f, 0.f};

    if (row_ok) {
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(f1), 0, (int)in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(f2), 0, (int)in_bytes, 0x00020000);
        // ---- per-lane DMA source offsets (bytes) of this wave's instructions: slot -> (tensor, row, chunk) ----
        unsigned voff[PERW];
#pragma unroll
        for (int j = 0; j < PERW; ++j) {
            const int id = wave + 4 * j;                 // wave-uniform instruction id
            const int slot = id * 64 + lane;
            unsigned v = CORR_OOB;
            if (id < NF1) {
                const int row = slot / F1P, pos = slot - row * F1P, chunk = pos - ((row >> 1) & 1);
                const int x = xc * 128 + 4 * chunk;
                if (chunk >= 0 && chunk < 32 && x + 3 < W)
                    v = (unsigned)((((long)n * C + row) * hw + (long)y * W + x) * 4);
            } else if (id < NDMA) {
                const int s2 = slot - F1S, row = s2 / F2P, pos = s2 - row * F2P, chunk = pos - ((row >> 1) & 1);
                const int x = xc * 128 - HALO + 4 * chunk;
                if (chunk >= 0 && chunk < F2C && x >= 0 && x + 3 < W)
                    v = (unsigned)((((long)n * C + row) * hw + (long)y2 * W + x) * 4);
            }
            voff[j] = v;
        }
        const int NG = C >> 3;
        auto issue = [&](int g) {
            const bool live = g < NG;                    // past the last stage: same instruction count, harmless target
            unsigned char *dst = smem + (g % NST) * STAGE;
            const unsigned soff = live ? (unsigned)((long)g * 8 * hw * 4) : 0u;
#pragma unroll
            for (int j = 0; j < PERW; ++j) {
                const int id = wave + 4 * j;
                if (id >= NDMA || !live) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (corr_lptr_t)dummy, 16, CORR_OOB, 0, 0, 0);
                else if (id < NF1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (corr_lptr_t)(dst + id * 1024), 16, voff[j], soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (corr_lptr_t)(dst + id * 1024), 16, voff[j], soff, 0, 0);
            }
        };
        // ---- per-lane LDS read offsets: row = this lane's channel within the group of 8, shifted if cs is odd ----
        const int row = 2 * cs + ch, rot = cs & 1;
        const unsigned aofs = (unsigned)((row * F1P + rot + 2 * xo) * 16);
        const unsigned bofs = (unsigned)((F1S + row * F2P + rot + 2 * xo + (S2 * t0) / 4) * 16);
        auto run = [&]<int NT>(std::integral_constant<int, NT>) {
            issue(0);
            issue(1);
            issue(2);
            for (int g = 0; g < NG; ++g) {
                // stages g+1, g+2 (2 * PERW instructions of this wave) may still be in flight; past the end the
                // no-op stages keep the count uniform
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PERW) : "memory");
                __builtin_amdgcn_s_barrier();            // stage g has landed for every wave; stage g-1 is consumed
                issue(g + 3);
                const unsigned char *st = smem + (g % NST) * STAGE;
                corr_f2 a[4], b[NB4 * 2];
                {
                    const float4 u0 = *reinterpret_cast<const float4 *>(st + aofs), u1 = *reinterpret_cast<const float4 *>(st + aofs + 16);
                    a[0] = (corr_f2){u0.x, u0.y}; a[1] = (corr_f2){u0.z, u0.w};
                    a[2] = (corr_f2){u1.x, u1.y}; a[3] = (corr_f2){u1.z, u1.w};
                }
#pragma unroll
                for (int j = 0; j < (8 + S2 * (NT - 1) + 3) / 4; ++j) {
                    const float4 v = *reinterpret_cast<const float4 *>(st + bofs + 16 * j);
                    b[2 * j] = (corr_f2){v.x, v.y};
                    b[2 * j + 1] = (corr_f2){v.z, v.w};
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[t][k] = __builtin_elementwise_fma(a[k], b[k + t], acc[t][k]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing no-op stages
        };
        if (th == 0) run(std::integral_constant<int, T0>{});
        else run(std::integral_constant<int, D - T0>{});
        __builtin_amdgcn_s_barrier();                // the ring is free: it becomes the reduction scratch
    }

    float (*red)[TMAX * 8][64] = reinterpret_cast<float (*)[TMAX * 8][64]>(smem);
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            corr_f2 v = acc[t][k];
            v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
            v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
            acc[t][k] = v;
        }
    if (ch == 1) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[th][t * 8 + 2 * k][lane] = acc[t][k].x; red[th][t * 8 + 2 * k + 1][lane] = acc[t][k].y; }
    }
    __syncthreads();
    if (ch == 0 && lane_ok) {
        float *o = out + (((long)n * (D * D) + (long)(tj + DR) * D + t0) * H + y) * (long)W + x0;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < nt && (t & 3) == cs) {
                float r[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[2 * k] = (acc[t][k].x + red[th][t * 8 + 2 * k][lane]) * inv_nelems;
                    r[2 * k + 1] = (acc[t][k].y + red[th][t * 8 + 2 * k + 1][lane]) * inv_nelems;
                }
                float4 *q = reinterpret_cast<float4 *>(o + (long)t * hw);
                q[0] = make_float4(r[0], r[1], r[2], r[3]);
                q[1] = make_float4(r[4], r[5], r[6], r[7]);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// generic forward: one lane per output element, any (pad, k, md, s1, s2)
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
corr_fwd_generic(const float *__restrict__ f1, const float *__restrict__ f2, float *__restrict__ out, int C,
                 int H, int W, int pad, int krad, int md, int s1, int s2, int drad, int outH, int outW,
                 long total, float inv_nelems) {
    const int dsz = 2 * drad + 1;
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long r = g;
        int ox = (int)(r % outW); r /= outW;
        int oy = (int)(r % outH); r /= outH;
        int tc = (int)(r % (dsz * dsz));
        int n = (int)(r / (dsz * dsz));
        int tj = tc / dsz - drad, ti = tc % dsz - drad;
        // coordinates in the padded frame, then back to the unpadded one
        int y1 = oy * s1 + md - pad, x1 = ox * s1 + md - pad;
        int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
        float acc = 0.f;
        for (int j = -krad; j <= krad; ++j)
            for (int i = -krad; i <= krad; ++i) {
                int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
                if (ya < 0 || ya >= H || xa < 0 || xa >= W || yb < 0 || yb >= H || xb < 0 || xb >= W) continue;
                const float *pa = f1 + (long)n * C * hw + (long)ya * W + xa;
                const float *pb = f2 + (long)n * C * hw + (long)yb * W + xb;
                for (int c = 0; c < C; ++c) acc = fmaf(pa[(long)c * hw], pb[(long)c * hw], acc);
            }
        out[g] = acc * inv_nelems;
    }
}

// ----------------------------------------------------------------------------------------
// backward (stride1 == 1): one lane per (n, c, y, x), both gradients.
//   gin1[n,c,y,x] = 1/nelems * sum_tc sum_{(j,i) in window1}      gout[n,tc,j,i] * f2pad[n,c,y+j2,x+i2]
//   gin2[n,c,y,x] = 1/nelems * sum_tc sum_{(j,i) in window2(tc)}  gout[n,tc,j,i] * f1pad[n,c,y-j2,x-i2]
// with the windows of correlation_cuda_kernel.cu:166-186 / :278-299.  Lanes are consecutive in
// x so every plane access is coalesced; the op is not on the training path (FlowNet2 runs under
// no_grad, reference models/flownet.py:21) and is provided for API completeness.
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
corr_bwd_kernel(const float *__restrict__ f1, const float *__restrict__ f2, const float *__restrict__ gout,
                float *__restrict__ gin1, float *__restrict__ gin2, int C, int H, int W, int pad, int krad,
                int md, int s2, int drad, int outH, int outW, long total, float inv_nelems) {
    const int dsz = 2 * drad + 1, outC = dsz * dsz;
    const long hw = (long)H * W, ohw = (long)outH * outW;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long r = g;
        int bx = (int)(r % W); r /= W;
        int by = (int)(r % H); r /= H;
        int c = (int)(r % C);
        int n = (int)(r / C);
        int y = by + pad, x = bx + pad;  // padded-frame coordinates (stride1 == 1)
        const float *go = gout + (long)n * outC * ohw;
        const float *p1 = f1 + ((long)n * C + c) * hw;
        const float *p2 = f2 + ((long)n * C + c) * hw;
        float s1acc = 0.f, s2acc = 0.f;
        // window of input1 does not depend on tc
        int xmin1 = x - krad - md, ymin1 = y - krad - md, xmax1 = x + krad - md, ymax1 = y + krad - md;
        bool ok1 = !(xmax1 < 0 || ymax1 < 0 || xmin1 >= outW || ymin1 >= outH) && !(xmin1 > xmax1 || ymin1 > ymax1);
        xmin1 = max(0, xmin1); xmax1 = min(outW - 1, xmax1);
        ymin1 = max(0, ymin1); ymax1 = min(outH - 1, ymax1);
        for (int tc = 0; tc < outC; ++tc) {
            int i2 = (tc % dsz - drad) * s2, j2 = (tc / dsz - drad) * s2;
            const float *got = go + (long)tc * ohw;
            if (ok1) {
                int yy = y + j2 - pad, xx = x + i2 - pad;  // unpadded f2 coordinates
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    float v2 = p2[(long)yy * W + xx];
                    float s = 0.f;
                    for (int j = ymin1; j <= ymax1; ++j)
                        for (int i = xmin1; i <= xmax1; ++i) s += got[(long)j * outW + i];
                    s1acc = fmaf(s, v2, s1acc);
                }
            }
            int xmin = x - krad - md - i2, ymin = y - krad - md - j2;
            int xmax = x + krad - md - i2, ymax = y + krad - md - j2;
            if (xmax < 0 || ymax < 0 || xmin >= outW || ymin >= outH) continue;
            if (xmin > xmax || ymin > ymax) continue;
            xmin = max(0, xmin); xmax = min(outW - 1, xmax);
            ymin = max(0, ymin); ymax = min(outH - 1, ymax);
            int yy = y - j2 - pad, xx = x - i2 - pad;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            float v1 = p1[(long)yy * W + xx];
            float s = 0.f;
            for (int j = ymin; j <= ymax; ++j)
                for (int i = xmin; i <= xmax; ++i) s += got[(long)j * outW + i];
            s2acc = fmaf(s, v1, s2acc);
        }
        gin1[g] = s1acc * inv_nelems;
        gin2[g] = s2acc * inv_nelems;
    }
}

static int check_params(int N, int C, int H, int W, int pad, int ksize, int md, int s1, int s2) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return IR2RGB_EINVAL;
    if (s1 < 1 || s2 < 1 || ksize < 1 || !(ksize & 1) || pad < 0 || md < 0) return IR2RGB_EINVAL;
    return IR2RGB_OK;
}

extern "C" int ir2rgb_correlation_fwd(const float *in1, const float *in2, float *out, int N, int C, int H, int W,
                                      int pad_size, int kernel_size, int max_displacement, int stride1,
                                      int stride2, void *stream) {
    int rc = check_params(N, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    if (rc) return rc;
    int oc, oh, ow;
    out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (oh <= 0 || ow <= 0) return IR2RGB_EINVAL;
    long total = (long)N * oc * oh * ow;
    if (total == 0) return IR2RGB_OK;
    const int drad = max_displacement / stride2;
    const float inv = 1.0f / (float)(kernel_size * kernel_size * C);
    hipStream_t s = as_stream(stream);
    const bool aligned = (((uintptr_t)in1 | (uintptr_t)in2 | (uintptr_t)out) % 16) == 0;
    const bool fast = kernel_size == 1 && stride1 == 1 && pad_size == max_displacement && stride2 == 2 &&
                      drad == 10 && max_displacement == 20 && (W % 8 == 0) && aligned && C > 0;
    if (fast) {
        int xchunks = cdiv(W, 128);
        long units = (long)N * H * 21 * xchunks;  // (n, y, tj, xchunk): one workgroup each
        static int use_lds = -1;
        if (use_lds < 0) { const char *e = getenv("IR2RGB_CORR_LDS"); use_lds = e ? atoi(e) : 1; }
        const long in_bytes = (long)N * C * H * W * 4;
        if (use_lds && (C % 8) == 0 && in_bytes < (1L << 31))
            corr_fwd_lds<10, 2, 12><<<(unsigned)units, 256, 0, s>>>(in1, in2, out, C, H, W, xchunks, inv, (unsigned)in_bytes);
        else
            corr_fwd_tile<10, 2, 12><<<(unsigned)units, 256, 0, s>>>(in1, in2, out, C, H, W, xchunks, inv);
        return ir2rgb_launch_status();
    }
    corr_fwd_generic<<<stream_grid(total, 256), 256, 0, s>>>(in1, in2, out, C, H, W, pad_size,
                                                             (kernel_size - 1) / 2, max_displacement, stride1,
                                                             stride2, drad, oh, ow, total, inv);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_correlation_bwd(const float *in1, const float *in2, const float *gout, float *gin1,
                                      float *gin2, int N, int C, int H, int W, int pad_size, int kernel_size,
                                      int max_displacement, int stride1, int stride2, void *stream) {
    int rc = check_params(N, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    if (rc) return rc;
    if (stride1 != 1) return IR2RGB_ENOSUP;
    int oc, oh, ow;
    out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (oh <= 0 || ow <= 0) return IR2RGB_EINVAL;
    long total = (long)N * C * H * W;
    if (total == 0) return IR2RGB_OK;
    const float inv = 1.0f / (float)(kernel_size * kernel_size * C);
    corr_bwd_kernel<<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>(
        in1, in2, gout, gin1, gin2, C, H, W, pad_size, (kernel_size - 1) / 2, max_displacement, stride2,
        max_displacement / stride2, oh, ow, total, inv);
    return ir2rgb_launch_status();
}
