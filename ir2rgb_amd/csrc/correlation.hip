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
// A generic one-lane-per-output kernel covers every other parameter set.
//
// Algorithmic bytes (forward) = 4*N*(2*C*H*W + outC*outH*outW); flops = 2*N*outC*outH*outW*k*k*C.
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
