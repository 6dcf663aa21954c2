// heads.hip -- generator output heads (gfx950, HBM-bound, fp32):
//   * head_finish : vertical part of the separable 7x7 head convolutions + bias + tanh / *20 /
//     sigmoid, writing the reference's NCHW fp32 outputs (networks.py:166, :170-171, :200-201);
//   * warp_blend  : grid_sample warp of the previous frame by the predicted flow and the
//     soft-mask blend (networks.py:89-100, :207-209), one lane per pixel.
#include "common.h"

// One block = an 8-row x 16-column pixel tile: the (8 + KH - 1) x 16 pixel records it needs (Cout*KH contiguous
// floats each, reflection resolved at load time) are staged in LDS with coalesced reads -- one lane per pixel reading
// its own 256-byte record column by column touched 16 times more cache lines per instruction -- then every thread
// sums KH taps for three (pixel, cout) outputs and stores 16 consecutive x per (cout, row).
#define HF_TH 8
#define HF_TW 16
__global__ void __launch_bounds__(256)
head_finish_kernel(const float *__restrict__ T, const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                   int Cout, int KH, int CT, int pad, unsigned acts, float mul, int tiles_x, int tiles_y) {
    extern __shared__ float rec[];                       // [(HF_TH + KH - 1)][HF_TW][CU + 1]
    const int CU = Cout * KH, pitch = CU + 1;            // used channels per record (+1: odd pitch, no bank conflicts)
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y;
    const long n = blockIdx.x / (tiles_x * tiles_y);
    const int x0 = bx * HF_TW, y0 = by * HF_TH;
    const int rows = HF_TH + KH - 1;
    const long hw = (long)H * W;
    for (int i = threadIdx.x; i < rows * HF_TW * CU; i += 256) {
        const int k = i % CU, pc = (i / CU) % HF_TW, pr = i / (CU * HF_TW);
        int iy = y0 + pr - pad;
        iy = iy < 0 ? -iy : iy;
        iy = iy >= H ? 2 * H - 2 - iy : iy;
        const int x = x0 + pc;
        float v = 0.f;
        if (x < W && iy >= 0 && iy < H) v = T[((n * H + iy) * (long)W + x) * CT + k];
        rec[(pr * HF_TW + pc) * pitch + k] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < HF_TH * HF_TW * Cout; o += 256) {
        const int tx = o % HF_TW, ty = (o / HF_TW) % HF_TH, co = o / (HF_TW * HF_TH);
        const int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        float acc = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < KH; ++ky) acc += rec[((ty + ky) * HF_TW + tx) * pitch + co * KH + ky];
        const unsigned a = (acts >> (4 * co)) & 15u;
        const float v = a == 1 ? tanhf(acc) : (a == 2 ? 1.f / (1.f + expf(-acc)) : acc * mul);
        out[(n * Cout + co) * hw + (long)y * W + x] = v;
    }
}

// torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border', align_corners=False)
// evaluated on grid = linspace(-1,1,W)[x] + flow_x / ((W-1)/2)  (an align_corners=True lattice:
// the reference's mismatch, SURVEY section 8 a7, is reproduced on purpose).
__device__ __forceinline__ float lin_m1_p1(int i, int n) {
    // torch.linspace(-1, 1, n)[i]: symmetric evaluation from both ends
    if (n == 1) return -1.f;
    float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ void __launch_bounds__(256)
warp_blend_kernel(const float *__restrict__ raw, const float *__restrict__ prev, const float *__restrict__ flow,
                  const float *__restrict__ wgt, float *__restrict__ out, float *__restrict__ warp_out, int Cp, int H,
                  int W, long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        float gx = lin_m1_p1(x, W) + flow[(n * 2 + 0) * hw + p] / ((W - 1.0f) / 2.0f);
        float gy = lin_m1_p1(y, H) + flow[(n * 2 + 1) * hw + p] / ((H - 1.0f) / 2.0f);
        float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
        float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
        float fx = floorf(ix), fy = floorf(iy);
        float tx = ix - fx, ty = iy - fy;
        int x0 = (int)fx, y0 = (int)fy;
        int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);  // weight of the clamped neighbour is 0
        float wnw = (1.f - tx) * (1.f - ty), wne = tx * (1.f - ty), wsw = (1.f - tx) * ty, wse = tx * ty;
        float m = wgt[n * hw + p];
        for (int c = 0; c < 3; ++c) {
            const float *pl = prev + (n * Cp + (Cp - 3 + c)) * hw;
            float v = pl[(long)y0 * W + x0] * wnw + pl[(long)y0 * W + x1] * wne + pl[(long)y1 * W + x0] * wsw +
                      pl[(long)y1 * W + x1] * wse;
            if (warp_out) warp_out[(n * 3 + c) * hw + p] = v;
            float r = raw[(n * 3 + c) * hw + p];
            out[(n * 3 + c) * hw + p] = r * m + v * (1.f - m);
        }
    }
}

// ----------------------------------------------------------------------------------------
// backward of head_finish: gout [N,Cout,H,W] fp32, out = saved forward output ->
//   dpre = gout * f'(pre)  (tanh: 1-out^2, sigmoid: out(1-out), linear: mul)
//   dT[n][y'][x][co*KH+ky] = sum_{y : refl(y+ky-pad) == y'} dpre[n][co][y][x]   (half, CT channels, rest 0)
//   dbias_partial[block][co] = sum over the block's pixels of dpre  (reduced on the host side by a tiny sum)
// One lane per (n, y', x): gathers the <= 3 source rows per ky (identity + two mirrors).
// ----------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
head_finish_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ out, uint16_t *__restrict__ dT,
                       float *__restrict__ partial, int H, int W, int Cout, int KH, int CT, int pad, unsigned acts,
                       float mul, long total) {
    const long hw = (long)H * W;
    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int yp = (int)(p / W), x = (int)(p - (long)yp * W);
        // the CT halfs of this pixel leave as 16-byte stores (CT % 8 == 0): channels are produced in order
        uint4 *dst = reinterpret_cast<uint4 *>(dT + g * CT);
        uint32_t pk[4] = {0, 0, 0, 0};
        int c = 0;
        auto put = [&](float v) {
            uint16_t h;
            if (DT == IR2RGB_BF16) { __bf16 b = (__bf16)v; h = __builtin_bit_cast(uint16_t, b); }
            else { _Float16 b = (_Float16)v; h = __builtin_bit_cast(uint16_t, b); }
            pk[(c & 7) >> 1] |= (uint32_t)h << (16 * (c & 1));
            if ((++c & 7) == 0) {
                dst[(c >> 3) - 1] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                pk[0] = pk[1] = pk[2] = pk[3] = 0;
            }
        };
        for (int co = 0; co < Cout; ++co) {
            const unsigned a = (acts >> (4 * co)) & 15u;
            const float *go = gout + (n * Cout + co) * hw, *oo = out + (n * Cout + co) * hw;
            auto dpre = [&](int y) -> float {
                float o = oo[(long)y * W + x], gg = go[(long)y * W + x];
                return a == 1 ? gg * (1.f - o * o) : (a == 2 ? gg * o * (1.f - o) : gg * mul);
            };
            bsum[co] += dpre(yp);
            for (int ky = 0; ky < KH; ++ky) {
                // rows y with refl(y + ky - pad) == yp:  t = y + ky - pad in {yp, -yp, 2H-2-yp}
                float acc = 0.f;
                int t = yp;
                int y = t - ky + pad;
                if (y >= 0 && y < H) acc += dpre(y);
                if (yp >= 1) { t = -yp; y = t - ky + pad; if (y >= 0 && y < H && t >= -pad) acc += dpre(y); }
                if (yp <= H - 2) { t = 2 * H - 2 - yp; y = t - ky + pad; if (y >= 0 && y < H && t <= H - 1 + pad) acc += dpre(y); }
                put(acc);
            }
        }
        while (c < CT) put(0.f);
    }
    // block reduction of the bias sums -> one row of the partial buffer per block (summed in a fixed order by
    // head_bias_sum_kernel: float atomics made the bias gradients differ in the last bit from run to run)
    __shared__ float red[8][256];
    for (int co = 0; co < Cout; ++co) red[co][threadIdx.x] = bsum[co];
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        if (threadIdx.x < Cout)
            for (int i = 0; i < 256; ++i) s += red[threadIdx.x][i];
        partial[(long)blockIdx.x * 8 + threadIdx.x] = s;
    }
}

// dbias[co] = sum over the rows of partial, in a fixed order: thread t of channel co sums rows t, t + 32, ... and the 32
// threads' sums are added in lane order
__global__ void __launch_bounds__(256)
head_bias_sum_kernel(const float *__restrict__ partial, float *__restrict__ dbias, int rows, int Cout) {
    __shared__ float red[8][32];
    const int co = threadIdx.x >> 5, t = threadIdx.x & 31;
    float s = 0.f;
    for (int r = t; r < rows; r += 32) s += partial[(long)r * 8 + co];
    red[co][t] = s;
    __syncthreads();
    if (t == 0 && co < Cout) {
        float a = 0.f;
        for (int i = 0; i < 32; ++i) a += red[co][i];
        dbias[co] = a;
    }
}

// ----------------------------------------------------------------------------------------
// backward of warp_blend w.r.t. raw, weight and flow (prev is a detached input on the training path)
//   out = raw*w + warp*(1-w);  warp = bilinear(prev, ix(flow_x), iy(flow_y)) with border clamping
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
warp_blend_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ raw, const float *__restrict__ prev,
                      const float *__restrict__ flow, const float *__restrict__ wgt, float *__restrict__ graw,
                      float *__restrict__ gflow, float *__restrict__ gw, int Cp, int H, int W, long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        float gx = lin_m1_p1(x, W) + flow[(n * 2 + 0) * hw + p] / ((W - 1.0f) / 2.0f);
        float gy = lin_m1_p1(y, H) + flow[(n * 2 + 1) * hw + p] / ((H - 1.0f) / 2.0f);
        float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
        float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        // d(ix)/d(flow_x) = W/(W-1) inside the image, 0 where the border clamp is active (torch clip_coordinates_set_grad)
        float dix = (ix > 0.f && ix < (float)(W - 1)) ? (float)W / (W - 1.0f) : 0.f;
        float diy = (iy > 0.f && iy < (float)(H - 1)) ? (float)H / (H - 1.0f) : 0.f;
        ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
        float fx = floorf(ix), fy = floorf(iy);
        float tx = ix - fx, ty = iy - fy;
        int x0 = (int)fx, y0 = (int)fy;
        bool x1ok = x0 + 1 <= W - 1, y1ok = y0 + 1 <= H - 1;
        int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
        float m = wgt[n * hw + p];
        float gwsum = 0.f, gfx = 0.f, gfy = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float *pl = prev + (n * Cp + (Cp - 3 + c)) * hw;
            float p00 = pl[(long)y0 * W + x0], p01 = x1ok ? pl[(long)y0 * W + x1] : 0.f;
            float p10 = y1ok ? pl[(long)y1 * W + x0] : 0.f, p11 = (x1ok && y1ok) ? pl[(long)y1 * W + x1] : 0.f;
            float warp = p00 * (1.f - tx) * (1.f - ty) + p01 * tx * (1.f - ty) + p10 * (1.f - tx) * ty + p11 * tx * ty;
            float go = gout[(n * 3 + c) * hw + p];
            float r = raw[(n * 3 + c) * hw + p];
            graw[(n * 3 + c) * hw + p] = go * m;
            gwsum += go * (r - warp);
            float gwarp = go * (1.f - m);
            gfx += gwarp * ((p01 - p00) * (1.f - ty) + (p11 - p10) * ty);
            gfy += gwarp * ((p10 - p00) * (1.f - tx) + (p11 - p01) * tx);
        }
        gw[n * hw + p] = gwsum;
        gflow[(n * 2 + 0) * hw + p] = gfx * dix;  // d(ix)/d(flow_x) = (W/2) * 2/(W-1)
        gflow[(n * 2 + 1) * hw + p] = gfy * diy;
    }
}

extern "C" int ir2rgb_head_finish(const float *T, const float *bias, float *out, int N, int H, int W, int Cout, int KH,
                                  int CT, int pad_h, unsigned acts, float mul, void *stream) {
    if (N < 0 || H < 1 || W < 1 || Cout < 1 || Cout > 8 || KH < 1 || CT < Cout * KH || pad_h < 0 || pad_h >= H)
        return IR2RGB_EINVAL;
    if ((long)N * H * W == 0) return IR2RGB_OK;
    if (KH > 16) return IR2RGB_ENOSUP;
    const int tiles_x = (W + HF_TW - 1) / HF_TW, tiles_y = (H + HF_TH - 1) / HF_TH;
    const long blocks = (long)N * tiles_x * tiles_y;
    if (blocks > 0x7fffffffL) return IR2RGB_EINVAL;
    const size_t lds = (size_t)(HF_TH + KH - 1) * HF_TW * (Cout * KH + 1) * sizeof(float);   // <= 23 x 16 x 129 x 4 = 190 KB in theory
    if (lds > 64 * 1024) return IR2RGB_ENOSUP;                                              // 7x7 heads, Cout <= 8: 51 KB
    head_finish_kernel<<<(unsigned)blocks, 256, lds, as_stream(stream)>>>(T, bias, out, H, W, Cout, KH, CT, pad_h, acts, mul,
                                                                          tiles_x, tiles_y);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_warp_blend_fwd(const float *raw, const float *prev, const float *flow, const float *w, float *out,
                                     float *warp_out, int N, int Cp, int H, int W, void *stream) {
    if (N < 0 || Cp < 3 || H < 1 || W < 1) return IR2RGB_EINVAL;
    long total = (long)N * H * W;
    if (total == 0) return IR2RGB_OK;
    warp_blend_kernel<<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>(raw, prev, flow, w, out, warp_out, Cp, H, W,
                                                                              total);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_head_finish_bwd_rows(int N, int H, int W) {
    if (N < 0 || H < 1 || W < 1) return IR2RGB_EINVAL;
    return stream_grid((long)N * H * W, 256);
}

extern "C" int ir2rgb_head_finish_bwd(const float *gout, const float *out, void *dT, float *dbias, float *partial, int N, int H,
                                      int W, int Cout, int KH, int CT, int pad_h, unsigned acts, float mul, int dtype,
                                      void *stream) {
    if (N < 0 || H < 2 || W < 1 || Cout < 1 || Cout > 8 || KH < 1 || CT < Cout * KH || (CT & 7) || pad_h < 0 || pad_h >= H || !partial)
        return IR2RGB_EINVAL;
    if (((uintptr_t)dT & 15) != 0) return IR2RGB_EALIGN;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    long total = (long)N * H * W;
    hipStream_t s = as_stream(stream);
    if (total == 0) {
        hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * Cout, s);
        return e == hipSuccess ? IR2RGB_OK : (int)e;
    }
    int grid = stream_grid(total, 256);
    if (dtype == IR2RGB_BF16)
        head_finish_bwd_kernel<IR2RGB_BF16><<<grid, 256, 0, s>>>(gout, out, (uint16_t *)dT, partial, H, W, Cout, KH, CT, pad_h, acts, mul, total);
    else
        head_finish_bwd_kernel<IR2RGB_F16><<<grid, 256, 0, s>>>(gout, out, (uint16_t *)dT, partial, H, W, Cout, KH, CT, pad_h, acts, mul, total);
    head_bias_sum_kernel<<<1, 256, 0, s>>>(partial, dbias, grid, Cout);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_warp_blend_bwd(const float *gout, const float *raw, const float *prev, const float *flow,
                                     const float *w, float *graw, float *gflow, float *gw, int N, int Cp, int H, int W,
                                     void *stream) {
    if (N < 0 || Cp < 3 || H < 2 || W < 2) return IR2RGB_EINVAL;
    long total = (long)N * H * W;
    if (total == 0) return IR2RGB_OK;
    warp_blend_bwd_kernel<<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>(gout, raw, prev, flow, w, graw, gflow, gw,
                                                                                  Cp, H, W, total);
    return ir2rgb_launch_status();
}

// ----------------------------------------------------------------------------------------
// FlowNet2 flow up-sampler: ConvTranspose2d(2, 2, kernel 4, stride 2, padding 1) on a 2-channel fp32
// flow (reference FlowNetC.py:47-50, FlowNetS.py:42-45 ...), written as NHWC half into channels
// [c_off, c_off+2) of a concatenation buffer.  8 MACs per output: one lane per output pixel.
// ----------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
flow_up_kernel(const float *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
               uint16_t *__restrict__ out, int h, int w, int ld, int c_off, long total) {
    const int H = 2 * h, W = 2 * w;
    const long hw = (long)h * w, HW = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / HW, p = g - n * HW;
        int oy = (int)(p / W), ox = (int)(p - (long)oy * W);
        float acc[2] = {bias ? bias[0] : 0.f, bias ? bias[1] : 0.f};
        for (int ky = (oy + 1) & 1; ky < 4; ky += 2) {
            int iy = (oy + 1 - ky) / 2;  // oy = 2*iy - 1 + ky
            if (oy + 1 - ky < 0 || iy >= h) continue;
            for (int kx = (ox + 1) & 1; kx < 4; kx += 2) {
                int ix = (ox + 1 - kx) / 2;
                if (ox + 1 - kx < 0 || ix >= w) continue;
                for (int ci = 0; ci < 2; ++ci) {
                    float v = in[(n * 2 + ci) * hw + (long)iy * w + ix];
                    acc[0] += v * wgt[((ci * 2 + 0) * 4 + ky) * 4 + kx];
                    acc[1] += v * wgt[((ci * 2 + 1) * 4 + ky) * 4 + kx];
                }
            }
        }
        uint16_t h0, h1;
        if (DT == IR2RGB_BF16) { __bf16 a = (__bf16)acc[0], b = (__bf16)acc[1]; h0 = __builtin_bit_cast(uint16_t, a); h1 = __builtin_bit_cast(uint16_t, b); }
        else { _Float16 a = (_Float16)acc[0], b = (_Float16)acc[1]; h0 = __builtin_bit_cast(uint16_t, a); h1 = __builtin_bit_cast(uint16_t, b); }
        uint16_t *dst = out + g * ld + c_off;
        dst[0] = h0; dst[1] = h1;
    }
}

extern "C" int ir2rgb_flow_upsample_slice(const float *in, const float *weight, const float *bias, void *out, int N,
                                          int h, int w, int ld, int c_off, int dtype, void *stream) {
    if (N < 0 || h < 1 || w < 1 || ld < 2 || c_off < 0 || c_off + 2 > ld) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    long total = (long)N * 4 * h * w;
    if (total == 0) return IR2RGB_OK;
    int grid = stream_grid(total, 256);
    if (dtype == IR2RGB_BF16) flow_up_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>(in, weight, bias, (uint16_t *)out, h, w, ld, c_off, total);
    else flow_up_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>(in, weight, bias, (uint16_t *)out, h, w, ld, c_off, total);
    return ir2rgb_launch_status();
}
