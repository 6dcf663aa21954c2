// pointwise.hip -- HBM-bound companions of the MFMA convolution (gfx950):
//   * bn_finalize / bn_apply : training-mode BatchNorm2d + ReLU/LeakyReLU (+ residual adds),
//     the norm_layer(...) + activation modules of reference models/networks.py:141-171,
//     :253-271, :556-586, :678-699 (nn.BatchNorm2d semantics incl. running statistics);
//   * layout converters between the reference's NCHW fp32 tensors and NHWC half;
//   * xexpand: x-direction im2col for the small-Cin first layers (7x7 on 9/6 channels, 4x4 on
//     6/13 channels) so they run on the same MFMA kernel as a k x 1 convolution over 64 channels.
// All kernels move 16 bytes per lane on the NHWC side.
#include "common.h"

static __device__ __forceinline__ float h2f(uint16_t h, int dt) {
    if (dt == IR2RGB_BF16) return __uint_as_float(((uint32_t)h) << 16);
    _Float16 v = __builtin_bit_cast(_Float16, h);
    return (float)v;
}
static __device__ __forceinline__ uint16_t f2h(float f, int dt) {
    if (dt == IR2RGB_BF16) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// ----------------------------------------------------------------------------------------
// BatchNorm statistics: reduce the per-tile partials written by the conv epilogue.
// Block = 32 channels x 32 row groups (1024 threads): lanes of a wave read 32 consecutive channels
// of two rows (coalesced 128 B each), row groups stride the partial rows, LDS combines them.
// Sums in double: var = E[y^2] - E[y]^2.
// ----------------------------------------------------------------------------------------
// CL = 8 (8 channels x 128 row groups): four times the workgroups for layers with thousands of partial rows and few
// channels (the 64-channel layers of a 1024x2048 frame write 16384 rows: two 32-channel workgroups walked them in
// 64 dependent rounds, 48 us; eight workgroups of 128 row groups need 16).
template <int CL>
__global__ void __launch_bounds__(1024)
bn_finalize_kernel(const float *__restrict__ partial, int rows, int C, double count, const float *__restrict__ gamma,
                   const float *__restrict__ beta, const float *__restrict__ conv_bias, float *__restrict__ running_mean,
                   float *__restrict__ running_var, float momentum, float eps, float *__restrict__ scale,
                   float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ invstd_out,
                   int stat_updates) {
    constexpr int RG = 1024 / CL;
    __shared__ double red[2][RG][CL + 1];
    const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = rg; r < rows; r += RG * 8) {
            // eight row pairs in flight per thread (the plain loop waits for every pair: ~0.2 us x rows/32, 23 us
            // for the 4096 partial rows of a full-resolution layer).  The loads are unconditional on a clamped row
            // and masked afterwards: a predicated load compiles to a branch with its own wait, i.e. serial again.
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = min(r + RG * u, rows - 1);
                a[u] = partial[((long)rr * 2 + 0) * C + c];
                b[u] = partial[((long)rr * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = r + RG * u < rows;
                s1 += ok ? (double)a[u] : 0.0;      // summed in row order, as the plain loop
                s2 += ok ? (double)b[u] : 0.0;
            }
        }
    red[0][rg][cl] = s1;
    red[1][rg][cl] = s2;
    __syncthreads();
    if (rg != 0 || c >= C) return;
    s1 = s2 = 0.0;
    // unrolled in full the 64 doubles are all loaded first: 128 VGPRs = the cap of a 1024-thread block, and the
    // compiler spilled 7 of them (a kernel with scratch pays for it at every dispatch)
#pragma unroll 8
    for (int r = 0; r < RG; ++r) { s1 += red[0][r][cl]; s2 += red[1][r][cl]; }
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (mean_out) mean_out[c] = (float)mean;
    if (invstd_out) invstd_out[c] = invstd;
    // stat_updates > 1: this forward stands for that many identical forwards of the reference.
    // conv_bias: the statistics are those of the bias-free convolution output; nn.BatchNorm2d saw y + bias
    if (running_mean) {
        float r = running_mean[c];
        const float m = (float)mean + (conv_bias ? conv_bias[c] : 0.f);
        for (int u = 0; u < stat_updates; ++u) r = (1.f - momentum) * r + momentum * m;
        running_mean[c] = r;
    }
    if (running_var) {
        double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        float r = running_var[c];
        for (int u = 0; u < stat_updates; ++u) r = (1.f - momentum) * r + momentum * (float)unbiased;
        running_var[c] = r;
    }
}

// Evaluation-mode BatchNorm2d (module.eval(), running statistics): scale / shift for a bias-free input y,
//   z = (y + bias - running_mean) * gamma * rsqrt(running_var + eps) + beta
__global__ void __launch_bounds__(256)
bn_frozen_kernel(int C, const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ conv_bias,
                 const float *__restrict__ running_mean, const float *__restrict__ running_var, float eps,
                 float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                 float *__restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(running_var[c] + eps);
    const float mu = running_mean[c] - (conv_bias ? conv_bias[c] : 0.f);
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - mu * sc;
    if (mean_out) mean_out[c] = mu;
    if (invstd_out) invstd_out[c] = invstd;
}

// y = act(x * scale[c] + shift[c]) + r1 + r2   (NHWC half, 8 channels per lane)
template <int DT>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const uint4 *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                const uint4 *__restrict__ r1, const uint4 *__restrict__ r2, uint4 *__restrict__ y, long total8,
                int C8, int act) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C8) * 8;
        uint4 v = x[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        if (r1) { uint4 t = r1[i]; a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w; }
        if (r2) { uint4 t = r2[i]; b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w; }
        const float4 sc0 = *reinterpret_cast<const float4 *>(scale + c0), sc1 = *reinterpret_cast<const float4 *>(scale + c0 + 4);
        const float4 sh0 = *reinterpret_cast<const float4 *>(shift + c0), sh1 = *reinterpret_cast<const float4 *>(shift + c0 + 4);
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float f[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t = h2f((uint16_t)(w[j] >> (16 * h)), DT) * sc[2 * j + h] + sh[2 * j + h];
                if (act == 1) t = t > 0.f ? t : 0.f;
                else if (act == 2) t = t > 0.f ? t : 0.2f * t;
                if (r1) t += h2f((uint16_t)(a[j] >> (16 * h)), DT);
                if (r2) t += h2f((uint16_t)(b[j] >> (16 * h)), DT);
                f[h] = t;
            }
            o[j] = (uint32_t)f2h(f[0], DT) | ((uint32_t)f2h(f[1], DT) << 16);
        }
        y[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ----------------------------------------------------------------------------------------
// Statistics + apply in ONE launch, for layers whose convolution wrote few partial rows (the residual blocks: 8192
// pixels = 32 rows).  bn_finalize -> bn_apply are two dependent launches of ~6 and ~16 us for a few MB: the first is
// pure latency.  Here a block owns a 64-channel slice and a pixel range: it reduces the partial rows of ITS 64
// channels itself (rows x 2 x 64 floats, <= 64 KB, L2-resident: every block of a slice reads the same lines), then
// applies scale / shift / activation / residuals to its slice of the pixels (128-byte runs per pixel).  The sums run
// in the same order and precision as bn_finalize_kernel (row order, double), so the result is bit-identical to the
// two-launch path.  Block (slice, 0) also publishes scale / shift / mean / invstd (autograd saves them) and updates
// the running statistics.  (Round 1 measured a variant in which every block reduced ALL channels: 27 us against
// 12 + 10; with the slice it is the apply pass plus ~2 us.)
// ----------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
bn_finalize_apply_kernel(const float *__restrict__ partial, int rows, int C, double count, const float *__restrict__ gamma,
                         const float *__restrict__ beta, const float *__restrict__ conv_bias, float *__restrict__ running_mean,
                         float *__restrict__ running_var, float momentum, float eps, float *__restrict__ scale_out,
                         float *__restrict__ shift_out, float *__restrict__ mean_out, float *__restrict__ invstd_out,
                         int stat_updates, const uint4 *__restrict__ x, const uint4 *__restrict__ r1, const uint4 *__restrict__ r2,
                         uint4 *__restrict__ y, long npix, int act, long pix_per_block) {
    __shared__ double red[2][2][64];
    __shared__ float ssc[64], ssh[64];
    const int cg = blockIdx.x, c0 = cg * 64;
    {
        // 256 threads = (row half, which, channel): the two row halves are summed in row order, first half then second
        const int cl = threadIdx.x & 63, which = (threadIdx.x >> 6) & 1, half = threadIdx.x >> 7;
        const int h0 = half * ((rows + 1) / 2), h1 = half ? rows : (rows + 1) / 2;
        double s = 0.0;
        for (int r = h0; r < h1; r += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[((long)min(r + u, rows - 1) * 2 + which) * C + c0 + cl];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += r + u < h1 ? (double)t[u] : 0.0;
        }
        red[half][which][cl] = s;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x, c = c0 + cl;
        const double s1 = red[0][0][cl] + red[1][0][cl], s2 = red[0][1][cl] + red[1][1][cl];
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = g * invstd, sh = b - (float)mean * sc;
        ssc[cl] = sc;
        ssh[cl] = sh;
        if (blockIdx.y == 0) {
            scale_out[c] = sc;
            shift_out[c] = sh;
            if (mean_out) mean_out[c] = (float)mean;
            if (invstd_out) invstd_out[c] = invstd;
            if (running_mean) {
                float r = running_mean[c];
                const float m = (float)mean + (conv_bias ? conv_bias[c] : 0.f);
                for (int u = 0; u < stat_updates; ++u) r = (1.f - momentum) * r + momentum * m;
                running_mean[c] = r;
            }
            if (running_var) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                float r = running_var[c];
                for (int u = 0; u < stat_updates; ++u) r = (1.f - momentum) * r + momentum * (float)unbiased;
                running_var[c] = r;
            }
        }
    }
    __syncthreads();
    // apply: lane = (pixel row within 32, channel octet of the slice); 16 bytes per lane, 128 bytes per pixel
    const int oc = threadIdx.x & 7, prow = threadIdx.x >> 3;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = ssc[oc * 8 + j]; sh[j] = ssh[oc * 8 + j]; }
    const int C8 = C >> 3;
    const long p_begin = (long)blockIdx.y * pix_per_block, p_end = min(npix, p_begin + pix_per_block);
    for (long p = p_begin + prow; p < p_end; p += 32 * 4) {
        uint4 vx[4], va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = min(p + 32 * u, p_end - 1), i = q * C8 + cg * 8 + oc;      // clamped: masked at the store
            vx[u] = x[i];
            va[u] = r1 ? r1[i] : make_uint4(0, 0, 0, 0);
            vb[u] = r2 ? r2[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = p + 32 * u;
            if (q >= p_end) break;
            const uint32_t w[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w}, a[4] = {va[u].x, va[u].y, va[u].z, va[u].w},
                           b[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float t = h2f((uint16_t)(w[j] >> (16 * h)), DT) * sc[2 * j + h] + sh[2 * j + h];
                    if (act == 1) t = t > 0.f ? t : 0.f;
                    else if (act == 2) t = t > 0.f ? t : 0.2f * t;
                    if (r1) t += h2f((uint16_t)(a[j] >> (16 * h)), DT);
                    if (r2) t += h2f((uint16_t)(b[j] >> (16 * h)), DT);
                    f[h] = t;
                }
                o[j] = (uint32_t)f2h(f[0], DT) | ((uint32_t)f2h(f[1], DT) << 16);
            }
            y[q * C8 + cg * 8 + oc] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ----------------------------------------------------------------------------------------
// layout converters (LDS-tiled transposes: coalesced on both sides)
//   NCHW fp32 [N][C][HW]  <->  NHWC half [N][HW][C]
// tile = 64 pixels x 64 channels per workgroup of 256 threads.
// ----------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_slice_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, int C, long HW, int ld, int c_off,
                          int act) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, n = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        int c = c0 + j; long p = p0 + tx;
        float v = (c < C && p < HW) ? in[((long)n * C + c) * HW + p] : 0.f;
        if (act == 2) v = v > 0.f ? v : 0.1f * v;
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        long p = p0 + j; int c = c0 + tx;
        if (p < HW && c < C) out[((long)n * HW + p) * ld + c_off + c] = f2h(tile[tx][j], DT);
    }
}

template <int DT>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, int C, long HW) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, n = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        int c = c0 + j; long p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? in[((long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        long p = p0 + j; int c = c0 + tx;
        if (p < HW && c < C) out[((long)n * HW + p) * C + c] = f2h(tile[tx][j], DT);
    }
}

template <int DT>
__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, int C, long HW) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, n = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        long p = p0 + j; int c = c0 + tx;
        tile[j][tx] = (p < HW && c < C) ? h2f(in[((long)n * HW + p) * C + c], DT) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int c = c0 + j; long p = p0 + tx;
        if (c < C && p < HW) out[((long)n * C + c) * HW + p] = tile[tx][j];
    }
}

// ----------------------------------------------------------------------------------------
// xexpand: out[n][y][ox][ci*KW + kx] = in[n][ci][y][pad(ox*sx + kx - px)]   (channels >= Cin*KW: 0)
//   in NCHW fp32 (Cin*KW <= 64), out NHWC half with 64 channels.  One lane per 8 output
//   channels (16-byte stores); the NCHW reads of a wave walk x, so they coalesce per (ci,kx).
// ----------------------------------------------------------------------------------------
// One block = one image row (n, y) x XE_TW output columns: the fp32 row segments of all Cin planes
// are staged in LDS with coalesced reads (padding resolved there), then every lane assembles its 8
// output channels -- whose (ci, kx) pairs are fixed for the lane -- and stores 16 B.
#define XE_TW 128
template <int DT>
__global__ void __launch_bounds__(256)
xexpand_kernel(const float *__restrict__ in, uint4 *__restrict__ out, int Cin, int H, int W, int Wout, int KW,
               int sx, int px, int pad_mode, int lanes, int xtiles) {
    extern __shared__ float seg[];                       // [Cin][span]
    const int xt = blockIdx.x % xtiles;
    const long row = blockIdx.x / xtiles;                // n*H + y
    const int n = (int)(row / H), y = (int)(row - (long)n * H);
    const int ox0 = xt * XE_TW;
    const int tw = min(XE_TW, Wout - ox0);
    const int span = (tw - 1) * sx + KW;
    const int ix0 = ox0 * sx - px;
    // all (channel, column) elements of the block as one index range, eight unconditional (clamped) loads in
    // flight per thread: a loop trip per channel cost one dependent HBM round trip each (9..13 per block)
    const int total = Cin * span, pitch = (XE_TW - 1) * sx + KW;
    const float *base = in + ((long)n * Cin * H + y) * W;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
        float v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 256 * u, total - 1);
            const int ci = i / span, t = i - ci * span;
            int ix = ix0 + t;
            bool inb = true;
            if (pad_mode) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= W ? 2 * W - 2 - ix : ix;
            } else {
                inb = ix >= 0 && ix < W;
            }
            ix = max(0, min(ix, W - 1));
            const float x = base[(long)ci * H * W + ix];
            v[u] = inb ? x : 0.f;
            dst[u] = i0 + 256 * u < total ? ci * pitch + t : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) seg[dst[u]] = v[u];
    }
    __syncthreads();
    const int lane = threadIdx.x % lanes, slot = threadIdx.x / lanes, slots = 256 / lanes;
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 8 + j;
        const int ci = c / KW, kx = c - ci * KW;
        off[j] = c < Cin * KW ? ci * ((XE_TW - 1) * sx + KW) + kx : -1;
    }
    uint4 *dst = out + (row * Wout + ox0) * lanes + lane;
    for (int o = slot; o < tw; o += slots) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = off[2 * q] >= 0 ? seg[off[2 * q] + o * sx] : 0.f;
            const float v1 = off[2 * q + 1] >= 0 ? seg[off[2 * q + 1] + o * sx] : 0.f;
            w[q] = (uint32_t)f2h(v0, DT) | ((uint32_t)f2h(v1, DT) << 16);
        }
        dst[(long)o * lanes] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
extern "C" int ir2rgb_bn_finalize(const float *stats_partial, int rows, int C, long count, const float *gamma,
                                  const float *beta, float *running_mean, float *running_var, float momentum,
                                  float eps, float *scale, float *shift, float *mean_out, float *invstd_out,
                                  int stat_updates, void *stream) {
    return ir2rgb_bn_finalize_ex(stats_partial, rows, C, count, gamma, beta, nullptr, running_mean, running_var, momentum,
                                 eps, scale, shift, mean_out, invstd_out, stat_updates, 0, stream);
}

extern "C" int ir2rgb_bn_finalize_ex(const float *stats_partial, int rows, int C, long count, const float *gamma,
                                     const float *beta, const float *conv_bias, float *running_mean,
                                     float *running_var, float momentum, float eps, float *scale, float *shift,
                                     float *mean_out, float *invstd_out, int stat_updates, int frozen, void *stream) {
    if (C < 1 || !scale || !shift) return IR2RGB_EINVAL;
    if (frozen) {
        if (!running_mean || !running_var) return IR2RGB_EINVAL;
        bn_frozen_kernel<<<cdiv(C, 256), 256, 0, as_stream(stream)>>>(C, gamma, beta, conv_bias, running_mean, running_var,
                                                                      eps, scale, shift, mean_out, invstd_out);
        return ir2rgb_launch_status();
    }
    if (rows < 1 || count < 1 || !stats_partial || stat_updates < 1) return IR2RGB_EINVAL;
    static int narrow = -1;     // IR2RGB_BN_FINALIZE_NARROW=0: 32-channel workgroups for every layer (A/B measurements)
    if (narrow < 0) { const char *e = getenv("IR2RGB_BN_FINALIZE_NARROW"); narrow = e ? atoi(e) : 1; }
    if (narrow && rows >= 2048 && C <= 256)
        bn_finalize_kernel<8><<<cdiv(C, 8), 1024, 0, as_stream(stream)>>>(stats_partial, rows, C, (double)count, gamma, beta,
                                                                         conv_bias, running_mean, running_var, momentum, eps,
                                                                         scale, shift, mean_out, invstd_out, stat_updates);
    else
        bn_finalize_kernel<32><<<cdiv(C, 32), 1024, 0, as_stream(stream)>>>(stats_partial, rows, C, (double)count, gamma, beta,
                                                                           conv_bias, running_mean, running_var, momentum, eps,
                                                                           scale, shift, mean_out, invstd_out, stat_updates);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_bn_finalize_apply(const float *stats_partial, int rows, int C, long count, const float *gamma,
                                        const float *beta, const float *conv_bias, float *running_mean,
                                        float *running_var, float momentum, float eps, float *scale, float *shift,
                                        float *mean_out, float *invstd_out, int stat_updates, const void *x,
                                        const void *res1, const void *res2, void *y, long npix, int act, int dtype,
                                        void *stream) {
    if (rows < 1 || rows > IR2RGB_BN_FUSED_MAX_ROWS || C < 64 || (C % 64) || count < 1 || npix < 1 || !stats_partial ||
        !scale || !shift || !x || !y || stat_updates < 1 || act < 0 || act > 2)
        return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res1 | (uintptr_t)res2) & 15)) return IR2RGB_EALIGN;
    // about 2048 blocks in all, at least 128 pixels each
    long chunks = 2048 / (C / 64);
    const long cap = (npix + 127) / 128;
    chunks = chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
    const long per = (npix + chunks - 1) / chunks;
    dim3 grid((unsigned)(C / 64), (unsigned)((npix + per - 1) / per));
    hipStream_t s = as_stream(stream);
    if (dtype == IR2RGB_BF16)
        bn_finalize_apply_kernel<IR2RGB_BF16><<<grid, 256, 0, s>>>(stats_partial, rows, C, (double)count, gamma, beta, conv_bias,
            running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out, stat_updates, (const uint4 *)x,
            (const uint4 *)res1, (const uint4 *)res2, (uint4 *)y, npix, act, per);
    else
        bn_finalize_apply_kernel<IR2RGB_F16><<<grid, 256, 0, s>>>(stats_partial, rows, C, (double)count, gamma, beta, conv_bias,
            running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out, stat_updates, (const uint4 *)x,
            (const uint4 *)res1, (const uint4 *)res2, (uint4 *)y, npix, act, per);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_bn_apply(const void *x, const float *scale, const float *shift, const void *res1,
                               const void *res2, void *y, long npix, int C, int act, int dtype, void *stream) {
    if (npix < 0 || C < 8 || (C % 8) || act < 0 || act > 2) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res1 | (uintptr_t)res2 | (uintptr_t)scale | (uintptr_t)shift) & 15))
        return IR2RGB_EALIGN;
    long total8 = npix * (C / 8);
    if (total8 == 0) return IR2RGB_OK;
    int grid = stream_grid(total8, 256);
    if (dtype == IR2RGB_BF16)
        bn_apply_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>((const uint4 *)x, scale, shift, (const uint4 *)res1,
                                                                          (const uint4 *)res2, (uint4 *)y, total8, C / 8, act);
    else
        bn_apply_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>((const uint4 *)x, scale, shift, (const uint4 *)res1,
                                                                         (const uint4 *)res2, (uint4 *)y, total8, C / 8, act);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_nchw_f32_to_nhwc_half(const float *in, void *out, int N, int C, int H, int W, int dtype,
                                            void *stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (N == 0) return IR2RGB_OK;
    long HW = (long)H * W;
    dim3 grid((unsigned)cdiv(HW, 64), (unsigned)cdiv(C, 64), (unsigned)N);
    if (dtype == IR2RGB_BF16) nchw_to_nhwc_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>(in, (uint16_t *)out, C, HW);
    else nchw_to_nhwc_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>(in, (uint16_t *)out, C, HW);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_nhwc_half_to_nchw_f32(const void *in, float *out, int N, int C, int H, int W, int dtype,
                                            void *stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (N == 0) return IR2RGB_OK;
    long HW = (long)H * W;
    dim3 grid((unsigned)cdiv(HW, 64), (unsigned)cdiv(C, 64), (unsigned)N);
    if (dtype == IR2RGB_BF16) nhwc_to_nchw_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t *)in, out, C, HW);
    else nhwc_to_nchw_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t *)in, out, C, HW);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_xexpand_cx(const float *in, void *out, int N, int Cin, int H, int W, int Wout, int KW,
                                 int stride_w, int pad_w, int pad_mode, int Cx, int dtype, void *stream) {
    if (N < 0 || Cin < 1 || H < 1 || W < 1 || Wout < 1 || KW < 1 || (Cx != 64 && Cx != 128) || Cin * KW > Cx ||
        stride_w < 1 || pad_w < 0)
        return IR2RGB_EINVAL;
    if (pad_mode == 1 && pad_w >= W) return IR2RGB_EINVAL;
    if ((Wout - 1) * stride_w + KW - pad_w > W + pad_w) return IR2RGB_EINVAL;  // would read past the padded row
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    const int lanes = Cx / 8;
    if ((long)N * H * Wout == 0) return IR2RGB_OK;
    const int xtiles = cdiv(Wout, XE_TW);
    const long blocks = (long)N * H * xtiles;
    if (blocks > 0x7fffffffL) return IR2RGB_EINVAL;
    const size_t lds = (size_t)Cin * ((XE_TW - 1) * stride_w + KW) * sizeof(float);
    if (lds > 64 * 1024) return IR2RGB_ENOSUP;
    if (dtype == IR2RGB_BF16)
        xexpand_kernel<IR2RGB_BF16><<<(unsigned)blocks, 256, lds, as_stream(stream)>>>(in, (uint4 *)out, Cin, H, W, Wout, KW, stride_w, pad_w, pad_mode, lanes, xtiles);
    else
        xexpand_kernel<IR2RGB_F16><<<(unsigned)blocks, 256, lds, as_stream(stream)>>>(in, (uint4 *)out, Cin, H, W, Wout, KW, stride_w, pad_w, pad_mode, lanes, xtiles);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_xexpand(const float *in, void *out, int N, int Cin, int H, int W, int Wout, int KW, int stride_w,
                              int pad_w, int pad_mode, int dtype, void *stream) {
    return ir2rgb_xexpand_cx(in, out, N, Cin, H, W, Wout, KW, stride_w, pad_w, pad_mode, 64, dtype, stream);
}

extern "C" int ir2rgb_nchw_f32_to_nhwc_half_slice(const float *in, void *out, int N, int C, int H, int W, int ld,
                                                  int c_off, int act, int dtype, void *stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1 || ld < C || c_off < 0 || c_off + C > ld || (act != 0 && act != 2)) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (N == 0) return IR2RGB_OK;
    long HW = (long)H * W;
    dim3 grid((unsigned)cdiv(HW, 64), (unsigned)cdiv(C, 64), (unsigned)N);
    if (dtype == IR2RGB_BF16) nchw_to_nhwc_slice_kernel<IR2RGB_BF16><<<grid, 256, 0, as_stream(stream)>>>(in, (uint16_t *)out, C, HW, ld, c_off, act);
    else nchw_to_nhwc_slice_kernel<IR2RGB_F16><<<grid, 256, 0, as_stream(stream)>>>(in, (uint16_t *)out, C, HW, ld, c_off, act);
    return ir2rgb_launch_status();
}


// ------------------------------------------------------------------------------------------
// AvgPool2d(3, stride 2, padding 1, count_include_pad=False) on fp32 planes -- the image pyramids of the multi-scale
// discriminators (reference networks.py:639, :658-666) and of the generator inputs (base_model.py:64-82), forward and
// backward.  (torch's kernels take 38 / 102 us per call on three 6-channel 512x1024 frames; the tensors are 10 us of HBM time.)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
avgpool3s2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int Ho, int Wo, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const long r = i / Wo;
        const int oy = (int)(r % Ho);
        const long p = r / Ho;
        const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 1, H - 1), x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 1, W - 1);
        const float *src = x + p * (long)H * W;
        float s = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) s += src[(long)yy * W + xx];
        y[i] = s / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
    }
}

__global__ void __launch_bounds__(256)
avgpool3s2_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx, int H, int W, int Ho, int Wo, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        const long r = i / W;
        const int yy = (int)(r % H);
        const long p = r / H;
        // outputs whose 3x3 window (rows 2oy-1 .. 2oy+1) contains this pixel: oy in [ceil((yy-1)/2), floor((yy+1)/2)]
        const int oy0 = yy >> 1, oy1 = min((yy + 1) >> 1, Ho - 1), ox0 = xx >> 1, ox1 = min((xx + 1) >> 1, Wo - 1);
        const float *src = gy + p * (long)Ho * Wo;
        float s = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const int ny = min(2 * oy + 1, H - 1) - max(2 * oy - 1, 0) + 1;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const int nx = min(2 * ox + 1, W - 1) - max(2 * ox - 1, 0) + 1;
                s += src[(long)oy * Wo + ox] / (float)(ny * nx);
            }
        }
        gx[i] = s;
    }
}

extern "C" int ir2rgb_avgpool3s2(const float *x, float *y, long planes, int H, int W, int backward, void *stream) {
    if (!x || !y || planes < 0 || H < 1 || W < 1) return IR2RGB_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t s = as_stream(stream);
    if (!backward) {      // x [planes][H][W] -> y [planes][Ho][Wo]
        const long total = planes * Ho * Wo;
        if (total) avgpool3s2_fwd_kernel<<<stream_grid(total, 256), 256, 0, s>>>(x, y, H, W, Ho, Wo, total);
    } else {              // x = gradient [planes][Ho][Wo] -> y = input gradient [planes][H][W]
        const long total = planes * H * W;
        if (total) avgpool3s2_bwd_kernel<<<stream_grid(total, 256), 256, 0, s>>>(x, y, H, W, Ho, Wo, total);
    }
    return ir2rgb_launch_status();
}

// dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 -- a rearranged fp32 copy of a weight (x-im2col of the first layers, zero-padded
// widths, the separable heads' row split) as ONE launch from a precomputed index map (ir2rgb_amd.layers.packed_weight),
// instead of the chain of torch permute / reshape / pad / cat launches it replaces after every optimizer step.
__global__ void __launch_bounds__(256)
gather_f32_kernel(const float *__restrict__ src, const int *__restrict__ idx, float *__restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = idx[i];
        dst[i] = j >= 0 ? src[j] : 0.f;
    }
}

extern "C" int ir2rgb_gather_f32(const float *src, const int *idx, float *dst, long n, void *stream) {
    if (n < 0 || (n > 0 && (!src || !idx || !dst))) return IR2RGB_EINVAL;
    if (n) gather_f32_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(src, idx, dst, n);
    return ir2rgb_launch_status();
}
