// backward.hip -- HBM-bound backward companions of the MFMA convolution (gfx950):
//   * bn_bwd_reduce / bn_bwd_finalize / bn_bwd_apply : backward of activation + training-mode
//     BatchNorm2d on NHWC half tensors, producing the gradient w.r.t. the convolution output
//     plus dgamma / dbeta (and the bias gradient of norm-less stages);
//   * fold_reflect : adjoint of nn.ReflectionPad2d -- folds the border of the padded-domain
//     gradient produced by the data-gradient convolution back into the image;
//   * xexpand_bwd  : adjoint of the x-direction im2col of the first layers.
// Reductions are two-level and deterministic (per-block partial rows, then a finalize kernel).
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

__device__ __forceinline__ void unpack8(const uint4 &v, float *f, int dt) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = h2f((uint16_t)(w[j] & 0xffff), dt);
        f[2 * j + 1] = h2f((uint16_t)(w[j] >> 16), dt);
    }
}
__device__ __forceinline__ uint4 pack8(const float *f, int dt) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2h(f[2 * j], dt) | ((uint32_t)f2h(f[2 * j + 1], dt) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// g' = gz * act'(pre) where pre = y*scale+shift (BN output) or y itself (scale == nullptr)
__device__ __forceinline__ float act_grad(float g, float pre, int act) {
    if (act == 1) return pre > 0.f ? g : 0.f;
    if (act == 2) return pre > 0.f ? g : 0.2f * g;
    return g;
}

// One block (512 threads) = one 64-channel group x one pixel range.  partial[r][0][c] = sum g',
// partial[r][1][c] = sum g' * yhat over range r.  Thread t owns channel octet t & 7 of the group
// and pixel rows (t >> 3) + 64k of the range, four rows in flight: a dependent HBM round trip
// costs ~2 us here, so these small reductions are shaped to need as few of them as possible.
// (A last-block-finalises variant was measured: the device-scope fences it needs write back the
// XCD L2s and made this kernel 4x slower than reduce + a separate finalize launch.  With 1024 threads the
// 128-VGPR cap made the compiler spill 18 registers: scratch costs dispatch latency, ~14 us floor per call.)
__global__ void __launch_bounds__(512)
bn_bwd_reduce_kernel(const uint4 *__restrict__ gz, const uint4 *__restrict__ y, const float *__restrict__ scale,
                     const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
                     float *__restrict__ partial, long npix, int C, int act, int dt, long pix_per_block) {
    __shared__ float red[2][8][64];
    const int octs = C >> 3, ngroups = C >> 6;
    const int cg = blockIdx.x % ngroups, r = blockIdx.x / ngroups;
    const int oc = threadIdx.x & 7, row = threadIdx.x >> 3, wave = threadIdx.x >> 6;
    const long p_begin = (long)r * pix_per_block;
    const long p_end = min(npix, p_begin + pix_per_block);
    const int c0 = cg * 64 + oc * 8;
    float s1[8], s2[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        sc[j] = scale ? scale[c0 + j] : 1.f;
        sh[j] = shift ? shift[c0 + j] : 0.f;
        mu[j] = mean ? mean[c0 + j] : 0.f;
        is[j] = invstd ? invstd[c0 + j] : 1.f;
    }
    const long lane_off = (long)cg * 8 + oc;
    for (long p = p_begin + row; p < p_end; p += 256) {
        uint4 gq[4], yq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = p + 64 * u;
            const bool ok = q < p_end;
            gq[u] = ok ? gz[q * octs + lane_off] : make_uint4(0, 0, 0, 0);
            yq[u] = ok ? y[q * octs + lane_off] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float g[8], v[8];
            unpack8(gq[u], g, dt);
            unpack8(yq[u], v, dt);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gp = act_grad(g[j], v[j] * sc[j] + sh[j], act);   // g == 0 for the padding rows
                s1[j] += gp;
                s2[j] += gp * (v[j] - mu[j]) * is[j];
            }
        }
    }
    // the 8 pixel rows of a wave (lane bits 3..5), then the 8 waves through LDS in wave order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            s1[j] += __shfl_xor(s1[j], m);
            s2[j] += __shfl_xor(s2[j], m);
        }
    }
    if ((threadIdx.x & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][wave][oc * 8 + j] = s1[j]; red[1][wave][oc * 8 + j] = s2[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cl = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[which][k][cl];
        partial[((long)r * 2 + which) * C + cg * 64 + cl] = s;
    }
}

// One block (1024 threads) per 64-channel group: dbeta[c] = sum_r partial[r][0][c],
// dgamma[c] = sum_r partial[r][1][c] (R <= 128 rows: thread = (value, slice of 16 rows), every load
// of a thread in flight at once; slices and rows are added in a fixed order, in double), and the
// three coefficient vectors of the apply pass:  gy = cA*g' + cB*y + cC
//   cA = scale, cB = -scale*invstd*dgamma/n, cC = scale*(invstd*mean*dgamma/n - dbeta/n).
__global__ void __launch_bounds__(1024)
bn_bwd_finalize_kernel(const float *__restrict__ partial, int R, int C, const float *__restrict__ scale,
                       const float *__restrict__ mean, const float *__restrict__ invstd, float inv_n,
                       float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef, int acc) {
    __shared__ double fin[8][2][64];
    const int cg = blockIdx.x;
    {
        const int slice = threadIdx.x >> 7, which = (threadIdx.x >> 6) & 1, cl = threadIdx.x & 63;
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {     // unconditional loads of a clamped row, masked below: a predicated load
            const int k = min(slice + 8 * u, R - 1);   // compiles to a branch with its own s_waitcnt (16 serial round trips)
            t[u] = partial[((long)k * 2 + which) * C + cg * 64 + cl];
        }
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) a += slice + 8 * u < R ? (double)t[u] : 0.0;
        fin[slice][which][cl] = a;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x, c = cg * 64 + cl;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += fin[k][0][cl]; b += fin[k][1][cl]; }
        // (acc: the parameter gradients of an earlier sample group of the same layer are already there)
        dbeta[c] = (acc ? dbeta[c] : 0.f) + (float)a;
        dgamma[c] = (acc ? dgamma[c] : 0.f) + (float)b;
        if (scale) {
            const float scv = scale[c], isv = invstd[c], muv = mean[c];
            const float dg = (float)b * inv_n, db = (float)a * inv_n;
            coef[c] = scv;
            coef[C + c] = -scv * isv * dg;
            coef[2 * C + c] = scv * (isv * muv * dg - db);
        }
    }
}

// gy = cA*g' + cB*y + cC, g' = gz * act'(y*scale + shift)   (BatchNorm stage; coef = [cA|cB|cC])
// gy = g' = gz * act'(y)                                     (scale == nullptr: activation only)
// The grid-stride (a multiple of 256 items) is a multiple of C/8, so a lane keeps its channel octet
// for the whole loop and the per-channel vectors are loaded once.
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const uint4 *__restrict__ gz, const uint4 *__restrict__ y, const float *__restrict__ scale,
                    const float *__restrict__ shift, const float *__restrict__ coef, uint4 *__restrict__ gy,
                    long total8, int C8, int act, int dt) {
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c0 = (int)(i0 % C8) * 8, C = C8 * 8;
    float sc[8], sh[8], cA[8], cB[8], cC[8];
    if (scale) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = *reinterpret_cast<const float4 *>(scale + c0 + 4 * h), b = *reinterpret_cast<const float4 *>(shift + c0 + 4 * h);
            const float4 k0 = *reinterpret_cast<const float4 *>(coef + c0 + 4 * h), k1 = *reinterpret_cast<const float4 *>(coef + C + c0 + 4 * h),
                         k2 = *reinterpret_cast<const float4 *>(coef + 2 * C + c0 + 4 * h);
            sc[4 * h] = a.x; sc[4 * h + 1] = a.y; sc[4 * h + 2] = a.z; sc[4 * h + 3] = a.w;
            sh[4 * h] = b.x; sh[4 * h + 1] = b.y; sh[4 * h + 2] = b.z; sh[4 * h + 3] = b.w;
            cA[4 * h] = k0.x; cA[4 * h + 1] = k0.y; cA[4 * h + 2] = k0.z; cA[4 * h + 3] = k0.w;
            cB[4 * h] = k1.x; cB[4 * h + 1] = k1.y; cB[4 * h + 2] = k1.z; cB[4 * h + 3] = k1.w;
            cC[4 * h] = k2.x; cC[4 * h + 1] = k2.y; cC[4 * h + 2] = k2.z; cC[4 * h + 3] = k2.w;
        }
    }
    for (long i = i0; i < total8; i += (long)gridDim.x * blockDim.x) {
        float g[8], v[8], o[8];
        unpack8(gz[i], g, dt);
        unpack8(y[i], v, dt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (scale) o[j] = cA[j] * act_grad(g[j], v[j] * sc[j] + sh[j], act) + cB[j] * v[j] + cC[j];
            else o[j] = act_grad(g[j], v[j], act);
        }
        gy[i] = pack8(o, dt);
    }
}

// bn_bwd_finalize + bn_bwd_apply in one launch (BatchNorm stages): a block owns a 64-channel slice and a pixel range;
// it sums the R <= 128 partial rows of ITS 64 channels (same order and precision as bn_bwd_finalize_kernel: slices of
// 8 rows apart, then the 8 slices, in double), derives the three coefficients, and applies them to its pixels.
// Block (slice, 0) publishes dgamma / dbeta.  One dependent launch (~5 us of pure latency) less per BatchNorm layer
// and backward pass: 150 of them per training window.
__global__ void __launch_bounds__(256)
bn_bwd_finalize_apply_kernel(const float *__restrict__ partial, int R, int C, const float *__restrict__ scale,
                             const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
                             float inv_n, float *__restrict__ dgamma, float *__restrict__ dbeta, const uint4 *__restrict__ gz,
                             const uint4 *__restrict__ y, uint4 *__restrict__ gy, long npix, int act, int dt, long pix_per_block,
                             int acc) {
    __shared__ double fin[2][2][64];
    __shared__ float co[3][64], ssc[64], ssh[64];
    const int cg = blockIdx.x, c0 = cg * 64;
    {
        // 256 threads = (group of slices, which, channel): group h sums the slices k = 4h .. 4h+3 of bn_bwd_finalize_kernel's
        // order (slice k = rows k, k+8, k+16, ...), so that (group 0) + (group 1) = its slice-ordered total
        const int cl = threadIdx.x & 63, which = (threadIdx.x >> 6) & 1, h = threadIdx.x >> 7;
        double a = 0.0;
        for (int k = 4 * h; k < 4 * h + 4; ++k) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = partial[((long)min(k + 8 * u, R - 1) * 2 + which) * C + c0 + cl];
            double sl = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) sl += k + 8 * u < R ? (double)t[u] : 0.0;
            a += sl;
        }
        fin[h][which][cl] = a;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cl = threadIdx.x, c = c0 + cl;
        const double a = fin[0][0][cl] + fin[1][0][cl], b = fin[0][1][cl] + fin[1][1][cl];
        const float scv = scale[c], isv = invstd[c], muv = mean[c];
        const float dg = (float)b * inv_n, db = (float)a * inv_n;
        co[0][cl] = scv;
        co[1][cl] = -scv * isv * dg;
        co[2][cl] = scv * (isv * muv * dg - db);
        ssc[cl] = scv;
        ssh[cl] = shift[c];
        if (blockIdx.y == 0) { dbeta[c] = (acc ? dbeta[c] : 0.f) + (float)a; dgamma[c] = (acc ? dgamma[c] : 0.f) + (float)b; }
    }
    __syncthreads();
    const int oc = threadIdx.x & 7, prow = threadIdx.x >> 3;
    float sc[8], sh[8], cA[8], cB[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = ssc[oc * 8 + j]; sh[j] = ssh[oc * 8 + j];
        cA[j] = co[0][oc * 8 + j]; cB[j] = co[1][oc * 8 + j]; cC[j] = co[2][oc * 8 + j];
    }
    const int C8 = C >> 3;
    const long p_begin = (long)blockIdx.y * pix_per_block, p_end = min(npix, p_begin + pix_per_block);
    for (long p = p_begin + prow; p < p_end; p += 32 * 4) {
        uint4 gq[4], yq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = min(p + 32 * u, p_end - 1), i = q * C8 + cg * 8 + oc;
            gq[u] = gz[i];
            yq[u] = y[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = p + 32 * u;
            if (q >= p_end) break;
            float g[8], v[8], o[8];
            unpack8(gq[u], g, dt);
            unpack8(yq[u], v, dt);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = cA[j] * act_grad(g[j], v[j] * sc[j] + sh[j], act) + cB[j] * v[j] + cC[j];
            gy[q * C8 + cg * 8 + oc] = pack8(o, dt);
        }
    }
}

// dx[n][y][x][c] = sum over padded positions (py,px) that reflect onto (y,x) of dxpad[n][py][px][c]
__global__ void __launch_bounds__(256)
fold_reflect_kernel(const uint4 *__restrict__ dxpad, uint4 *__restrict__ dx, int H, int W, int C8, int pady, int padx,
                    long total8, int dt) {
    const int Hp = H + 2 * pady, Wp = W + 2 * padx;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int c8 = (int)(r % C8); r /= C8;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long n = r / H;
        // candidate padded rows: y+pad (identity), pad-y (top mirror, 1<=y<=pad), pad+2H-2-y (bottom mirror)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + pady;
        if (y >= 1 && y <= pady) ys[ny++] = pady - y;
        if (y <= H - 2 && y >= H - 1 - pady) ys[ny++] = pady + 2 * H - 2 - y;
        xs[nx++] = x + padx;
        if (x >= 1 && x <= padx) xs[nx++] = padx - x;
        if (x <= W - 2 && x >= W - 1 - padx) xs[nx++] = padx + 2 * W - 2 - x;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
                float f[8];
                unpack8(dxpad[((n * Hp + ys[a]) * (long)Wp + xs[b]) * C8 + c8], f, dt);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        dx[i] = pack8(acc, dt);
    }
}

// din[n][ci][y][ix] = sum_{kx, ox : pad(ox*sx + kx - px) == ix} dxe[n][y][ox][ci*KW + kx]   (fp32 NCHW out)
// One thread per input pixel (n, y, ix) and ALL its channels: the <= 3*KW expanded pixels that feed it are 128-byte rows,
// and the Cin values wanted from a row sit in that one row -- one thread per (channel, pixel) fetched every row Cin times
// from waves far apart (84 us for a 512x1024 image at 6 channels; the tensors are 20 us of HBM time).
#define XB_MAXC 32
__global__ void __launch_bounds__(256)
xexpand_bwd_kernel(const uint16_t *__restrict__ dxe, float *__restrict__ din, int Cin, int H, int W, int Wout, int KW,
                   int sx, int px, int pad_mode, long total, int dt) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int ix = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long n = r / H;
        float acc[XB_MAXC];
#pragma unroll
        for (int c = 0; c < XB_MAXC; ++c) acc[c] = 0.f;
        // unpadded source positions that map to ix: ix itself, and its mirror images under reflection
        int cand[3], nc = 0;
        cand[nc++] = ix;
        if (pad_mode) {
            if (ix >= 1) cand[nc++] = -ix;
            if (ix <= W - 2) cand[nc++] = 2 * W - 2 - ix;
        }
        for (int kx = 0; kx < KW; ++kx)
            for (int q = 0; q < nc; ++q) {
                const int t = cand[q] + px - kx;  // = ox * sx
                if (t < 0 || (t % sx) != 0) continue;
                const int ox = t / sx;
                if (ox >= Wout) continue;
                const int src = ox * sx + kx - px;  // must lie inside the padded range actually read in forward
                if (src < -px || src > W - 1 + px) continue;
                const uint16_t *row = dxe + ((n * H + y) * (long)Wout + ox) * 64 + kx;
#pragma unroll
                for (int c = 0; c < XB_MAXC; ++c)
                    if (c < Cin) acc[c] += h2f(row[c * KW], dt);
            }
#pragma unroll
        for (int c = 0; c < XB_MAXC; ++c)
            if (c < Cin) din[((n * Cin + c) * H + y) * (long)W + ix] = acc[c];
    }
}

// The same sum with the expanded rows staged once: one workgroup = one image row (n, y) x XB_TW input columns.  The
// <= XB_TW/sx + KW + 2 px expanded pixels that feed the tile are copied to LDS as whole 128-byte rows (16-byte loads, every
// byte of dxe read once from HBM instead of ~KW times through the caches from scattered 2-byte loads: 78 -> ~25 us for a
// 512x1024 image), then thread (ix, channel parity) gathers its values from LDS in the per-pixel kernel's order (kx outer,
// mirror candidates inner: bit-identical sums).  Rows are pitched 33 dwords apart (consecutive pixels on different banks).
#define XB_TW 128
#define XB_PITCH 33
__global__ void __launch_bounds__(256)
xexpand_bwd_tile_kernel(const uint4 *__restrict__ dxe, float *__restrict__ din, int Cin, int H, int W, int Wout, int KW,
                        int sx, int px, int pad_mode, int xtiles, int dt) {
    extern __shared__ uint32_t xrows[];                    // [rows][XB_PITCH]
    const int xt = blockIdx.x % xtiles;
    const long row = blockIdx.x / xtiles;                  // n*H + y
    const long n = row / H;
    const int y = (int)(row - n * H);
    const int ix0 = xt * XB_TW, tw = min(XB_TW, W - ix0), ixl = ix0 + tw - 1;
    // expanded columns ox with ox*sx in [cand + px - (KW-1), cand + px] for a candidate source column of the tile
    int tlo = ix0 + px - (KW - 1), thi = ixl + px;
    if (pad_mode && ixl >= W - 1 - px && ix0 <= W - 2) thi = max(thi, 2 * W - 2 - max(ix0, W - 1 - px) + px);   // right mirrors
    if (pad_mode && ix0 <= px) tlo = min(tlo, 0);                                                                // left mirrors
    const int oxlo = tlo <= 0 ? 0 : (tlo + sx - 1) / sx;
    const int oxhi = min(Wout - 1, thi / sx);
    const int nrows = oxhi - oxlo + 1;
    const uint4 *src = dxe + (row * Wout + oxlo) * 8;
    for (int i0 = threadIdx.x; i0 < nrows * 8; i0 += 256 * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + 256 * u, nrows * 8 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < nrows * 8) {
                uint32_t *d = xrows + (i >> 3) * XB_PITCH + (i & 7) * 4;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
    __syncthreads();
    const int il = threadIdx.x & (XB_TW - 1), par = threadIdx.x >> 7;      // channels par, par + 2, ...
    if (il >= tw) return;
    const int ix = ix0 + il;
    float acc[XB_MAXC / 2];
#pragma unroll
    for (int j = 0; j < XB_MAXC / 2; ++j) acc[j] = 0.f;
    int cand[3], nc = 0;
    cand[nc++] = ix;
    if (pad_mode) {
        if (ix >= 1) cand[nc++] = -ix;
        if (ix <= W - 2) cand[nc++] = 2 * W - 2 - ix;
    }
    for (int kx = 0; kx < KW; ++kx)
        for (int q = 0; q < nc; ++q) {
            const int t = cand[q] + px - kx;  // = ox * sx
            if (t < 0 || (t % sx) != 0) continue;
            const int ox = t / sx;
            if (ox >= Wout) continue;
            const int s = ox * sx + kx - px;  // must lie inside the padded range actually read in forward
            if (s < -px || s > W - 1 + px) continue;
            const uint32_t *r = xrows + (ox - oxlo) * XB_PITCH;
#pragma unroll
            for (int j = 0; j < XB_MAXC / 2; ++j) {
                const int c = par + 2 * j;
                if (c < Cin) {
                    const int e = c * KW + kx;
                    const uint32_t w = r[e >> 1];
                    acc[j] += h2f((uint16_t)((e & 1) ? (w >> 16) : (w & 0xffff)), dt);
                }
            }
        }
#pragma unroll
    for (int j = 0; j < XB_MAXC / 2; ++j) {
        const int c = par + 2 * j;
        if (c < Cin) din[((n * Cin + c) * H + y) * (long)W + ix] = acc[j];
    }
}

// Gradient of a thin fp32 output (the 1-channel PatchGAN logits, Cout <= 8) prepared for the MFMA kernels in
// one pass: g64 = the gradient as a 64-channel NHWC half tensor (data-gradient operand, channels >= Cout zero),
// g8 = the same in 8 channels (weight-gradient operand), dbias[c] = sum of the fp32 gradient.  Eight lanes
// share a pixel (one 16-byte store each into its 128-byte g64 row); the last Cout blocks reduce one channel
// each in a fixed order.
__global__ void __launch_bounds__(256)
thin_grad_expand_kernel(const float *__restrict__ gz, uint4 *__restrict__ g64, uint4 *__restrict__ g8,
                        float *__restrict__ dbias, long npix, long hw, int Cout, int dt, int expand_blocks) {
    if ((int)blockIdx.x >= expand_blocks) {
        __shared__ float red[256];
        const int c = blockIdx.x - expand_blocks;
        float s = 0.f;
        for (long p = threadIdx.x; p < npix; p += 256 * 8) {      // eight loads in flight (clamped, masked), index order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long pp = min(p + 256 * u, npix - 1);
                const long n = pp / hw;
                v[u] = gz[(n * Cout + c) * hw + (pp - n * hw)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += p + 256 * u < npix ? v[u] : 0.f;
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) dbias[c] = red[0];
        return;
    }
    const long total = npix * 8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += expand_blocks * 256L) {
        const long p = i >> 3;
        const int o8 = (int)(i & 7);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o8 == 0) {
            const long n = p / hw, q = p - n * hw;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = j < Cout ? gz[(n * Cout + j) * hw + q] : 0.f;
            v = pack8(f, dt);
            g8[p] = v;
        }
        g64[i] = v;
    }
}

// R pixel ranges per 64-channel group: about 512 blocks of 512 threads in all, at least 128 pixels
// per range and at most 128 ranges (bn_bwd_finalize_kernel's 8 slices x 16 rows).
static int bn_bwd_ranges(long npix, int C) {
    long want = 512 / (C / 64);
    want = want < 8 ? 8 : (want > 128 ? 128 : want);
    const long cap = (npix + 127) / 128;
    return (int)(want < cap ? want : cap);
}

// ----------------------------------------------------------------------------------------
// BatchNorm backward in ONE launch for layers with few pixels (the 1024-channel residual blocks at 32x64, the deep
// discriminator layers: half of the BatchNorm layers of a training window).  The two-launch form above needs the
// per-channel sums of the WHOLE tensor before any gradient can be written, i.e. a kernel boundary between reduce and
// apply; each of the two launches then costs ~13 us of dependent latency on tensors of a few MB.  Here a workgroup owns a
// group of CPB channels over ALL pixels: its slab of gz and y (P x CPB halfs each) fits the registers of 512 threads
// (K 16-byte loads per tensor and thread, all in flight at once; 512 threads leave a thread 256 registers -- at 1024 the
// 128-register cap spilled from K = 4 on, and scratch costs dispatch latency), so it sums, derives the coefficients and
// applies them without the tensors leaving the CU -- one read of gz and y, one write of gy, no partial rows, no second launch.
//   thread t: channel octet t % (CPB/8), pixels t / (CPB/8) + k * (512 / (CPB/8)), k < K
// Sums: per thread in fp32, across the lanes of a wave by shuffles, across the 8 waves in wave order in double.
// ----------------------------------------------------------------------------------------
template <int K, int CPB>
__global__ void __launch_bounds__(512)
bn_bwd_onepass_kernel(const uint4 *__restrict__ gz, const uint4 *__restrict__ y, const float *__restrict__ scale,
                      const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
                      uint4 *__restrict__ gy, float *__restrict__ dgamma, float *__restrict__ dbeta, long npix, int C, int act,
                      int dt, float inv_n, int acc, int xcd_remap) {
    constexpr int LPP = CPB / 8;                 // lanes per pixel
    constexpr int PPP = 512 / LPP;               // pixels per pass
    __shared__ double red[2][8][CPB];
    __shared__ float co[3][CPB];
    const int t = threadIdx.x, oc = t % LPP, prow = t / LPP, wave = t >> 6, lane = t & 63;
    // Workgroups whose channel groups share 128-byte lines (64 / CPB neighbours) go to ONE XCD (blockIdx % 8 under
    // round-robin placement): each L2 then fetches a line once and merges the 16-byte stores into whole lines, instead
    // of eight L2s fetching and partially writing every line (speed only: any placement gives the same result).
    const int G = gridDim.x;
    const int grp = xcd_remap && (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int c0 = grp * CPB + oc * 8, C8 = C >> 3;
    const long cidx = (long)grp * LPP + oc;
    uint4 gq[K], yq[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {                // unconditional loads from a clamped pixel, masked below (a predicated
        const long p = (long)prow + (long)k * PPP;   // load compiles to a branch with its own wait: K serial round trips)
        const long q = p < npix ? p : npix - 1;
        gq[k] = gz[q * C8 + cidx];
        yq[k] = y[q * C8 + cidx];
    }
    float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j]; s1[j] = s2[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool ok = (long)prow + (long)k * PPP < npix;
        float g[8], v[8];
        unpack8(gq[k], g, dt);
        unpack8(yq[k], v, dt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gp = ok ? act_grad(g[j], v[j] * sc[j] + sh[j], act) : 0.f;
            s1[j] += gp;
            s2[j] += gp * (v[j] - mu[j]) * is[j];
        }
        __builtin_amdgcn_sched_barrier(0);       // (one load's 16 unpacked values at a time: the slab itself stays packed)
    }
    // lanes of a wave that hold the same channel octet: lane bits log2(LPP) .. 5
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = LPP; m < 64; m <<= 1) {
            s1[j] += __shfl_xor(s1[j], m);
            s2[j] += __shfl_xor(s2[j], m);
        }
    }
    if (lane < LPP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][wave][lane * 8 + j] = (double)s1[j]; red[1][wave][lane * 8 + j] = (double)s2[j]; }
    }
    __syncthreads();
    if (t < CPB) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += red[0][w][t]; b += red[1][w][t]; }
        const int c = grp * CPB + t;
        const float scv = scale[c], isv = invstd[c], muv = mean[c];
        const float dg = (float)b * inv_n, db = (float)a * inv_n;
        co[0][t] = scv;
        co[1][t] = -scv * isv * dg;
        co[2][t] = scv * (isv * muv * dg - db);
        dbeta[c] = (acc ? dbeta[c] : 0.f) + (float)a;
        dgamma[c] = (acc ? dgamma[c] : 0.f) + (float)b;
    }
    __syncthreads();
    float cA[8], cB[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cA[j] = co[0][oc * 8 + j]; cB[j] = co[1][oc * 8 + j]; cC[j] = co[2][oc * 8 + j]; }
    // (the slab stays PACKED between the two passes: without this the compiler keeps the 16 unpacked floats of every
    // load alive across the reduction -- 238 registers at K = 8, scratch beyond)
#pragma unroll
    for (int k = 0; k < K; ++k)
        asm volatile("" : "+v"(gq[k].x), "+v"(gq[k].y), "+v"(gq[k].z), "+v"(gq[k].w), "+v"(yq[k].x), "+v"(yq[k].y), "+v"(yq[k].z), "+v"(yq[k].w));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long p = (long)prow + (long)k * PPP;
        float g[8], v[8], o[8];
        unpack8(gq[k], g, dt);
        unpack8(yq[k], v, dt);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = cA[j] * act_grad(g[j], v[j] * sc[j] + sh[j], act) + cB[j] * v[j] + cC[j];
        if (p < npix) gy[p * C8 + cidx] = pack8(o, dt);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CPB>
static bool bn_bwd_onepass_launch(int K, dim3 grid, hipStream_t s, const uint4 *gz, const uint4 *y, const float *scale,
                                  const float *shift, const float *mean, const float *invstd, uint4 *gy, float *dgamma,
                                  float *dbeta, long npix, int C, int act, int dt, float inv_n, int acc) {
    static int remap = -1;          // IR2RGB_BN_BWD_XCD=0: channel group = blockIdx (A/B measurements)
    if (remap < 0) { const char *e = getenv("IR2RGB_BN_BWD_XCD"); remap = e ? atoi(e) : 1; }
#define IR2RGB_ONEPASS(KK) bn_bwd_onepass_kernel<KK, CPB><<<grid, 512, 0, s>>>(gz, y, scale, shift, mean, invstd, gy, dgamma, dbeta, npix, C, act, dt, inv_n, acc, remap)
    if (K <= 4) IR2RGB_ONEPASS(4);
    else if (K <= 8) IR2RGB_ONEPASS(8);       // (K = 12 would need scratch: 256 registers are gone at K = 9)
    else return false;
#undef IR2RGB_ONEPASS
    return true;
}

extern "C" int ir2rgb_bn_bwd_blocks(long npix, int C) {
    if (npix < 1 || C < 64 || (C & (C - 1)) || C > 2048) return IR2RGB_EINVAL;
    const int R = bn_bwd_ranges(npix, C);
    const long per = (npix + R - 1) / R;
    return (int)((npix + per - 1) / per);
}

extern "C" int ir2rgb_bn_bwd(const void *gz, const void *y, const float *scale, const float *shift, const float *mean,
                             const float *invstd, void *gy, float *dgamma, float *dbeta, float *partial, long npix,
                             int C, int act, int dtype, void *stream) {
    const int R = ir2rgb_bn_bwd_blocks(npix, C);
    if (R < 0) return R;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    const bool frozen = act >= 0 && (act & 16);   // evaluation-mode BatchNorm: the mean / variance terms vanish
    const int acc = act >= 0 && (act & 32) ? 1 : 0;   // dgamma / dbeta += (a later sample group of a batched forward)
    if (act >= 0) act &= 15;
    if (!gz || !y || !gy || !dgamma || !dbeta || !partial || act < 0 || act > 2) return IR2RGB_EINVAL;
    const int R0 = bn_bwd_ranges(npix, C);
    const long per = (npix + R0 - 1) / R0;
    hipStream_t s = as_stream(stream);
    // few pixels per channel: the one-launch form (a workgroup owns CPB channels over all pixels, slabs in registers).
    // 8 channels per workgroup while that gives at most 8 loads per tensor and thread, else not applicable; wider
    // groups (better coalescing, fewer workgroups) when the layer still yields >= 128 workgroups.
    {
        static int onepass = -1;
        if (onepass < 0) { const char *e = getenv("IR2RGB_BN_BWD_ONEPASS"); onepass = e ? atoi(e) : 1; }
        if (onepass && scale && shift && mean && invstd && !frozen && npix <= 8 * 512) {
            const float inv_n = 1.0f / (float)npix;
            bool done;
            static int force_cpb = -1;      // IR2RGB_BN_BWD_ONEPASS_CPB=8|16|32: A/B measurements
            if (force_cpb < 0) { const char *e = getenv("IR2RGB_BN_BWD_ONEPASS_CPB"); force_cpb = e ? atoi(e) : 0; }
            const int min_wg = force_cpb ? 1 : 128;
            if ((!force_cpb || force_cpb == 32) && C / 32 >= min_wg && npix * 4 <= 8 * 512)
                done = bn_bwd_onepass_launch<32>((int)((npix * 4 + 511) / 512), dim3(C / 32), s, (const uint4 *)gz, (const uint4 *)y, scale, shift,
                                                 mean, invstd, (uint4 *)gy, dgamma, dbeta, npix, C, act, dtype, inv_n, acc);
            else if ((!force_cpb || force_cpb == 16) && C / 16 >= min_wg && npix * 2 <= 8 * 512)
                done = bn_bwd_onepass_launch<16>((int)((npix * 2 + 511) / 512), dim3(C / 16), s, (const uint4 *)gz, (const uint4 *)y, scale, shift,
                                                 mean, invstd, (uint4 *)gy, dgamma, dbeta, npix, C, act, dtype, inv_n, acc);
            else
                done = bn_bwd_onepass_launch<8>((int)((npix + 511) / 512), dim3(C / 8), s, (const uint4 *)gz, (const uint4 *)y, scale, shift,
                                                mean, invstd, (uint4 *)gy, dgamma, dbeta, npix, C, act, dtype, inv_n, acc);
            if (done) return ir2rgb_launch_status();
        }
    }
    // partial holds R*2*C floats followed by 3*C coefficient floats (see ir2rgb_hip.h)
    float *coef = partial + (long)R * 2 * C;
    bn_bwd_reduce_kernel<<<R * (C / 64), 512, 0, s>>>((const uint4 *)gz, (const uint4 *)y, scale, shift, mean, invstd,
                                                      partial, npix, C, act, dtype, per);
    static int fused = -1;
    if (fused < 0) { const char *e = getenv("IR2RGB_FUSED_BN_BWD"); fused = e ? atoi(e) : 1; }
    if (fused && scale && shift && mean && invstd) {
        long chunks = 2048 / (C / 64);
        const long cap = (npix + 127) / 128;
        chunks = chunks < 1 ? 1 : (chunks > cap ? cap : chunks);
        const long pp = (npix + chunks - 1) / chunks;
        dim3 grid((unsigned)(C / 64), (unsigned)((npix + pp - 1) / pp));
        bn_bwd_finalize_apply_kernel<<<grid, 256, 0, s>>>(partial, R, C, scale, shift, mean, invstd,
                                                          frozen ? 0.f : 1.0f / (float)npix, dgamma, dbeta, (const uint4 *)gz,
                                                          (const uint4 *)y, (uint4 *)gy, npix, act, dtype, pp, acc);
        return ir2rgb_launch_status();
    }
    bn_bwd_finalize_kernel<<<C / 64, 1024, 0, s>>>(partial, R, C, scale, mean, invstd, frozen ? 0.f : 1.0f / (float)npix, dgamma, dbeta,
                                                  coef, acc);
    long total8 = npix * (C / 8);
    bn_bwd_apply_kernel<<<stream_grid(total8, 256), 256, 0, s>>>((const uint4 *)gz, (const uint4 *)y, scale, shift, coef,
                                                                 (uint4 *)gy, total8, C / 8, act, dtype);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_fold_reflect(const void *dxpad, void *dx, int N, int H, int W, int C, int pad_h, int pad_w,
                                   int dtype, void *stream) {
    if (N < 0 || H < 1 || W < 1 || C < 8 || (C % 8) || pad_h < 0 || pad_w < 0 || pad_h >= H || pad_w >= W)
        return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    long total8 = (long)N * H * W * (C / 8);
    if (total8 == 0) return IR2RGB_OK;
    fold_reflect_kernel<<<stream_grid(total8, 256), 256, 0, as_stream(stream)>>>((const uint4 *)dxpad, (uint4 *)dx, H, W,
                                                                                 C / 8, pad_h, pad_w, total8, dtype);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_xexpand_bwd(const void *dxe, float *din, int N, int Cin, int H, int W, int Wout, int KW,
                                  int stride_w, int pad_w, int pad_mode, int dtype, void *stream) {
    if (N < 0 || Cin < 1 || H < 1 || W < 1 || Wout < 1 || KW < 1 || Cin * KW > 64 || stride_w < 1 || pad_w < 0)
        return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (Cin > XB_MAXC) return IR2RGB_ENOSUP;
    long total = (long)N * H * W;
    if (total == 0) return IR2RGB_OK;
    {
        static int tiled = -1;          // IR2RGB_XEXPAND_BWD_TILED=0: the one-thread-per-pixel kernel (A/B measurements)
        if (tiled < 0) { const char *e = getenv("IR2RGB_XEXPAND_BWD_TILED"); tiled = e ? atoi(e) : 1; }
        const int xtiles = (W + XB_TW - 1) / XB_TW;
        const long blocks = (long)N * H * xtiles;
        const size_t lds = (size_t)((XB_TW + KW + 2 * pad_w) / stride_w + 3) * XB_PITCH * sizeof(uint32_t);
        if (tiled && pad_w < XB_TW && blocks <= 0x7fffffffL && lds <= 64 * 1024 && !((uintptr_t)dxe & 15)) {
            xexpand_bwd_tile_kernel<<<(unsigned)blocks, 256, lds, as_stream(stream)>>>((const uint4 *)dxe, din, Cin, H, W, Wout, KW,
                                                                                    stride_w, pad_w, pad_mode, xtiles, dtype);
            return ir2rgb_launch_status();
        }
    }
    xexpand_bwd_kernel<<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>((const uint16_t *)dxe, din, Cin, H, W, Wout,
                                                                               KW, stride_w, pad_w, pad_mode, total, dtype);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_thin_grad_expand(const float *gz, void *g64, void *g8, float *dbias, int N, int Cout, int H, int W,
                                       int dtype, void *stream) {
    if (N < 0 || Cout < 1 || Cout > 8 || H < 1 || W < 1 || !gz || !g64 || !g8 || !dbias) return IR2RGB_EINVAL;
    if (dtype != IR2RGB_BF16 && dtype != IR2RGB_F16) return IR2RGB_ENOSUP;
    if (((uintptr_t)g64 | (uintptr_t)g8) & 15) return IR2RGB_EALIGN;
    const long npix = (long)N * H * W;
    if (npix == 0) return IR2RGB_OK;
    const int eb = stream_grid(npix * 8, 256);
    thin_grad_expand_kernel<<<eb + Cout, 256, 0, as_stream(stream)>>>(gz, (uint4 *)g64, (uint4 *)g8, dbias, npix,
                                                                      (long)H * W, Cout, dtype, eb);
    return ir2rgb_launch_status();
}
