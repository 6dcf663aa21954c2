/*
 * oracle/ops_ref.c -- TEST INFRASTRUCTURE ONLY (never imported by the product).
 *
 * Plain-C, single-threaded restatement of the arithmetic of the reference's three
 * FlowNet2 CUDA operators, written from a reading of the kernels (no code copied):
 *
 *   correlation  fwd : correlation_cuda_kernel.cu:46-147  (+ sizing correlation_cuda.cc:25-42)
 *   correlation  bwd : correlation_cuda_kernel.cu:150-334
 *   resample2d   fwd : resample2d_kernel.cu:15-64
 *   resample2d   bwd : resample2d_kernel.cu:67-190  (both quirks kept, see below)
 *   channelnorm  fwd : channelnorm_kernel.cu:18-60
 *   channelnorm  bwd : channelnorm_kernel.cu:63-96
 *
 * (paths relative to models/flownet2_pytorch/networks/<op>_package/ of the reference)
 *
 * Parity status: the reference ships no tests, fixtures or golden vectors for these
 * operators and its .cu files cannot be built here (no nvcc / CUDA device), so this
 * restatement is pinned by (i) an independent closed-form torch formulation
 * (oracle/closed_form.py) that must agree with it, and (ii) golden vectors frozen from
 * it under tests/golden/.  "parity unpinned by the reference's own tests".
 *
 * All tensors are dense NCHW fp32.  Accumulation is done in double so the oracle is the
 * more accurate side of any comparison.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---- correlation ------------------------------------------------------------------ */

/* Output geometry, correlation_cuda.cc:25-35. */
void oracle_correlation_out_shape(int C, int H, int W, int pad, int ksize, int max_disp,
                                  int stride1, int stride2, int *outC, int *outH, int *outW)
{
    int krad = (ksize - 1) / 2;
    int border = krad + max_disp;
    int pH = H + 2 * pad, pW = W + 2 * pad;
    int drad = max_disp / stride2;
    (void)C;
    *outC = (2 * drad + 1) * (2 * drad + 1);
    *outH = (int)ceilf((float)(pH - 2 * border) / (float)stride1);
    *outW = (int)ceilf((float)(pW - 2 * border) / (float)stride1);
}

/* zero padded channels-last copy, the reference's `channels_first` kernel (.cu:46-70) */
static float *pad_nhwc(const float *in, int N, int C, int H, int W, int pad)
{
    int pH = H + 2 * pad, pW = W + 2 * pad;
    float *r = (float *)calloc((size_t)N * pH * pW * C, sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    r[(((size_t)n * pH + (y + pad)) * pW + (x + pad)) * C + c] =
                        in[(((size_t)n * C + c) * H + y) * W + x];
    return r;
}

/* correlation_forward, .cu:73-147.  out is [N, outC, outH, outW]. */
int oracle_correlation_fwd(const float *in1, const float *in2, float *out, int N, int C, int H,
                           int W, int pad, int ksize, int max_disp, int stride1, int stride2)
{
    int outC, outH, outW;
    oracle_correlation_out_shape(C, H, W, pad, ksize, max_disp, stride1, stride2, &outC, &outH, &outW);
    int pH = H + 2 * pad, pW = W + 2 * pad;
    int krad = (ksize - 1) / 2;
    int drad = max_disp / stride2;
    int dsz = 2 * drad + 1;
    double nelems = (double)ksize * ksize * C;
    float *r1 = pad_nhwc(in1, N, C, H, W, pad);
    float *r2 = pad_nhwc(in2, N, C, H, W, pad);
    if (!r1 || !r2) return 1;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < outH; ++oy)
            for (int ox = 0; ox < outW; ++ox) {
                int y1 = oy * stride1 + max_disp, x1 = ox * stride1 + max_disp;
                for (int tj = -drad; tj <= drad; ++tj)
                    for (int ti = -drad; ti <= drad; ++ti) {
                        int y2 = y1 + tj * stride2, x2 = x1 + ti * stride2;
                        double acc = 0.0;
                        for (int j = -krad; j <= krad; ++j)
                            for (int i = -krad; i <= krad; ++i) {
                                /* For kernel_size > 1 the reference indexes the padded buffers at
                                 * y2 + j = max_disp - drad*stride2 - krad, which is negative whenever
                                 * pad <= that reach: undefined behaviour in the reference (an out-of-
                                 * bounds global read).  Positions outside the padded frame count as
                                 * zeros here, which is also what the HIP kernel does. */
                                if (y1 + j < 0 || y1 + j >= pH || x1 + i < 0 || x1 + i >= pW || y2 + j < 0 ||
                                    y2 + j >= pH || x2 + i < 0 || x2 + i >= pW)
                                    continue;
                                const float *p1 = r1 + (((size_t)n * pH + (y1 + j)) * pW + (x1 + i)) * C;
                                const float *p2 = r2 + (((size_t)n * pH + (y2 + j)) * pW + (x2 + i)) * C;
                                for (int c = 0; c < C; ++c) acc += (double)(p1[c] * p2[c]);
                            }
                        int tc = (tj + drad) * dsz + (ti + drad);
                        out[(((size_t)n * outC + tc) * outH + oy) * outW + ox] = (float)(acc / nelems);
                    }
            }
    free(r1);
    free(r2);
    return 0;
}

/* correlation_backward_input1 / _input2, .cu:150-334.  The reference launches a grid of
 * (H, W, C) blocks with y = blockIdx.x*stride1 + pad; that only tiles the input for
 * stride1 == 1 (larger strides write out of bounds), so stride1 must be 1 here. */
int oracle_correlation_bwd(const float *in1, const float *in2, const float *gout, float *gin1,
                           float *gin2, int N, int C, int H, int W, int pad, int ksize,
                           int max_disp, int stride1, int stride2)
{
    if (stride1 != 1) return 2;
    int outC, outH, outW;
    oracle_correlation_out_shape(C, H, W, pad, ksize, max_disp, stride1, stride2, &outC, &outH, &outW);
    int pH = H + 2 * pad, pW = W + 2 * pad;
    int krad = (ksize - 1) / 2;
    int drad = max_disp / stride2;
    int dsz = 2 * drad + 1;
    double nelems = (double)ksize * ksize * C;
    float *r1 = pad_nhwc(in1, N, C, H, W, pad);
    float *r2 = pad_nhwc(in2, N, C, H, W, pad);
    if (!r1 || !r2) return 1;
    memset(gin1, 0, sizeof(float) * (size_t)N * C * H * W);
    memset(gin2, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int by = 0; by < H; ++by)
                for (int bx = 0; bx < W; ++bx) {
                    int y = by * stride1 + pad, x = bx * stride1 + pad;
                    /* ---- input1 (.cu:166-240) ---- */
                    {
                        int xmin = (x - krad - max_disp) / stride1, ymin = (y - krad - max_disp) / stride1;
                        int xmax = (x + krad - max_disp) / stride1, ymax = (y + krad - max_disp) / stride1;
                        int skip = (xmax < 0 || ymax < 0 || xmin >= outW || ymin >= outH) ||
                                   (xmin > xmax || ymin > ymax);
                        if (!skip) {
                            xmin = imax(0, xmin); xmax = imin(outW - 1, xmax);
                            ymin = imax(0, ymin); ymax = imin(outH - 1, ymax);
                            double s = 0.0;
                            for (int tc = 0; tc < outC; ++tc) {
                                int i2 = (tc % dsz - drad) * stride2, j2 = (tc / dsz - drad) * stride2;
                                int yy = y + j2, xx = x + i2;
                                if (yy < 0 || yy >= pH || xx < 0 || xx >= pW) continue; /* reference would read OOB */
                                double v2 = r2[(((size_t)n * pH + yy) * pW + xx) * C + c];
                                for (int j = ymin; j <= ymax; ++j)
                                    for (int i = xmin; i <= xmax; ++i)
                                        s += (double)gout[(((size_t)n * outC + tc) * outH + j) * outW + i] * v2;
                            }
                            gin1[(((size_t)n * C + c) * H + (y - pad)) * W + (x - pad)] = (float)(s / nelems);
                        }
                    }
                    /* ---- input2 (.cu:259-332) ---- */
                    {
                        double s = 0.0;
                        for (int tc = 0; tc < outC; ++tc) {
                            int i2 = (tc % dsz - drad) * stride2, j2 = (tc / dsz - drad) * stride2;
                            int xmin = (x - krad - max_disp - i2) / stride1, ymin = (y - krad - max_disp - j2) / stride1;
                            int xmax = (x + krad - max_disp - i2) / stride1, ymax = (y + krad - max_disp - j2) / stride1;
                            if (xmax < 0 || ymax < 0 || xmin >= outW || ymin >= outH) continue;
                            if (xmin > xmax || ymin > ymax) continue;
                            xmin = imax(0, xmin); xmax = imin(outW - 1, xmax);
                            ymin = imax(0, ymin); ymax = imin(outH - 1, ymax);
                            int yy = y - j2, xx = x - i2;
                            if (yy < 0 || yy >= pH || xx < 0 || xx >= pW) continue;
                            double v1 = r1[(((size_t)n * pH + yy) * pW + xx) * C + c];
                            for (int j = ymin; j <= ymax; ++j)
                                for (int i = xmin; i <= xmax; ++i)
                                    s += (double)gout[(((size_t)n * outC + tc) * outH + j) * outW + i] * v1;
                        }
                        gin2[(((size_t)n * C + c) * H + (y - pad)) * W + (x - pad)] = (float)(s / nelems);
                    }
                }
    free(r1);
    free(r2);
    return 0;
}

/* ---- resample2d ------------------------------------------------------------------- */

/* kernel_resample2d_update_output, resample2d_kernel.cu:15-64, kernel_size == 1 (the only
 * value the reference ever passes, resample2d.py:40; larger values read out of bounds).
 * img [N,C,H,W], flow [N,2,H,W] -> out [N,C,H,W]. */
int oracle_resample2d_fwd(const float *img, const float *flow, float *out, int N, int C, int H, int W)
{
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float dx = flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    float dy = flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    float xf = (float)x + dx, yf = (float)y + dy;
                    float alpha = xf - floorf(xf), beta = yf - floorf(yf);
                    int xL = imax(imin((int)floorf(xf), W - 1), 0);
                    int xR = imax(imin((int)(floorf(xf) + 1), W - 1), 0);
                    int yT = imax(imin((int)floorf(yf), H - 1), 0);
                    int yB = imax(imin((int)(floorf(yf) + 1), H - 1), 0);
                    const float *p = img + ((size_t)n * C + c) * H * W;
                    /* the reference forms the weights in double (1. - alpha) and adds four
                     * float-rounded products into a float accumulator */
                    float v = 0.0f;
                    v += (float)((1. - alpha) * (1. - beta) * p[(size_t)yT * W + xL]);
                    v += (float)((alpha) * (1. - beta) * p[(size_t)yT * W + xR]);
                    v += (float)((1. - alpha) * (beta)*p[(size_t)yB * W + xL]);
                    v += (float)((alpha) * (beta)*p[(size_t)yB * W + xR]);
                    out[(((size_t)n * C + c) * H + y) * W + x] = v;
                }
    return 0;
}

/* kernel_resample2d_backward_input1 (.cu:67-117) and _input2 (.cu:119-190).
 * Quirks kept: (1) the image-gradient scatter weights use alpha = xf - int(xf)
 * (truncation, .cu:97-98) while the corner indices use floor; (2) the flow gradient takes
 * the branch on (c % 2): odd channel -> y-difference weighted by gamma = 1-alpha. */
int oracle_resample2d_bwd(const float *img, const float *flow, const float *gout, float *gimg,
                          float *gflow, int N, int C, int H, int W)
{
    memset(gimg, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float dx = flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    float dy = flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    float xf = (float)x + dx, yf = (float)y + dy;
                    float alpha = xf - (float)(int)xf, beta = yf - (float)(int)yf;
                    int xL = imax(imin((int)floorf(xf), W - 1), 0);
                    int xR = imax(imin((int)(floorf(xf) + 1), W - 1), 0);
                    int yT = imax(imin((int)floorf(yf), H - 1), 0);
                    int yB = imax(imin((int)(floorf(yf) + 1), H - 1), 0);
                    float g = gout[(((size_t)n * C + c) * H + y) * W + x];
                    float *q = gimg + ((size_t)n * C + c) * H * W;
                    q[(size_t)yT * W + xL] += (1 - alpha) * (1 - beta) * g;
                    q[(size_t)yT * W + xR] += (alpha) * (1 - beta) * g;
                    q[(size_t)yB * W + xL] += (1 - alpha) * (beta)*g;
                    q[(size_t)yB * W + xR] += (alpha) * (beta)*g;
                }
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < 2; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float dx = flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    float dy = flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    float xf = (float)x + dx, yf = (float)y + dy;
                    int xL = imax(imin((int)floorf(xf), W - 1), 0);
                    int xR = imax(imin((int)(floorf(xf) + 1), W - 1), 0);
                    int yT = imax(imin((int)floorf(yf), H - 1), 0);
                    int yB = imax(imin((int)(floorf(yf) + 1), H - 1), 0);
                    float o = 0.0f;
                    for (int ch = 0; ch < C; ++ch) {
                        const float *p = img + ((size_t)n * C + ch) * H * W;
                        float g = gout[(((size_t)n * C + ch) * H + y) * W + x];
                        if (c % 2) {
                            float gamma = 1 - (xf - floorf(xf));
                            o += gamma * g * p[(size_t)yB * W + xL];
                            o -= gamma * g * p[(size_t)yT * W + xL];
                            o += (1 - gamma) * g * p[(size_t)yB * W + xR];
                            o -= (1 - gamma) * g * p[(size_t)yT * W + xR];
                        } else {
                            float gamma = 1 - (yf - floorf(yf));
                            o += gamma * g * p[(size_t)yT * W + xR];
                            o -= gamma * g * p[(size_t)yT * W + xL];
                            o += (1 - gamma) * g * p[(size_t)yB * W + xR];
                            o -= (1 - gamma) * g * p[(size_t)yB * W + xL];
                        }
                    }
                    gflow[(((size_t)n * 2 + c) * H + y) * W + x] = o;
                }
    return 0;
}

/* ---- channelnorm ------------------------------------------------------------------ */

/* kernel_channelnorm_update_output, channelnorm_kernel.cu:18-60 (norm_deg is ignored by
 * the reference: it is always the L2 norm). in [N,C,H,W] -> out [N,1,H,W]. */
int oracle_channelnorm_fwd(const float *in, float *out, int N, int C, int H, int W)
{
    size_t hw = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < hw; ++p) {
            float r = 0.0f;
            for (int c = 0; c < C; ++c) {
                float v = in[((size_t)n * C + c) * hw + p];
                r += v * v;
            }
            out[(size_t)n * hw + p] = sqrtf(r);
        }
    return 0;
}

/* kernel_channelnorm_backward_input1, .cu:63-96: gin = gout * in / (out + 1e-9), the sum in
 * the denominator being formed in double (float + double literal). */
int oracle_channelnorm_bwd(const float *in, const float *out, const float *gout, float *gin, int N,
                           int C, int H, int W)
{
    size_t hw = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (size_t p = 0; p < hw; ++p) {
                size_t i = ((size_t)n * C + c) * hw + p, o = (size_t)n * hw + p;
                gin[i] = (float)((double)(gout[o] * in[i]) / ((double)out[o] + 1e-9));
            }
    return 0;
}
