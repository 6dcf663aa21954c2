// resample2d.hip -- FlowNet2 pixel-space bilinear warp (fwd/bwd) and the fused
// warp -> diff -> channel-norm step, HBM-bound kernels for gfx950.
// Semantics follow the reference's resample2d_kernel.cu:15-190 (see include/ir2rgb_hip.h).
//
// Design (vs. the reference's thread-per-(b,c,y,x) kernel that re-reads the flow C times):
// one lane owns a PIXEL: the flow pair is loaded once, corner indices and the four weights
// are computed once, then the C image planes are gathered and the C outputs stored.  Lanes
// of a wave are consecutive in x, so flow loads and output stores are fully coalesced and
// the gathers of smooth flows hit neighbouring cache lines.
// Algorithmic bytes per call = 4*N*H*W*(C in + 2 flow + C out).
#include "common.h"

struct Corner {
    int xL, xR, yT, yB;
    float alpha, beta;  // forward weights (floor based)
};

__device__ __forceinline__ Corner corners(float dx, float dy, int x, int y, int W, int H) {
    Corner k;
    float xf = (float)x + dx, yf = (float)y + dy;
    float fx = floorf(xf), fy = floorf(yf);
    k.alpha = xf - fx;
    k.beta = yf - fy;
    k.xL = max(min((int)fx, W - 1), 0);
    k.xR = max(min((int)(fx + 1.f), W - 1), 0);
    k.yT = max(min((int)fy, H - 1), 0);
    k.yB = max(min((int)(fy + 1.f), H - 1), 0);
    return k;
}

// mode bits: 1 = write warped, 2 = write diff (img1 - warped), 4 = write norm
template <int MODE>
__global__ void __launch_bounds__(256)
resample2d_fwd_kernel(const float *__restrict__ img1, const float *__restrict__ img2,
                      const float *__restrict__ flow, float *__restrict__ warped, float *__restrict__ diff,
                      float *__restrict__ norm, int C, int H, int W, long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        float dx = flow[(n * 2 + 0) * hw + p];
        float dy = flow[(n * 2 + 1) * hw + p];
        Corner k = corners(dx, dy, x, y, W, H);
        float wTL = (1.f - k.alpha) * (1.f - k.beta), wTR = k.alpha * (1.f - k.beta);
        float wBL = (1.f - k.alpha) * k.beta, wBR = k.alpha * k.beta;
        long oTL = (long)k.yT * W + k.xL, oTR = (long)k.yT * W + k.xR;
        long oBL = (long)k.yB * W + k.xL, oBR = (long)k.yB * W + k.xR;
        float nacc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float *pl = img2 + (n * C + c) * hw;
            // same association order as the reference: four products added in sequence
            float v = wTL * pl[oTL];
            v += wTR * pl[oTR];
            v += wBL * pl[oBL];
            v += wBR * pl[oBR];
            if (MODE & 1) warped[(n * C + c) * hw + p] = v;
            if (MODE & 6) {
                float d = img1[(n * C + c) * hw + p] - v;
                if (MODE & 2) diff[(n * C + c) * hw + p] = d;
                nacc += d * d;
            }
        }
        if (MODE & 4) norm[n * hw + p] = sqrtf(nacc);
    }
}

// image gradient: scatter gout to the four corners (float atomics at the memory side).
// Quirk kept from resample2d_kernel.cu:97-98: the weights use xf - int(xf) (truncation).
__global__ void __launch_bounds__(256)
resample2d_bwd_img_kernel(const float *__restrict__ flow, const float *__restrict__ gout,
                          float *__restrict__ gimg, int C, int H, int W, long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        float dx = flow[(n * 2 + 0) * hw + p];
        float dy = flow[(n * 2 + 1) * hw + p];
        Corner k = corners(dx, dy, x, y, W, H);
        float xf = (float)x + dx, yf = (float)y + dy;
        float a = xf - (float)(int)xf, b = yf - (float)(int)yf;
        long oTL = (long)k.yT * W + k.xL, oTR = (long)k.yT * W + k.xR;
        long oBL = (long)k.yB * W + k.xL, oBR = (long)k.yB * W + k.xR;
        for (int c = 0; c < C; ++c) {
            float go = gout[(n * C + c) * hw + p];
            float *q = gimg + (n * C + c) * hw;
            atomicAdd(q + oTL, (1 - a) * (1 - b) * go);
            atomicAdd(q + oTR, a * (1 - b) * go);
            atomicAdd(q + oBL, (1 - a) * b * go);
            atomicAdd(q + oBR, a * b * go);
        }
    }
}

// flow gradient, both channels by one lane (resample2d_kernel.cu:119-190): channel 0 gets the
// x-difference weighted by 1-beta/beta, channel 1 the y-difference weighted by 1-alpha/alpha.
__global__ void __launch_bounds__(256)
resample2d_bwd_flow_kernel(const float *__restrict__ img, const float *__restrict__ flow,
                           const float *__restrict__ gout, float *__restrict__ gflow, int C, int H, int W,
                           long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        float dx = flow[(n * 2 + 0) * hw + p];
        float dy = flow[(n * 2 + 1) * hw + p];
        Corner k = corners(dx, dy, x, y, W, H);
        float gx = 1.f - k.beta;   // "gamma" of the even (dx) channel
        float gy = 1.f - k.alpha;  // "gamma" of the odd  (dy) channel
        long oTL = (long)k.yT * W + k.xL, oTR = (long)k.yT * W + k.xR;
        long oBL = (long)k.yB * W + k.xL, oBR = (long)k.yB * W + k.xR;
        float ox = 0.f, oy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float *pl = img + (n * C + c) * hw;
            float go = gout[(n * C + c) * hw + p];
            float tl = pl[oTL], tr = pl[oTR], bl = pl[oBL], br = pl[oBR];
            ox += gx * go * tr;
            ox -= gx * go * tl;
            ox += (1 - gx) * go * br;
            ox -= (1 - gx) * go * bl;
            oy += gy * go * bl;
            oy -= gy * go * tl;
            oy += (1 - gy) * go * br;
            oy -= (1 - gy) * go * tr;
        }
        gflow[(n * 2 + 0) * hw + p] = ox;
        gflow[(n * 2 + 1) * hw + p] = oy;
    }
}

extern "C" int ir2rgb_resample2d_fwd(const float *img, const float *flow, float *out, int N, int C, int H,
                                     int W, int kernel_size, void *stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return IR2RGB_EINVAL;
    if (kernel_size != 1) return IR2RGB_ENOSUP;
    long total = (long)N * H * W;
    if (total == 0 || C == 0) return IR2RGB_OK;
    resample2d_fwd_kernel<1><<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>(
        nullptr, img, flow, out, nullptr, nullptr, C, H, W, total);
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_warp_diff_norm_fwd(const float *img1, const float *img2, const float *flow,
                                         float *warped, float *diff, float *norm, int N, int C, int H, int W,
                                         void *stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return IR2RGB_EINVAL;
    long total = (long)N * H * W;
    if (total == 0) return IR2RGB_OK;
    int mode = (warped ? 1 : 0) | (diff ? 2 : 0) | (norm ? 4 : 0);
    int grid = stream_grid(total, 256);
    hipStream_t s = as_stream(stream);
#define LAUNCH(M)                                                                                          \
    case M:                                                                                                \
        resample2d_fwd_kernel<M><<<grid, 256, 0, s>>>(img1, img2, flow, warped, diff, norm, C, H, W, total); \
        break;
    switch (mode) {
        LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7)
        default: return IR2RGB_EINVAL;
    }
#undef LAUNCH
    return ir2rgb_launch_status();
}

extern "C" int ir2rgb_resample2d_bwd(const float *img, const float *flow, const float *gout, float *gimg,
                                     float *gflow, int N, int C, int H, int W, int kernel_size, void *stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return IR2RGB_EINVAL;
    if (kernel_size != 1) return IR2RGB_ENOSUP;
    long total = (long)N * H * W;
    if (total == 0 || C == 0) return IR2RGB_OK;
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(gimg, 0, sizeof(float) * (size_t)total * C, s);
    if (e != hipSuccess) return (int)e;
    int grid = stream_grid(total, 256);
    resample2d_bwd_img_kernel<<<grid, 256, 0, s>>>(flow, gout, gimg, C, H, W, total);
    resample2d_bwd_flow_kernel<<<grid, 256, 0, s>>>(img, flow, gout, gflow, C, H, W, total);
    return ir2rgb_launch_status();
}
