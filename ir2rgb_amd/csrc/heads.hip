// heads.hip -- generator output heads (gfx950, HBM-bound, fp32):
//   * head_finish : vertical part of the separable 7x7 head convolutions + bias + tanh / *20 /
//     sigmoid, writing the reference's NCHW fp32 outputs (networks.py:166, :170-171, :200-201);
//   * warp_blend  : grid_sample warp of the previous frame by the predicted flow and the
//     soft-mask blend (networks.py:89-100, :207-209), one lane per pixel.
#include "common.h"

__global__ void __launch_bounds__(256)
head_finish_kernel(const float *__restrict__ T, const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                   int Cout, int KH, int CT, int pad, unsigned acts, float mul, long total) {
    const long hw = (long)H * W;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        long n = g / hw, p = g - n * hw;
        int y = (int)(p / W), x = (int)(p - (long)y * W);
        for (int co = 0; co < Cout; ++co) {
            float acc = bias ? bias[co] : 0.f;
            for (int ky = 0; ky < KH; ++ky) {
                int iy = y + ky - pad;
                iy = iy < 0 ? -iy : iy;
                iy = iy >= H ? 2 * H - 2 - iy : iy;
                acc += T[((n * H + iy) * (long)W + x) * CT + co * KH + ky];
            }
            unsigned a = (acts >> (4 * co)) & 15u;
            float v = a == 1 ? tanhf(acc) : (a == 2 ? 1.f / (1.f + expf(-acc)) : acc * mul);
            out[(n * Cout + co) * hw + p] = v;
        }
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

extern "C" int ir2rgb_head_finish(const float *T, const float *bias, float *out, int N, int H, int W, int Cout, int KH,
                                  int CT, int pad_h, unsigned acts, float mul, void *stream) {
    if (N < 0 || H < 1 || W < 1 || Cout < 1 || Cout > 8 || KH < 1 || CT < Cout * KH || pad_h < 0 || pad_h >= H)
        return IR2RGB_EINVAL;
    long total = (long)N * H * W;
    if (total == 0) return IR2RGB_OK;
    head_finish_kernel<<<stream_grid(total, 256), 256, 0, as_stream(stream)>>>(T, bias, out, H, W, Cout, KH, CT, pad_h,
                                                                               acts, mul, total);
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
