"""Independent closed-form (torch, CPU) statements of the three FlowNet2 operators --
TEST INFRASTRUCTURE ONLY.  They pin oracle/ops_ref.c from a second direction:

* correlation (k=1)  = mean over channels of f1 * shift(zero_pad(f2), stride2*tj, stride2*ti)
                       (meaning of correlation_cuda_kernel.cu:73-147)
* resample2d         = bilinear gather at (x+dx, y+dy) with corner *indices* clamped, which
                       equals grid_sample(align_corners=True, padding_mode='border') in pixel
                       coordinates (resample2d_kernel.cu:15-64)
* channelnorm        = sqrt(sum_c x^2) (channelnorm_kernel.cu:18-60)
* backward passes of correlation / channelnorm = autograd of the closed forms
  (the resample2d backward has reference quirks and is pinned by ops_ref.c alone)
"""
import torch
import torch.nn.functional as F


def correlation(f1, f2, pad_size, kernel_size, max_displacement, stride1, stride2):
    assert kernel_size == 1 and stride1 == 1 and pad_size == max_displacement
    N, C, H, W = f1.shape
    d = max_displacement // stride2
    f2p = F.pad(f2, (pad_size,) * 4)
    outs = []
    for tj in range(-d, d + 1):
        for ti in range(-d, d + 1):
            y0 = pad_size + tj * stride2
            x0 = pad_size + ti * stride2
            outs.append((f1 * f2p[:, :, y0:y0 + H, x0:x0 + W]).mean(1, keepdim=True))
    return torch.cat(outs, 1)


def resample2d(img, flow):
    N, C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=img.dtype), torch.arange(W, dtype=img.dtype), indexing="ij")
    xf = xs[None] + flow[:, 0]
    yf = ys[None] + flow[:, 1]
    gx = 2 * xf / max(W - 1, 1) - 1
    gy = 2 * yf / max(H - 1, 1) - 1
    grid = torch.stack([gx, gy], -1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)


def channelnorm(x):
    return x.pow(2).sum(1, keepdim=True).sqrt()
