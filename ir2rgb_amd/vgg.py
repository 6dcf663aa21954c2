"""VGG19 perceptual loss on the MFMA convolution kernels.

Reference: ``Vgg19`` (models/networks.py:721-752: torchvision ``vgg19().features[0:30]`` cut into five
slices ending at relu1_1, relu2_1, relu3_1, relu4_1, relu5_1) and ``VGGLoss`` (models/loss.py:44-59:
``sum_i w_i * L1(vgg(x)_i, vgg(y)_i.detach())`` with w = 1/32, 1/16, 1/8, 1/4, 1 after 2x average-pool
down-sampling while the width exceeds 1024).

The container mirrors the reference's module tree (``slice{1..5}.{torchvision index}``), so a torchvision
``vgg19`` ``features`` state_dict maps onto it by index and the reference's own checkpoints load.  There is
no network here: the pretrained weights cannot be downloaded, so the loop keeps ``no_vgg`` (the reference's
option, discriminator.py:65-66) and the tests pin this module against the same architecture evaluated by
plain torch with the same randomly initialised weights -- the loss VALUE is unpinned by any reference
fixture.  Every 3x3 convolution (bias + ReLU fused in the epilogue) runs on the HIP kernels, forward and
backward; the four 2x2 max-pools stay torch operators on channels_last half tensors.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as A
from . import conv as C
from .losses import fused_losses

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]   # features[0:30]
_SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]


def _features():
    layers, cin = [], 3
    for v in _CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers


class Vgg19(nn.Module):
    """forward(x [N,3,H,W] fp32) -> [relu1_1, relu2_1, relu3_1, relu4_1, relu5_1] (channels_last half)."""
    compute_dtype = torch.bfloat16

    def __init__(self, requires_grad=False):
        super().__init__()
        feats = _features()
        assert len(feats) == 30
        for s, (a, b) in enumerate(_SLICES, 1):
            seq = nn.Sequential()
            for i in range(a, b):
                seq.add_module(str(i), feats[i])
            setattr(self, f"slice{s}", seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        if not x.is_cuda:
            raise ValueError("Vgg19: GPU tensors only (no CPU fallback)")
        dt = self.compute_dtype
        outs, h, first = [], x, True
        for s in range(1, 6):
            for m in getattr(self, f"slice{s}"):
                if isinstance(m, nn.Conv2d):
                    h = A.conv_stage(h, m, None, 0, C.PAD_ZERO, dt, first=first, fused_relu=True)
                    first = False
                elif isinstance(m, nn.MaxPool2d):
                    h = F.max_pool2d(h, 2, 2)
                # nn.ReLU: fused into the convolution above
            outs.append(h)
        return outs


class VGGLoss(nn.Module):
    """models/loss.py:44-59."""

    def __init__(self, vgg=None):
        super().__init__()
        self.vgg = vgg if vgg is not None else Vgg19()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        while x.size(3) > 1024:
            x, y = F.avg_pool2d(x, 2, 2, count_include_pad=False), F.avg_pool2d(y, 2, 2, count_include_pad=False)
        fx = self.vgg(x)
        with torch.no_grad():
            fy = self.vgg(y)
        terms = [("l1", a, b, w, 0) for a, b, w in zip(fx, fy, self.weights)]
        return fused_losses(terms, 1, self.vgg.compute_dtype)[0]
