"""FlowNet2 graph (the caller of the three hand-written operators) for the reference-flow path.

Mirrors the module tree of reference models/flownet2_pytorch/models.py:30-161 and
networks/{FlowNetC,FlowNetS,FlowNetSD,FlowNetFusion,submodules}.py so that the published
``FlowNet2_checkpoint.pth.tar`` state_dict loads unchanged (same attribute names, batchNorm=False
variant: ``conv*.0`` = Conv2d, ``deconv*.0`` = ConvTranspose2d).  Only ``FlowNet2`` is built (the
C/S/SD/CS/CSS variants are unused by ir2rgb).

Scope: the convolution stacks here are ordinary torch convolutions on
the GPU (0.53 TFLOP per pair, frozen, no_grad); what is hand-written is what the reference
hand-wrote -- Correlation, Resample2d, ChannelNorm -- plus the fused warp->diff->norm step
(ir2rgb_warp_diff_norm_fwd) that replaces three launches at models.py:109-111, :121-123.
``conv_dtype`` selects the torch convolution precision (bf16 channels_last by default, matching
the reference's optional --fp16 path in spirit; the three operators always run in fp32).
"""
import torch
import torch.nn as nn
from torch.nn import init

from ..ext import warp_diff_norm
from .networks.channelnorm_package.channelnorm import ChannelNorm
from .networks.correlation_package.correlation import Correlation
from .networks.resample2d_package.resample2d import Resample2d


def conv(in_planes, out_planes, kernel_size=3, stride=1):
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


def i_conv(in_planes, out_planes):
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, 3, 1, 1, bias=True))


def predict_flow(in_planes):
    return nn.Conv2d(in_planes, 2, 3, 1, 1, bias=True)


def deconv(in_planes, out_planes):
    return nn.Sequential(nn.ConvTranspose2d(in_planes, out_planes, 4, 2, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))


def _flownet_init(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                init.uniform_(m.bias)
            init.xavier_uniform_(m.weight)


class _Refiner(nn.Module):
    """Shared coarse-to-fine decoder of FlowNetC / FlowNetS: levels 6 -> 2."""

    def _build_decoder(self, up_bias):
        self.deconv5, self.deconv4, self.deconv3, self.deconv2 = deconv(1024, 512), deconv(1026, 256), deconv(770, 128), deconv(386, 64)
        for lvl, ch in ((6, 1024), (5, 1026), (4, 770), (3, 386), (2, 194)):
            setattr(self, f"predict_flow{lvl}", predict_flow(ch))
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, f"upsampled_flow{a}_to_{b}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=up_bias))

    def _decode(self, c6, c5, c4, c3, c2):
        feat, skips = c6, {5: c5, 4: c4, 3: c3, 2: c2}
        for lvl in (6, 5, 4, 3):
            flow = getattr(self, f"predict_flow{lvl}")(feat)
            up = getattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}")(flow)
            feat = torch.cat((skips[lvl - 1], getattr(self, f"deconv{lvl - 1}")(feat), up), 1)
        return self.predict_flow2(feat)


class FlowNetC(_Refiner):
    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = conv(3, 64, 7, 2), conv(64, 128, 5, 2), conv(128, 256, 5, 2)
        self.conv_redir = conv(256, 32, 1, 1)
        self.corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = nn.LeakyReLU(0.1, inplace=True)
        self.conv3_1 = conv(473, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        self._build_decoder(up_bias=True)
        _flownet_init(self)

    def forward(self, x):
        c2a = self.conv2(self.conv1(x[:, 0:3]))  # also the level-2 skip connection
        a3 = self.conv3(c2a)
        b3 = self.conv3(self.conv2(self.conv1(x[:, 3:])))
        cost = self.corr_activation(self.corr(a3.float().contiguous(), b3.float().contiguous()).to(a3.dtype))
        c3 = self.conv3_1(torch.cat((self.conv_redir(a3), cost), 1))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._decode(c6, c5, c4, c3, c2a)


class FlowNetS(_Refiner):
    def __init__(self, input_channels=12):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = conv(input_channels, 64, 7, 2), conv(64, 128, 5, 2), conv(128, 256, 5, 2)
        self.conv3_1 = conv(256, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        self._build_decoder(up_bias=False)
        _flownet_init(self)

    def forward(self, x):
        c2 = self.conv2(self.conv1(x))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._decode(c6, c5, c4, c3, c2)


class FlowNetSD(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = conv(6, 64)
        self.conv1, self.conv1_1 = conv(64, 64, stride=2), conv(64, 128)
        self.conv2, self.conv2_1 = conv(128, 128, stride=2), conv(128, 128)
        self.conv3, self.conv3_1 = conv(128, 256, stride=2), conv(256, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        self.deconv5, self.deconv4, self.deconv3, self.deconv2 = deconv(1024, 512), deconv(1026, 256), deconv(770, 128), deconv(386, 64)
        self.inter_conv5, self.inter_conv4 = i_conv(1026, 512), i_conv(770, 256)
        self.inter_conv3, self.inter_conv2 = i_conv(386, 128), i_conv(194, 64)
        for lvl, ch in ((6, 1024), (5, 512), (4, 256), (3, 128), (2, 64)):
            setattr(self, f"predict_flow{lvl}", predict_flow(ch))
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, f"upsampled_flow{a}_to_{b}", nn.ConvTranspose2d(2, 2, 4, 2, 1))
        _flownet_init(self)

    def forward(self, x):
        c1 = self.conv1_1(self.conv1(self.conv0(x)))
        c2 = self.conv2_1(self.conv2(c1))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        feat, flow_in, skips = c6, c6, {5: c5, 4: c4, 3: c3, 2: c2}
        for lvl in (6, 5, 4, 3):
            flow = getattr(self, f"predict_flow{lvl}")(flow_in)
            up = getattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}")(flow)
            feat = torch.cat((skips[lvl - 1], getattr(self, f"deconv{lvl - 1}")(feat), up), 1)
            flow_in = getattr(self, f"inter_conv{lvl - 1}")(feat)
        return self.predict_flow2(flow_in)


class FlowNetFusion(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = conv(11, 64)
        self.conv1, self.conv1_1 = conv(64, 64, stride=2), conv(64, 128)
        self.conv2, self.conv2_1 = conv(128, 128, stride=2), conv(128, 128)
        self.deconv1, self.deconv0 = deconv(128, 32), deconv(162, 16)
        self.inter_conv1, self.inter_conv0 = i_conv(162, 32), i_conv(82, 16)
        self.predict_flow2, self.predict_flow1, self.predict_flow0 = predict_flow(128), predict_flow(32), predict_flow(16)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        _flownet_init(self)

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        up2 = self.upsampled_flow2_to_1(self.predict_flow2(c2))
        cat1 = torch.cat((c1, self.deconv1(c2), up2), 1)
        up1 = self.upsampled_flow1_to_0(self.predict_flow1(self.inter_conv1(cat1)))
        cat0 = torch.cat((c0, self.deconv0(cat1), up1), 1)
        return self.predict_flow0(self.inter_conv0(cat0))


class FlowNet2(nn.Module):
    """inputs [B,3,2,H,W] (H, W multiples of 64) -> flow [B,2,H,W] fp32 (reference models.py:96-161)."""

    def __init__(self, args=None, batchNorm=False, div_flow=20.0, fp16=False, conv_dtype=torch.bfloat16):
        super().__init__()
        if batchNorm:
            raise NotImplementedError("ir2rgb uses the batchNorm=False FlowNet2")
        self.div_flow, self.rgb_max, self.conv_dtype = div_flow, 1.0, conv_dtype
        self.channelnorm = ChannelNorm()
        self.flownetc = FlowNetC()
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.flownets_1 = FlowNetS()
        self.upsample2 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.upsample3 = nn.Upsample(scale_factor=4, mode="nearest")
        self.upsample4 = nn.Upsample(scale_factor=4, mode="nearest")
        self.resample = Resample2d()
        self.flownetfusion = FlowNetFusion()
        _flownet_init(self)

    use_hip_convs = True  # False: torch convolutions (kept as the GPU reference for tests/test_flownet2_gpu.py)

    def _net(self, net, x):
        """One sub-network.  Product path: the MFMA engine of ir2rgb_amd/flownet2_hip.py; test reference:
        the torch convolution stack in conv_dtype / channels_last.  Result in fp32 either way."""
        if self.use_hip_convs:
            from .. import flownet2_hip as FH
            run = {FlowNetC: FH.flownetc, FlowNetS: FH.flownets, FlowNetSD: FH.flownetsd, FlowNetFusion: FH.flownetfusion}
            dt = torch.bfloat16 if self.conv_dtype == torch.float32 else self.conv_dtype
            return run[type(net)](net, x.float().contiguous(), dt)
        if self.conv_dtype == torch.float32:
            return net(x)
        with torch.autocast("cuda", dtype=self.conv_dtype):
            return net(x.contiguous(memory_format=torch.channels_last)).float()

    def forward(self, inputs):
        rgb_mean = inputs.contiguous().view(inputs.size()[:2] + (-1,)).mean(dim=-1).view(inputs.size()[:2] + (1, 1, 1))
        x = (inputs - rgb_mean) / self.rgb_max
        x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
        im0, im1 = x[:, :3].contiguous(), x[:, 3:].contiguous()

        flow_c = self.upsample1(self._net(self.flownetc, x) * self.div_flow).contiguous()
        warped, _, norm = warp_diff_norm(im0, im1, flow_c, want_diff=False)        # models.py:109-111 fused
        flow_s1 = self._net(self.flownets_1, torch.cat((x, warped, flow_c / self.div_flow, norm), 1))
        flow_s1 = self.upsample2(flow_s1 * self.div_flow).contiguous()
        warped, _, norm = warp_diff_norm(im0, im1, flow_s1, want_diff=False)       # :121-123 fused
        flow_s2 = self._net(self.flownets_2, torch.cat((x, warped, flow_s1 / self.div_flow, norm), 1))
        flow_s2 = self.upsample4(flow_s2 * self.div_flow).contiguous()
        norm_s2 = self.channelnorm(flow_s2)                                        # :131
        _, _, diff_s2 = warp_diff_norm(im0, im1, flow_s2, want_warped=False, want_diff=False)   # :133-137

        flow_sd = self.upsample3(self._net(self.flownets_d, x) / self.div_flow).contiguous()
        norm_sd = self.channelnorm(flow_sd)                                        # :144
        _, _, diff_sd = warp_diff_norm(im0, im1, flow_sd, want_warped=False, want_diff=False)   # :146-150

        fused_in = torch.cat((im0, flow_sd, flow_s2, norm_sd, norm_s2, diff_sd, diff_s2), 1)
        return self._net(self.flownetfusion, fused_in)
