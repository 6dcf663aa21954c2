"""Plain-torch fp32 restatement of the generator / discriminator forward passes --
TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` "port" that bench.py times on the host cores).

It evaluates the SAME parameter containers the product modules hold (nn.Conv2d, nn.BatchNorm2d,
nn.ReflectionPad2d ... in the reference's module order) with ordinary torch operators, i.e. the
computation graph of reference models/networks.py:

    CompositeGeneratorModule.forward        networks.py:191-220
    CompositeLocalGeneratorModule.forward   networks.py:288-317
    ResnetBlock.forward                     networks.py:584-586
    MultiScaleDiscriminator.forward         networks.py:656-668
    resample / grid_sample (a7)             networks.py:89-100

Pinned by tests/golden/net_*.npz, which were produced by the reference's own module code
(tests/golden/make_net_goldens.py): tests/test_oracle_networks.py requires agreement to 1e-5.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _is_resblock(m):
    return hasattr(m, "conv_block") and isinstance(m.conv_block, nn.Sequential)


def run_seq(seq, x):
    for m in seq:
        x = x + m.conv_block(x) if _is_resblock(m) else m(x)
    return x


def resample(image, flow):
    """networks.py:93-100 with the default align_corners=False of F.grid_sample kept on purpose."""
    b, c, h, w = image.shape
    xs = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    grid = torch.cat([xs, ys], 1).to(flow)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    return F.grid_sample(image, (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border",
                         align_corners=False)


def generator_forward(g, input, img_prev, img_feat_coarse=None, flow_feat_coarse=None, use_raw_only=False):
    """Returns the reference's 7-tuple for a Composite(Local)GeneratorModule parameter container."""
    local = not hasattr(g, "model_res_img")
    if local:
        down = run_seq(g.model_down_seg, input) + run_seq(g.model_down_img, img_prev)
        img_feat = run_seq(g.model_up_img, down + img_feat_coarse)
        mult = 20 * (2 ** g.scale)
    else:
        down = run_seq(g.model_down_seg, input) + run_seq(g.model_down_img, img_prev)
        img_feat = run_seq(g.model_up_img, run_seq(g.model_res_img, down))
        mult = 20
    img_raw = run_seq(g.model_final_img, img_feat)
    flow = weight = flow_feat = None
    if not g.no_flow:
        if local:
            flow_feat = run_seq(g.model_up_flow, down + flow_feat_coarse)
        else:
            flow_feat = run_seq(g.model_up_flow, run_seq(g.model_res_flow, down))
        flow = run_seq(g.model_final_flow, flow_feat) * mult
        weight = run_seq(g.model_final_w, flow_feat)
    if use_raw_only or g.no_flow:
        img_final = img_raw
    else:
        warp = resample(img_prev[:, -3:], flow)
        img_final = img_raw * weight + warp * (1 - weight)
    return img_final, flow, weight, img_raw, img_feat, flow_feat, None


def discriminator_forward(d, x):
    result = []
    for i in range(d.num_D):
        idx = d.num_D - 1 - i
        if d.getIntermFeat:
            outs, h = [], x
            for j in range(d.n_layers + 2):
                h = getattr(d, f"scale{idx}_layer{j}")(h)
                outs.append(h)
            result.append(outs)
        else:
            result.append([getattr(d, f"layer{idx}")(x)])
        if i != d.num_D - 1:
            x = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    return result
