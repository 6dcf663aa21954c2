"""Rounding-emulating restatement of the generator / discriminator passes -- TEST INFRASTRUCTURE ONLY.

oracle/networks_oracle.py is the reference's graph in plain fp32 (pinned by the reference goldens).  The HIP
path keeps activations in half precision, so against that fp32 oracle a whole-network comparison can only be
held to a rounding-noise bound (ReLU masks and L1 signs flip, BatchNorm over 32 pixels amplifies).  This file
evaluates THE SAME GRAPH on the same parameter containers with fp32 torch operators, but rounds to the compute
dtype exactly where the HIP path stores a half tensor -- so what is left between the two is summation order, and
whole-network forward AND backward comparisons can be held to a bound that catches a missing or mis-scaled term.

Derivatives come from torch.autograd (not from restated backward formulas); only the rounding points of the
gradients are placed by hand (``_RoundFB`` rounds the gradient that passes through it, as a half tensor would):

    where the HIP path rounds                                           here
    -------------------------------------------------------------------------------------------------------
    packed MFMA weights (half copy of the fp32 master)                   _round_ste(W): gradient stays fp32
    convolution output y (statistics taken from the fp32 accumulators)   stats of y32, then _RoundFB(y32)
    stage output z = act(bn(y)) + residuals                              _RoundFB
    gradient w.r.t. y and w.r.t. the stage input (half tensors)          the backward half of the two above
    first-layer image operand (x-im2col to half)                         _RoundFB on the image
    fp32 outputs (PatchGAN logits, head responses): gradient -> half     _RoundB (identity forward)
    feature-matching L1: gradient = sign(a-b) * half(g*w/n)              l1_half

The bias of a convolution in front of BatchNorm is left out (it cancels; ir2rgb_bn_finalize_ex).  Graph citations
as in networks_oracle.py: reference models/networks.py:191-220, :288-317, :584-586, :656-668, :89-100.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _RoundFB(torch.autograd.Function):
    """Half-precision storage of a tensor: value rounded forward, gradient rounded backward."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


class _RoundSTE(torch.autograd.Function):
    """Half copy of an fp32 master parameter: rounded forward, gradient passed through in fp32."""

    @staticmethod
    def forward(ctx, x, dt):
        return x.to(dt).float()

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundB(torch.autograd.Function):
    """An fp32 tensor whose gradient is handed on in half precision."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


def rfb(x, dt):
    return _RoundFB.apply(x, dt)


def rste(x, dt):
    return _RoundSTE.apply(x, dt)


def rb(x, dt):
    return _RoundB.apply(x, dt)


def _is_resblock(m):
    return hasattr(m, "conv_block") and isinstance(m.conv_block, nn.Sequential)


def _act(x, a):
    if isinstance(a, nn.ReLU):
        return F.relu(x)
    if isinstance(a, nn.LeakyReLU):
        return F.leaky_relu(x, a.negative_slope)
    raise RuntimeError(f"unexpected activation {type(a).__name__}")


def _bn(y32, bn, dt, training):
    """BatchNorm as the HIP path applies it: statistics of the fp32 accumulators, normalisation of the rounded y."""
    if training or not bn.track_running_stats:
        mean = y32.mean((0, 2, 3), keepdim=True)
        var = y32.var((0, 2, 3), unbiased=False, keepdim=True)
        shift_bias = 0.0
    else:   # evaluation mode: running statistics of the BIASED convolution output (handled by the caller)
        raise RuntimeError("evaluation-mode emulation is not needed: oracle/networks_oracle.py covers it")
    y = rfb(y32, dt)
    return (y - mean) * torch.rsqrt(var + bn.eps) * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1) + shift_bias


def conv_stage(x, conv, bn, act, dt, pad_reflect=0, res=(), training=True):
    """One [pad, conv, norm, activation] run (+ residual adds) with the HIP path's rounding points.  ``x`` already
    holds half-representable values."""
    if pad_reflect:
        x = F.pad(x, (pad_reflect,) * 4, mode="reflect")
    w = rste(conv.weight, dt)
    if isinstance(conv, nn.ConvTranspose2d):
        y32 = F.conv_transpose2d(x, w, None, conv.stride, conv.padding, conv.output_padding)
    else:
        y32 = F.conv2d(x, w, None, conv.stride, conv.padding if not pad_reflect else 0)
    if bn is not None:
        z = _bn(y32, bn, dt, training)
        if act is not None:
            z = _act(z, act)
        for r in res:
            if r is not None:
                z = z + r
        return rfb(z, dt)
    y32 = y32 + conv.bias.view(1, -1, 1, 1)
    if act is not None:
        y32 = _act(y32, act)
    return rfb(y32, dt)


def run_seq(seq, x, dt, training=True, final_residual=None, image_input=False):
    """An nn.Sequential of the generator (reference module order) on a half-valued tensor."""
    mods = list(seq)
    last_block = max((j for j, m in enumerate(mods) if _is_resblock(m)), default=-1)
    i = 0
    if image_input:
        x = rfb(x, dt)
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReflectionPad2d):
            x = conv_stage(x, mods[i + 1], mods[i + 2], mods[i + 3], dt, pad_reflect=m.padding[0], training=training)
            i += 4
        elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            x = conv_stage(x, m, mods[i + 1], mods[i + 2], dt, training=training)
            i += 3
        elif _is_resblock(m):
            cb = m.conv_block
            h = conv_stage(x, cb[1], cb[2], cb[3], dt, pad_reflect=1, training=training)
            x = conv_stage(h, cb[5], cb[6], None, dt, pad_reflect=1, res=(x, final_residual if i == last_block else None),
                           training=training)
            i += 1
        else:
            raise RuntimeError(f"unexpected module {type(m).__name__}")
    return x


def heads(feat, convs, acts, mul, dt):
    """ReflectionPad2d(3) + Conv7x7 heads evaluated as the HIP path does: a 1x7 pass giving Cout*7 fp32 row responses
    (whose gradient is stored in half), then the 7-tap vertical gather with bias and output non-linearity in fp32."""
    w = torch.cat([rste(c.weight, dt) for c in convs], 0)                       # [co, ci, 7, 7]
    b = torch.cat([c.bias for c in convs], 0)
    co, ci, kh, kw = w.shape
    n, _, h, wd = feat.shape
    t = F.conv2d(F.pad(feat, (kw // 2, kw // 2, 0, 0), mode="reflect"), w.permute(0, 2, 1, 3).reshape(co * kh, ci, 1, kw))
    t = rb(t, dt)
    tp = F.pad(t, (0, 0, kh // 2, kh // 2), mode="reflect").view(n, co, kh, h + kh - 1, wd)
    out = sum(tp[:, :, ky, ky:ky + h] for ky in range(kh)) + b.view(1, -1, 1, 1)
    cols = []
    for c in range(co):
        v = out[:, c:c + 1]
        cols.append(v * mul if acts[c] == 0 else (torch.tanh(v) if acts[c] == 1 else torch.sigmoid(v)))
    return torch.cat(cols, 1)


def resample(image, flow):
    """networks.py:93-100 (default align_corners=False of F.grid_sample kept on purpose), fp32."""
    b, c, h, w = image.shape
    xs = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    grid = torch.cat([xs, ys], 1).to(flow)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    return F.grid_sample(image, (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=False)


def generator_forward(g, input, img_prev, img_feat_coarse=None, flow_feat_coarse=None, use_raw_only=False,
                      dtype=torch.bfloat16):
    """The reference's 7-tuple for a Composite(Local)GeneratorModule parameter container, train-mode BatchNorm."""
    dt = dtype
    local = not hasattr(g, "model_res_img")
    seg = run_seq(g.model_down_seg, input, dt, image_input=True)
    if local:
        # the second encoder's last stage adds the first encoder's output in its epilogue pass; then the coarse
        # features are added by a separate (rounded) add per branch (ir2rgb_amd.networks: _encode, A.add)
        mods = list(g.model_down_img)
        h = conv_stage(rfb(img_prev, dt), mods[1], mods[2], mods[3], dt, pad_reflect=3)
        down = conv_stage(h, mods[4], mods[5], mods[6], dt, res=(seg,))
        img_feat = run_seq(g.model_up_img, rfb(down + rfb(img_feat_coarse, dt), dt), dt)
        mult = 20.0 * (2 ** g.scale)
    else:
        down = run_seq(g.model_down_img, img_prev, dt, final_residual=seg, image_input=True)
        img_feat = run_seq(g.model_up_img, run_seq(g.model_res_img, down, dt), dt)
        mult = 20.0
    img_raw = heads(img_feat, [g.model_final_img[1]], [1] * g.model_final_img[1].out_channels, 1.0, dt)
    flow = weight = flow_feat = None
    if not g.no_flow:
        if local:
            flow_feat = run_seq(g.model_up_flow, rfb(down + rfb(flow_feat_coarse, dt), dt), dt)
        else:
            flow_feat = run_seq(g.model_up_flow, run_seq(g.model_res_flow, down, dt), dt)
        fw = heads(flow_feat, [g.model_final_flow[1], g.model_final_w[1]], [0, 0, 2], mult, dt)
        flow, weight = fw[:, 0:2], fw[:, 2:3]
    if use_raw_only or g.no_flow:
        img_final = img_raw
    else:
        warp = resample(img_prev[:, -3:], flow)
        img_final = img_raw * weight + warp * (1 - weight)
    return img_final, flow, weight, img_raw, img_feat, flow_feat, None


def discriminator_forward(d, x, dtype=torch.bfloat16):
    """MultiScaleDiscriminator (networks.py:656-668): list[num_D] of the five intermediates, logits in fp32."""
    dt = dtype
    result = []
    for i in range(d.num_D):
        idx = d.num_D - 1 - i
        if d.getIntermFeat:
            groups = [getattr(d, f"scale{idx}_layer{j}") for j in range(d.n_layers + 2)]
        else:
            raise RuntimeError("emulation covers getIntermFeat=True (the training configuration)")
        h = conv_stage(rfb(x, dt), groups[0][0], None, groups[0][1], dt)
        outs = [h]
        for grp in groups[1:-1]:
            h = conv_stage(h, grp[0], grp[1], grp[2], dt)
            outs.append(h)
        last = groups[-1][0]
        logits = F.conv2d(h, rste(last.weight, dt), last.bias, last.stride, last.padding)
        outs.append(rb(logits, dt))
        result.append(outs)
        if i != d.num_D - 1:
            x = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    return result


class _L1Half(torch.autograd.Function):
    """weight * mean|a - b| on half-valued features; gradient sign(a-b) * half(g*weight/n) (losses.hip, term 'l1')."""

    @staticmethod
    def forward(ctx, a, b, weight, dt):
        ctx.save_for_backward(a, b)
        ctx.weight, ctx.dt = weight, dt
        return (a - b).abs().mean() * weight

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        s = (g.float() * (torch.tensor(ctx.weight, dtype=torch.float32) / torch.tensor(float(a.numel()), dtype=torch.float32)))
        return torch.sign(a - b) * s.to(ctx.dt).float(), None, None, None


def l1_half(a, b, weight, dtype):
    return _L1Half.apply(a, b.detach(), float(weight), dtype)
