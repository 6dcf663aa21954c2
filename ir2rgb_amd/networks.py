"""MI355X-native generator / discriminator modules behind the reference's factory API.

Drop-in for the factories of reference models/networks.py:

    build_generator_module(input_nc, output_nc, prev_output_nc, ngf, model_name, n_downsampling,
                           norm, scale, **opt)                                   (:51-75)
    build_discriminator_module(input_nc, ndf, n_layers_D, norm, num_D, get_interm_feat)   (:78-82)

Same signatures, same returned-module call signatures / return tuples (:191,:220,:288,:317,
:656-668), same ``state_dict`` key names and the same construction order of the parameter
containers -- so the same ``torch.manual_seed`` yields bit-identical initial weights and
reference checkpoints (``*_net_G0.pth`` ...) load unchanged.  What differs is everything that
executes: ``forward`` never calls a torch convolution; it drives the hand-written gfx950 kernels
of libir2rgb_hip.so (MFMA implicit-GEMM convolutions on NHWC half tensors with fp32
accumulation, fused BatchNorm statistics, fp32 heads / warp / blend).  There is no CPU path:
CPU tensors raise.

Compute precision: ``module.compute_dtype`` (torch.bfloat16 default, torch.float16 optional) is
the MFMA operand type of the dense layers; batch statistics, heads, flow, warp and blend are fp32.

Feature maps returned for the next scale (``img_feat``, ``flow_feat``) are logical NCHW tensors
in channels_last half precision (the kernels' native layout, zero copy); ``.float()`` gives the
reference's dense fp32 form.
"""
import copy
import functools

import torch
import torch.nn as nn
from torch.nn import init

from . import autograd as A
from . import conv as C
from . import layers as L

__all__ = ["build_generator_module", "build_discriminator_module", "get_grid", "weights_init", "get_norm_layer",
           "CompositeGeneratorModule", "CompositeLocalGeneratorModule", "GlobalGenerator", "MultiScaleDiscriminator",
           "NLayerDiscriminator", "ResnetBlock", "branch_streams"]


# ---------------------------------------------------------------------------------------------
# helpers shared with the reference API surface
# ---------------------------------------------------------------------------------------------
def get_grid(batch_size, rows, cols, device="cuda:0", dtype=torch.float32):
    """[-1,1] x [-1,1] lattice, channel 0 = x, channel 1 = y (reference networks.py:15-28)."""
    xs = torch.linspace(-1.0, 1.0, cols).view(1, 1, 1, cols).expand(batch_size, 1, rows, cols)
    ys = torch.linspace(-1.0, 1.0, rows).view(1, 1, rows, 1).expand(batch_size, 1, rows, cols)
    return torch.cat([xs, ys], 1).to(dtype).to(device)


def weights_init(m):
    """Reference init (networks.py:31-38): conv W ~ N(0, 0.02); BN gamma ~ N(1, 0.02), beta = 0."""
    if isinstance(m, (nn.Conv2d, nn.Conv3d)):
        init.normal_(m.weight, 0.0, 0.02)
    if isinstance(m, nn.BatchNorm2d):
        init.normal_(m.weight, 1.0, 0.02)
        init.zeros_(m.bias)
    if isinstance(m, nn.InstanceNorm2d) and m.weight is not None:
        init.normal_(m.weight, 1.0, 0.02)


def get_norm_layer(norm_type="instance"):
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == "instance":
        # The reference's instance branch crashes in weights_init (affine=False leaves weight=None,
        # networks.py:37-38,:45); the IR->RGB path always runs norm='batch'.
        raise NotImplementedError("norm='instance' is not runnable in the reference either; use norm='batch'")
    raise NotImplementedError(f"normalization layer {norm_type} is not found")


def _seq(mods):
    return nn.Sequential(*mods)


class ResnetBlock(nn.Module):
    """x + [pad, conv3x3, norm, ReLU, pad, conv3x3, norm](x)   (reference networks.py:547-586).

    ``conv_block`` indices 1,2,5,6 hold the parameters, as in the reference."""

    def __init__(self, dim, padding_type, norm_layer, activation=None, use_dropout=False):
        super().__init__()
        if padding_type != "reflect" or use_dropout:
            raise NotImplementedError("only padding_type='reflect' without dropout is used by the IR->RGB path")
        self.conv_block = _seq([nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0), norm_layer(dim),
                                nn.ReLU(True), nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0),
                                norm_layer(dim)])

    def run(self, x, extra_residual=None):
        """NHWC half in/out; ``extra_residual`` is added to the block output (encoder sum).  It may be a callable
        returning the tensor: it is called as late as possible (a branch computed on another stream is joined there)."""
        cb, tr, dt = self.conv_block, self.training, x.dtype
        h = A.conv_stage(x, cb[1], cb[2], L.ACT_RELU, C.PAD_REFLECT, dt, pad=1, training=tr)
        if callable(extra_residual):
            extra_residual = extra_residual()
        return A.conv_stage(h, cb[5], cb[6], L.ACT_NONE, C.PAD_REFLECT, dt, pad=1, res1=x, res2=extra_residual,
                            training=tr)

    def forward(self, x):
        dt = getattr(self, "compute_dtype", torch.bfloat16)
        return _real(self.run(_padded(A.to_nhwc_half(x, dt))), x.shape[1])


# ---------------------------------------------------------------------------------------------
# runners over nn.Sequential parameter containers
# ---------------------------------------------------------------------------------------------
def _run_sequence(seq, x, dtype, training, final_residual=None):
    """Executes a reference-structured nn.Sequential on the HIP kernels.

    Recognised runs: [ReflectionPad2d(3), Conv7x7, BN, ReLU] on an NCHW fp32 image (first layer),
    [Conv3x3 s2, BN, ReLU], [ConvTranspose3x3 s2, BN, ReLU], ResnetBlock.  ``final_residual`` is
    added to the output of the last ResnetBlock (fuses the encoder sum of networks.py:192)."""
    mods = list(seq)
    i, n = 0, len(mods)
    last_block = max((j for j, m in enumerate(mods) if isinstance(m, ResnetBlock)), default=-1)
    while i < n:
        m = mods[i]
        if isinstance(m, nn.ReflectionPad2d):
            conv, bn = mods[i + 1], mods[i + 2]
            x = A.conv_stage(x, conv, bn, L.ACT_RELU, C.PAD_REFLECT, dtype, first=True, pad=3, training=training)
            i += 4
        elif isinstance(m, nn.ConvTranspose2d):
            x = A.conv_stage(x, m, mods[i + 1], L.ACT_RELU, C.PAD_ZERO, dtype, transposed=True,
                             output_padding=m.output_padding[0], training=training)
            i += 3
        elif isinstance(m, nn.Conv2d):
            x = A.conv_stage(x, m, mods[i + 1], L.ACT_RELU, C.PAD_ZERO, dtype, training=training)
            i += 3
        elif isinstance(m, ResnetBlock):
            x = m.run(x, final_residual if i == last_block else None)
            i += 1
        else:
            raise RuntimeError(f"unexpected module {type(m).__name__} in generator sequence")
    if final_residual is not None and last_block < 0:
        raise RuntimeError("final_residual needs a ResnetBlock in the sequence")
    return x


def _first7(nc_in, nc_out, norm_layer):
    return [nn.ReflectionPad2d(3), nn.Conv2d(nc_in, nc_out, kernel_size=7, padding=0), norm_layer(nc_out), nn.ReLU(True)]


def _down(nc_in, norm_layer):
    return [nn.Conv2d(nc_in, nc_in * 2, kernel_size=3, stride=2, padding=1), norm_layer(nc_in * 2), nn.ReLU(True)]


def _up(nc_in, norm_layer):
    return [nn.ConvTranspose2d(nc_in, nc_in // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
            norm_layer(nc_in // 2), nn.ReLU(True)]


def _padded(x):
    """A half feature map entering the kernels' domain: widths that are not a power of two >= 64 are zero-padded to
    the next one (autograd.padded_width); the usual widths pass through untouched."""
    return A.pad_channels(x, A.padded_width(x.shape[1]))


def _real(x, channels):
    """The reference-shaped view of a feature map that ran at a padded width (a dense copy: the extra channels are
    zero and callers expect [N, channels, H, W])."""
    if x is None or x.shape[1] == channels:
        return x
    return x[:, :channels].contiguous(memory_format=torch.channels_last)


# ---------------------------------------------------------------------------------------------
# independent branches on two HIP streams.  The composite generators are two chains twice over: the label encoder
# and the previous-frame encoder (networks.py:192 / :290 sums them), then the image decoder and the flow decoder
# (:193-201 / :291-299).  One chain alone alternates chip-filling convolutions with BatchNorm statistics / apply
# kernels that are a few MB of traffic behind a dependent launch -- the chip idles through a fifth of the time.  Two
# chains on two streams fill each other's gaps (the convolutions are MFMA-bound, the BatchNorm passes HBM- and
# latency-bound).  Same kernels, same operands: results are bit-identical to the one-stream order
# (tests/test_networks_gpu.py::test_branch_streams_are_bit_exact: varying inputs, both generators;
# tests/test_streams_gpu.py: the first forward of fresh modules).
# Measured on the 512x1024 single-scale forward (tools/prof_forward.py --graph): 7.07 -> 6.61 ms.
#
# IR2RGB_BRANCH_STREAMS: "auto" (default) = autograd-free forwards take two streams, forwards that record a graph stay
# on one; "1" = always; "0" = never; ``branch_streams()`` overrides it for a block.  Under autograd torch replays every
# node on the stream its forward ran on, so loss.backward() would overlap the same way (37.4 -> 36.4 ms per training
# window measured); the trainer keeps one stream because bench.py's per-kernel roofline bracket must time kernels that
# have the chip to themselves.
# History: round 2 confined two streams to warm modules after a fresh module's first forward came out locally wrong
# (8-24 consecutive channels of a stage off, a different stage every time).  Round 3 found the cause -- not a stream-
# ordering bug at all: a write-after-read race INSIDE the convolution kernels' LDS rings that needs an LDS-heavy kernel of
# another stream on the same CU to show (the cold forward's weight-packing kernel); see conv_mfma.hip wait_stage,
# tools/check_lds_war.py and DESIGN.md section 8.  With that fixed the restriction is gone.
# ---------------------------------------------------------------------------------------------
import contextlib as _contextlib
import os as _os

BRANCH_STREAMS = _os.environ.get("IR2RGB_BRANCH_STREAMS", "auto")   # "auto" | "1" | "0" | set by branch_streams()
_SIDE_STREAMS = {}
# side-stream slot of the image discriminator's scale 1 (scale i: + i - 1).  Default 0 = the generators' second-branch
# stream, which is idle while the discriminators run, so that fewer streams compete for the device's four hardware queues
# (26.2-27.0 ms per window against 26.4-27.9 with a stream of its own, same box; GPU_MAX_HW_QUEUES=6 / 8 made the window
# 45 % SLOWER: the streams are fitted to the queues, not the other way round).
_D_SCALE_SLOT0 = int(_os.environ.get("IR2RGB_D_SCALE_SLOT", "0"))


@_contextlib.contextmanager
def branch_streams(enabled=True):
    """Run the generators' independent branches on two HIP streams inside this context (or, ``enabled=False``, on one)."""
    global BRANCH_STREAMS
    old, BRANCH_STREAMS = BRANCH_STREAMS, "1" if enabled else "0"
    try:
        yield
    finally:
        BRANCH_STREAMS = old


class _Branch:
    """``with _Branch(x) as b: y = f(x)`` runs the body on the device's side stream after everything queued on the
    current stream; ``b.join(y, ...)`` makes the current stream wait for it and returns the tensors."""

    def __init__(self, *inputs, owner=None, slot=0, force=None):
        self.inputs = [t for t in inputs if isinstance(t, torch.Tensor)]
        # (``owner.branch_streams_training``: a module may ask for two streams also while autograd records -- the trainer
        # sets it on the finer spatial scales, whose kernels bench.py does not bracket)
        # ``slot``: which of the device's side streams (0: the generators' second branch; the discriminators use their own,
        # see MultiScaleDiscriminator.forward and Vid2VidTrainer.train_window).  ``force``: the caller decides (True / False)
        # instead of IR2RGB_BRANCH_STREAMS.
        if force is None:
            on = BRANCH_STREAMS == "1" or (BRANCH_STREAMS == "auto" and (not torch.is_grad_enabled() or
                                                                          getattr(owner, "branch_streams_training", False)))
        else:
            on = bool(force) and BRANCH_STREAMS != "0"
        self.enabled = on and bool(self.inputs) and self.inputs[0].is_cuda      # (also inside a HIP-graph capture: the fork / join is captured)
        self.slot = slot
        self.ctx = None

    def __enter__(self):
        if self.enabled:
            dev = self.inputs[0].device
            self.main = torch.cuda.current_stream(dev)
            key = dev.index if self.slot == 0 else (dev.index, self.slot)
            self.side = _SIDE_STREAMS.get(key)
            if self.side is None:
                self.side = _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
            self.side.wait_stream(self.main)
            for t in self.inputs:
                t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self, *outs):
        if self.enabled:
            self.main.wait_stream(self.side)
            for t in outs:
                if isinstance(t, torch.Tensor):
                    t.record_stream(self.main)
            self.enabled = False
        return outs[0] if len(outs) == 1 else outs


class _CompositeBase(nn.Module):
    compute_dtype = torch.bfloat16

    def _img_head(self, img_feat):
        return A.head_stage(img_feat, [self.model_final_img[1]], [1] * self.model_final_img[1].out_channels)

    def _flow_heads(self, flow_feat, flow_mul):
        """flow (x flow_mul) and weight (sigmoid) heads evaluated as one separable convolution: [N,3,H,W] fp32."""
        return A.head_stage(flow_feat, [self.model_final_flow[1], self.model_final_w[1]], [0, 0, 2], mul=flow_mul)

    def _blend(self, img_raw, fw, img_prev, use_raw_only):
        flow = weight = None
        if fw is not None:
            flow, weight = fw[:, 0:2], fw[:, 2:3]
        if use_raw_only or self.no_flow:
            img_final = img_raw
        else:
            img_final = A.warp_blend(img_raw, img_prev.float(), flow, weight)
        return img_final, flow, weight

    @staticmethod
    def _check_inputs(input, img_prev):
        if not input.is_cuda or not img_prev.is_cuda:
            raise ValueError("ir2rgb_amd generators run on an AMD GPU only (no CPU fallback)")


class CompositeGeneratorModule(_CompositeBase):
    """Coarse-scale vid2vid generator (reference networks.py:103-220), HIP execution."""

    def __init__(self, input_nc, output_nc, prev_output_nc, ngf, n_downsampling, n_blocks, use_fg_model=False,
                 no_flow=False, norm_layer=nn.BatchNorm2d, padding_type="reflect"):
        super().__init__()
        if use_fg_model:
            raise NotImplementedError("foreground model (fg=True) is a dead branch for IR->RGB")
        self.use_fg_model, self.no_flow, self.ngf = False, no_flow, ngf
        res = lambda: ResnetBlock(ngf * 2 ** n_downsampling, padding_type, norm_layer)  # noqa: E731

        # construction order == reference order: same RNG stream, same initial weights
        down_seg = _first7(input_nc, ngf, norm_layer)
        for i in range(n_downsampling):
            down_seg += _down(ngf * 2 ** i, norm_layer)
        down_seg += [res() for _ in range(n_blocks - n_blocks // 2)]
        down_img = _first7(prev_output_nc, ngf, norm_layer) + copy.deepcopy(down_seg[4:])
        res_img = [res() for _ in range(n_blocks // 2)]
        res_flow = copy.deepcopy(res_img) if not no_flow else None
        up_img = []
        for i in range(n_downsampling):
            up_img += _up(ngf * 2 ** (n_downsampling - i), norm_layer)
        final_img = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
        if not no_flow:
            up_flow = copy.deepcopy(up_img)
            final_flow = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, 2, kernel_size=7, padding=0)]
            final_w = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, 1, kernel_size=7, padding=0), nn.Sigmoid()]

        self.model_down_seg = _seq(down_seg)
        self.model_down_img = _seq(down_img)
        self.model_res_img = _seq(res_img)
        self.model_up_img = _seq(up_img)
        self.model_final_img = _seq(final_img)
        if not no_flow:
            self.model_res_flow = _seq(res_flow)
            self.model_up_flow = _seq(up_flow)
            self.model_final_flow = _seq(final_flow)
            self.model_final_w = _seq(final_w)

    def forward(self, input, img_prev, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only):
        self._check_inputs(input, img_prev)
        dt, tr = self.compute_dtype, self.training
        with _Branch(input, owner=self) as enc:              # the label encoder beside the previous-frame encoder
            seg = _run_sequence(self.model_down_seg, input, dt, tr)
        downsample = _run_sequence(self.model_down_img, img_prev, dt, tr, final_residual=lambda: enc.join(seg))  # (:192)
        flow_feat = fw = None
        with _Branch(downsample, owner=self) as dec:         # the flow decoder + its heads beside the image decoder
            if not self.no_flow:
                flow_feat = _run_sequence(self.model_up_flow, _run_sequence(self.model_res_flow, downsample, dt, tr), dt, tr)
                fw = self._flow_heads(flow_feat, 20.0)
        img_feat = _run_sequence(self.model_up_img, _run_sequence(self.model_res_img, downsample, dt, tr), dt, tr)
        img_raw = self._img_head(img_feat)
        dec.join(flow_feat, fw)
        img_final, flow, weight = self._blend(img_raw, fw, img_prev, use_raw_only)
        L.flush_bn_counters()
        return img_final, flow, weight, img_raw, _real(img_feat, self.ngf), _real(flow_feat, self.ngf), None


class CompositeLocalGeneratorModule(_CompositeBase):
    """Finer-scale generator fed with coarse features (reference networks.py:223-317), HIP execution."""

    def __init__(self, input_nc, output_nc, prev_output_nc, ngf, n_downsampling, n_blocks_local, use_fg_model=False,
                 no_flow=False, norm_layer=nn.BatchNorm2d, padding_type="reflect", scale=1):
        super().__init__()
        if use_fg_model:
            raise NotImplementedError("foreground model (fg=True) is a dead branch for IR->RGB")
        self.use_fg_model, self.no_flow, self.scale, self.ngf = False, no_flow, scale, ngf
        down_seg = _first7(input_nc, ngf, norm_layer) + _down(ngf, norm_layer)
        down_img = _first7(prev_output_nc, ngf, norm_layer) + _down(ngf, norm_layer)
        up_img = [ResnetBlock(ngf * 2, padding_type, norm_layer) for _ in range(n_blocks_local)] + _up(ngf * 2, norm_layer)
        final_img = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
        if not no_flow:
            up_flow = copy.deepcopy(up_img)
            final_flow = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, 2, kernel_size=7, padding=0)]
            final_w = [nn.ReflectionPad2d(3), nn.Conv2d(ngf, 1, kernel_size=7, padding=0), nn.Sigmoid()]
        self.model_down_seg = _seq(down_seg)
        self.model_down_img = _seq(down_img)
        self.model_up_img = _seq(up_img)
        self.model_final_img = _seq(final_img)
        if not no_flow:
            self.model_up_flow = _seq(up_flow)
            self.model_final_flow = _seq(final_flow)
            self.model_final_w = _seq(final_w)

    def _encode(self, seq, x, dt, tr, res1=None):
        mods = list(seq)
        h = A.conv_stage(x, mods[1], mods[2], L.ACT_RELU, C.PAD_REFLECT, dt, first=True, pad=3, training=tr)
        if callable(res1):      # the other encoder ran on the side stream: joined as late as possible
            res1 = res1()
        # the stride-2 stage ends the encoder: the other encoder's output is added in its epilogue pass
        return A.conv_stage(h, mods[4], mods[5], L.ACT_RELU, C.PAD_ZERO, dt, res1=res1, training=tr)

    def forward(self, input, img_prev, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only):
        self._check_inputs(input, img_prev)
        dt, tr = self.compute_dtype, self.training
        with _Branch(input, owner=self) as enc:
            seg = self._encode(self.model_down_seg, input, dt, tr)
        down_img = self._encode(self.model_down_img, img_prev, dt, tr, res1=lambda: enc.join(seg))  # (:290)
        flow_feat = fw = None
        with _Branch(down_img, flow_feat_coarse, owner=self) as dec:
            if not self.no_flow:
                flow_in = A.add(down_img, _padded(A.to_nhwc_half(flow_feat_coarse, dt)))   # (:297)
                flow_feat = _run_sequence(self.model_up_flow, flow_in, dt, tr)
                fw = self._flow_heads(flow_feat, 20.0 * (2 ** self.scale))
        img_in = A.add(down_img, _padded(A.to_nhwc_half(img_feat_coarse, dt)))             # (:291)
        img_feat = _run_sequence(self.model_up_img, img_in, dt, tr)
        img_raw = self._img_head(img_feat)
        dec.join(flow_feat, fw)
        img_final, flow, weight = self._blend(img_raw, fw, img_prev, use_raw_only)
        L.flush_bn_counters()
        return img_final, flow, weight, img_raw, _real(img_feat, self.ngf), _real(flow_feat, self.ngf), None


class GlobalGenerator(nn.Module):
    """pix2pixHD global generator (reference networks.py:320-352, factory name 'global'): one nn.Sequential ``model`` =
    [ReflPad3, Conv7x7, norm, ReLU] + n_downsampling x [Conv3x3 s2, norm, ReLU] (channels capped at 1024) + n_blocks
    ResnetBlocks + n_downsampling x [ConvT3x3 s2, norm, ReLU] + [ReflPad3, Conv7x7, Tanh].  HIP execution: the same
    stage runner as the composite generators, the head as the separable 7x7 kernel with tanh."""
    compute_dtype = torch.bfloat16

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type="reflect"):
        assert n_blocks >= 0
        super().__init__()
        ch_max = 1024
        model = _first7(input_nc, ngf, norm_layer)
        for i in range(n_downsampling):
            mult = 2 ** i
            model += [nn.Conv2d(min(ch_max, ngf * mult), min(ch_max, ngf * mult * 2), kernel_size=3, stride=2, padding=1),
                      norm_layer(min(ch_max, ngf * mult * 2)), nn.ReLU(True)]
        mult = 2 ** n_downsampling
        model += [ResnetBlock(min(ch_max, ngf * mult), padding_type, norm_layer) for _ in range(n_blocks)]
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            model += [nn.ConvTranspose2d(min(ch_max, ngf * mult), min(ch_max, int(ngf * mult / 2)), kernel_size=3, stride=2,
                                         padding=1, output_padding=1), norm_layer(min(ch_max, int(ngf * mult / 2))), nn.ReLU(True)]
        model += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
        self.model = _seq(model)

    def forward(self, input, feat=None):
        if feat is not None:
            input = torch.cat([input, feat], dim=1)
        if not input.is_cuda:
            raise ValueError("ir2rgb_amd generators run on an AMD GPU only (no CPU fallback)")
        mods = list(self.model)
        x = _run_sequence(mods[:-3], input, self.compute_dtype, self.training)
        out = A.head_stage(x, [mods[-2]], [1] * mods[-2].out_channels)
        L.flush_bn_counters()
        return out


# ---------------------------------------------------------------------------------------------
# discriminators
# ---------------------------------------------------------------------------------------------
class NLayerDiscriminator(nn.Module):
    """PatchGAN (reference networks.py:672-718): Conv4x4 s2 -> LReLU; (n_layers-1) x [Conv4x4 s2, BN,
    LReLU]; [Conv4x4 s1, BN, LReLU]; Conv4x4 s1 -> 1 channel.  padding = ceil(3/2) = 2."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, getIntermFeat=False):
        super().__init__()
        self.getIntermFeat, self.n_layers = getIntermFeat, n_layers
        kw, padw = 4, 2
        groups = [[nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            groups.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=2, padding=padw), norm_layer(nf),
                           nn.LeakyReLU(0.2, True)])
        nf_prev, nf = nf, min(nf * 2, 512)
        groups.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=1, padding=padw), norm_layer(nf),
                       nn.LeakyReLU(0.2, True)])
        groups.append([nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)])
        if getIntermFeat:
            for n, g in enumerate(groups):
                setattr(self, "model" + str(n), _seq(g))
        else:
            self.model = _seq([m for g in groups for m in g])


def _run_patchgan(groups, x_nchw, dtype, training, sample_groups=1, group_order=None):
    """groups: list of nn.Sequential ([conv, (bn), (lrelu)]).  Returns every group's output (channels_last
    half; the last one -- the 1-channel logits -- as fp32).  ``sample_groups``: see MultiScaleDiscriminator.forward."""
    g0 = groups[0]
    h = A.conv_stage(x_nchw, g0[0], None, L.ACT_NONE, C.PAD_ZERO, dtype, first=True, fused_leaky=True, training=training)
    outs = [_real(h, g0[0].out_channels)]
    for g in groups[1:-1]:
        h = A.conv_stage(h, g[0], g[1], L.ACT_LEAKY, C.PAD_ZERO, dtype, training=training, groups=sample_groups,
                         group_order=group_order)
        outs.append(_real(h, g[0].out_channels))
    outs.append(A.conv_stage(h, groups[-1][0], None, L.ACT_NONE, C.PAD_ZERO, dtype, out_f32=True, training=training))
    return outs


class MultiScaleDiscriminator(nn.Module):
    """num_D PatchGANs on an average-pooled pyramid (reference networks.py:627-668)."""
    compute_dtype = torch.bfloat16

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, num_D=3, getIntermFeat=False):
        super().__init__()
        self.num_D, self.n_layers, self.getIntermFeat = num_D, n_layers, getIntermFeat
        ndf_max = 64
        for i in range(num_D):
            net = NLayerDiscriminator(input_nc, min(ndf_max, ndf * (2 ** (num_D - 1 - i))), n_layers, norm_layer,
                                      getIntermFeat)
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, f"scale{i}_layer{j}", getattr(net, "model" + str(j)))
            else:
                setattr(self, f"layer{i}", net.model)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def _groups(self, i):
        if self.getIntermFeat:
            return [getattr(self, f"scale{i}_layer{j}") for j in range(self.n_layers + 2)]
        flat = list(getattr(self, f"layer{i}"))
        groups, cur = [], []
        for m in flat:
            if isinstance(m, nn.Conv2d) and cur:
                groups.append(cur)
                cur = []
            cur.append(m)
        groups.append(cur)
        return groups

    def forward(self, input, sample_groups=1, group_order=None):
        """``sample_groups`` = G > 1: ``input`` stacks G independent forwards along the batch axis (N / G samples each,
        e.g. real | generated | raw frames, discriminator.py:154-166).  Every convolution then runs ONCE over the whole
        batch -- the deep layers of a one-frame forward fill a fraction of the chip -- while BatchNorm keeps treating the
        groups as separate calls (batch statistics per group, running statistics advanced group by group -- in
        ``group_order`` if given, else in batch order), so each group's outputs are what its own forward would have produced."""
        if not input.is_cuda:
            raise ValueError("ir2rgb_amd discriminators run on an AMD GPU only (no CPU fallback)")
        if input.shape[0] % sample_groups:
            raise ValueError("MultiScaleDiscriminator: the batch is not a multiple of sample_groups")
        result = []
        x = input.float().contiguous()
        if getattr(self, "scale_streams", False) and self.num_D > 1:
            # The PatchGANs of the pyramid are independent networks of small, latency-shaped layers (a 512-channel layer at
            # 34 x 66 gives a convolution 36 workgroups for 256 CUs): the coarser scales run on their own HIP streams beside
            # the finest one (set by the trainer, ``scale_streams``).  Under autograd every node's backward runs on the
            # stream of its forward, so the backward passes overlap the same way.  Same kernels, same operands.
            xs = [x]
            for i in range(1, self.num_D):
                xs.append(A.avg_pool3s2(xs[-1]))
            pending = []
            for i in range(1, self.num_D):
                br = _Branch(xs[i], slot=_D_SCALE_SLOT0 + i - 1, force=True)
                with br:
                    outs = _run_patchgan(self._groups(self.num_D - 1 - i), xs[i], self.compute_dtype, self.training,
                                         sample_groups, group_order)
                pending.append((br, outs))
            outs0 = _run_patchgan(self._groups(self.num_D - 1), xs[0], self.compute_dtype, self.training, sample_groups,
                                  group_order)
            result.append(outs0 if self.getIntermFeat else [outs0[-1]])
            for br, outs in pending:
                br.join(*outs)
                result.append(outs if self.getIntermFeat else [outs[-1]])
            L.flush_bn_counters()
            return result
        for i in range(self.num_D):
            outs = _run_patchgan(self._groups(self.num_D - 1 - i), x, self.compute_dtype, self.training, sample_groups,
                                 group_order)
            result.append(outs if self.getIntermFeat else [outs[-1]])
            if i != self.num_D - 1:
                x = A.avg_pool3s2(x)       # = self.downsample (networks.py:639), forward and backward in one HIP kernel each
        L.flush_bn_counters()
        return result


# ---------------------------------------------------------------------------------------------
# factories
# ---------------------------------------------------------------------------------------------
def build_generator_module(input_nc, output_nc, prev_output_nc, ngf, model_name, n_downsampling, norm, scale, **opt):
    norm_layer = get_norm_layer(norm_type=norm)
    if model_name == "composite":
        generator = CompositeGeneratorModule(input_nc, output_nc, prev_output_nc, ngf, n_downsampling, opt["gen_blocks"],
                                             opt["fg"], opt["no_flow"], norm_layer)
    elif model_name == "composite-local":
        generator = CompositeLocalGeneratorModule(input_nc, output_nc, prev_output_nc, ngf, n_downsampling,
                                                  opt["n_blocks_local"], opt["fg"], opt["no_flow"], norm_layer,
                                                  scale=scale)
    elif model_name == "global":
        generator = GlobalGenerator(input_nc, output_nc, ngf, n_downsampling, opt["gen_blocks"], norm_layer)
    elif model_name in ("local", "global-with-features", "local-with-features", "encoder"):
        raise NotImplementedError(f"generator '{model_name}' is outside the IR->RGB hot path (SURVEY section 2, row 1)")
    else:
        raise NotImplementedError(f"Generator model named {model_name} is not implemented")
    generator.apply(weights_init)
    return generator


def build_discriminator_module(input_nc, ndf, n_layers_D, norm="instance", num_D=1, get_interm_feat=False):
    norm_layer = get_norm_layer(norm_type=norm)
    discriminator = MultiScaleDiscriminator(input_nc, ndf, n_layers_D, norm_layer, num_D, get_interm_feat)
    discriminator.apply(weights_init)
    return discriminator
