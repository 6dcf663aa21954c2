"""Generates the harness goldens from the REFERENCE's own code.

What is imported from /root/reference (importable in the authoring container; never travels to the GPU box):

    models.networks     build_generator_module / build_discriminator_module / get_grid     (networks.py:15-82)
    models.loss         GANLoss, MaskedL1Loss                                               (loss.py:8-41, :105-113)
    models.utils        concat                                                              (utils.py:12-24)
    models.flownet2_pytorch.networks.{FlowNetS,FlowNetSD,FlowNetFusion,submodules}         (pure torch)

What is NOT importable (``models.generator`` / ``models.discriminator`` pull in cv2 through util/util.py,
``models.flownet`` and ``FlowNetC`` pull in the CUDA extensions; ordinary ModuleNotFoundError) is composed here
from the pieces above, statement by statement, each block citing the reference lines it follows:

    Vid2VidGenerator.forward / generate_frame_train / generate_first_frame   generator.py:99-182, :217-235
    Model.build_pyr / Model.resample                                         base_model.py:64-82, :123-136
    Vid2VidModelD.forward / compute_loss_D / compute_loss_D_T /
        GAN_and_FM_loss / get_losses / get_skipped_frames / get_skipped_flows discriminator.py:90-200, :236-284
    train loop body, loss_backward, reshape                                  train_vid2vid.py:54-111, :166-178
    FlowNet2.forward, FlowNet.compute_flow_and_conf                          flownet2_pytorch/models.py:96-161, flownet.py:38-57

Two reference behaviours that cannot run on CPU are worked around exactly as tests/golden/make_net_goldens.py does:
the generator's own warp calls ``.cuda()`` (networks.py:93-100), so the module runs with use_raw_only=True and the
blend of networks.py:207-209 is applied outside with the same arithmetic; ``Model.resample`` likewise.
FlowNet (pretrained checkpoint = a download) is replaced by tests/golden/window_stub.py on both sides.

Weights are never stored: both trees are built from the same seed and asserted bit-identical here.

    python tests/golden/make_window_goldens.py [flownet] [losses] [window]
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, OUT)
sys.path.insert(0, "/root/reference")
from models import loss as ref_loss          # noqa: E402  (the reference)
from models import networks as ref           # noqa: E402
from models.utils import concat              # noqa: E402
from models.flownet2_pytorch.networks import FlowNetFusion as ref_fusion   # noqa: E402
from models.flownet2_pytorch.networks import FlowNetS as ref_s             # noqa: E402
from models.flownet2_pytorch.networks import FlowNetSD as ref_sd           # noqa: E402

from window_stub import smooth, stub_flow_and_conf, stub_flownetc          # noqa: E402

from ir2rgb_amd import networks as mine                                    # noqa: E402
from ir2rgb_amd.flownet2_pytorch import models as mine_fn                  # noqa: E402
from oracle import closed_form                                             # noqa: E402
from oracle import emulated                                                # noqa: E402

G_OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


def assert_same_init(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys()), "state_dict keys differ"
    for k in sa:
        assert torch.equal(sa[k], sb[k]), f"init differs at {k}"


# ------------------------------------------------------------------------------------------------
# a12: FlowNet2 sub-networks and the FlowNet2 / FlowNet composition
# ------------------------------------------------------------------------------------------------
class _Args(dict):
    pass


def _tame(m):
    """The reference init (uniform(0,1) biases, models.py:68-77) blows activations up layer by layer; a trained
    checkpoint does not.  Biases x0.05 on both sides keeps the fp32-vs-half comparison in range."""
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.mul_(0.05)
    return m


def _ref_subnet(kind, seed):
    args = _Args()
    args.rgb_max, args.fp16, args.grads = 1, False, {}
    torch.manual_seed(seed)
    if kind == "S":
        r = ref_s.FlowNetS(args, batchNorm=False)
    elif kind == "SD":
        r = ref_sd.FlowNetSD(args, batchNorm=False)
    else:
        r = ref_fusion.FlowNetFusion(args, batchNorm=False)
    torch.manual_seed(seed)
    m = {"S": mine_fn.FlowNetS, "SD": mine_fn.FlowNetSD, "F": mine_fn.FlowNetFusion}[kind]()
    assert_same_init(r, m)
    return _tame(r).eval()


def flownet_cases():
    for kind, cin, seed in (("S", 12, 21), ("SD", 6, 22), ("F", 11, 23)):
        r = _ref_subnet(kind, seed)
        x = smooth((1, cin, 64, 128), seed + 100) * 0.5
        with torch.no_grad():
            out = r(x)
        out = out[0] if isinstance(out, tuple) else out
        np.savez_compressed(os.path.join(OUT, f"flownet_{kind}.npz"), seed=seed, x=x.numpy(), out=out.numpy(),
                            keys=np.array(list(r.state_dict().keys())))
        print("flownet", kind, tuple(out.shape), float(out.abs().mean()))

    # --- FlowNet2.forward (models.py:96-161) with FlowNetC stubbed, then FlowNet.compute_flow_and_conf (flownet.py:38-57)
    s1, s2, sd, fu = _ref_subnet("S", 31), _ref_subnet("S", 32), _ref_subnet("SD", 33), _ref_subnet("F", 34)
    up_bil = nn.Upsample(scale_factor=4, mode="bilinear")      # upsample1 / upsample2 (models.py:48, :52)
    up_near = nn.Upsample(scale_factor=4, mode="nearest")      # upsample3 / upsample4 (models.py:59-60)
    resample, channelnorm = closed_form.resample2d, closed_form.channelnorm    # pinned by tests/test_oracle_ops.py
    div_flow, rgb_max = 20.0, 1

    def flownet2_forward(inputs):
        rgb_mean = inputs.contiguous().view(inputs.size()[:2] + (-1,)).mean(dim=-1).view(inputs.size()[:2] + (1, 1, 1,))
        x = (inputs - rgb_mean) / rgb_max
        x1, x2 = x[:, :, 0, :, :], x[:, :, 1, :, :]
        x = torch.cat((x1, x2), dim=1)
        flownetc_flow2 = stub_flownetc(x)                                   # models.py:104 (stubbed)
        flownetc_flow = up_bil(flownetc_flow2 * div_flow)
        resampled_img1 = resample(x[:, 3:, :, :], flownetc_flow)
        diff_img0 = x[:, :3, :, :] - resampled_img1
        norm_diff_img0 = channelnorm(diff_img0)
        concat1 = torch.cat((x, resampled_img1, flownetc_flow / div_flow, norm_diff_img0), dim=1)
        flownets1_flow2 = s1(concat1)[0]
        flownets1_flow = up_bil(flownets1_flow2 * div_flow)
        resampled_img1 = resample(x[:, 3:, :, :], flownets1_flow)
        diff_img0 = x[:, :3, :, :] - resampled_img1
        norm_diff_img0 = channelnorm(diff_img0)
        concat2 = torch.cat((x, resampled_img1, flownets1_flow / div_flow, norm_diff_img0), dim=1)
        flownets2_flow2 = s2(concat2)[0]
        flownets2_flow = up_near(flownets2_flow2 * div_flow)
        norm_flownets2_flow = channelnorm(flownets2_flow)
        diff_flownets2_flow = resample(x[:, 3:, :, :], flownets2_flow)
        diff_flownets2_img1 = channelnorm((x[:, :3, :, :] - diff_flownets2_flow))
        flownetsd_flow2 = sd(x)[0]
        flownetsd_flow = up_near(flownetsd_flow2 / div_flow)
        norm_flownetsd_flow = channelnorm(flownetsd_flow)
        diff_flownetsd_flow = resample(x[:, 3:, :, :], flownetsd_flow)
        diff_flownetsd_img1 = channelnorm((x[:, :3, :, :] - diff_flownetsd_flow))
        concat3 = torch.cat((x[:, :3, :, :], flownetsd_flow, flownets2_flow, norm_flownetsd_flow, norm_flownets2_flow,
                             diff_flownetsd_img1, diff_flownets2_img1), dim=1)
        return fu(concat3)

    def compute_flow_and_conf(im1, im2):                                    # flownet.py:38-54
        old_h, old_w = im1.size()[2], im1.size()[3]
        new_h, new_w = old_h // 64 * 64, old_w // 64 * 64
        if old_h != new_h:
            downsample = torch.nn.Upsample(size=(new_h, new_w), mode="bilinear")
            upsample = torch.nn.Upsample(size=(old_h, old_w), mode="bilinear")
            im1, im2 = downsample(im1), downsample(im2)
        data1 = torch.cat([im1.unsqueeze(2), im2.unsqueeze(2)], dim=2)
        flow1 = flownet2_forward(data1)
        d = im1 - resample(im2, flow1)
        conf = (torch.sum(d * d, dim=1, keepdim=True) < 0.02).float()
        if old_h != new_h:
            flow1 = upsample(flow1) * old_h / new_h
            conf = upsample(conf)
        return flow1, conf

    out = {}
    for tag, (h, w) in (("a", (64, 128)), ("b", (80, 128))):           # b: the resize branch (80 -> 64 rows)
        base = smooth((1, 3, h + 8, w + 8), 40 + h, blur=15, gain=6.0)
        im1, im2 = base[:, :, 4:-4, 4:-4].contiguous(), base[:, :, 4:-4, 5:-3].contiguous()
        with torch.no_grad():
            flow, conf = compute_flow_and_conf(im1, im2)
            if tag == "a":
                out["flow2_a"] = flownet2_forward(torch.stack([im1, im2], 2)).numpy()
        out.update({f"im1_{tag}": im1.numpy(), f"im2_{tag}": im2.numpy(), f"flow_{tag}": flow.numpy(), f"conf_{tag}": conf.numpy()})
        print("flownet2 glue", tag, "flow |mean|", float(flow.abs().mean()), "conf mean", float(conf.mean()))
    np.savez_compressed(os.path.join(OUT, "flownet2_glue.npz"), seeds=np.array([31, 32, 33, 34]), **out)


# ------------------------------------------------------------------------------------------------
# a13: the discriminator-side loss assembly
# ------------------------------------------------------------------------------------------------
def ref_resample(image, flow):
    """Model.resample (base_model.py:129-136) without its .cuda() calls."""
    b, c, h, w = image.size()
    grid = ref.get_grid(b, h, w, device="cpu", dtype=flow.dtype)
    flow = torch.cat([flow[:, 0:1, :, :] / ((w - 1.0) / 2.0), flow[:, 1:2, :, :] / ((h - 1.0) / 2.0)], dim=1)
    final_grid = (grid + flow).permute(0, 2, 3, 1)
    return torch.nn.functional.grid_sample(image, final_grid, mode="bilinear", padding_mode="border")


class RefModelD:
    """Vid2VidModelD restated around the reference's networks and criteria (discriminator.py:19-88)."""
    loss_names = ["G_VGG", "G_GAN", "G_GAN_Feat", "D_real", "D_fake", "G_Warp", "F_Flow", "F_Warp", "W"]
    loss_names_T = ["G_T_GAN", "G_T_GAN_Feat", "D_T_real", "D_T_fake", "G_T_Warp"]

    emulate = None      # a torch half dtype: evaluate the SAME statements with oracle/emulated.py (rounding floor)

    def _D(self, net, x):
        return net.forward(x) if self.emulate is None else emulated.discriminator_forward(net, x, self.emulate)

    def _feat(self, a, b):
        return self.criterionFeat(a, b) if self.emulate is None else emulated.l1_half(a, b, 1.0, self.emulate)

    def __init__(self, opt, seeds):
        self.opt = opt
        self.tD, self.output_nc = opt["n_frames_D"], opt["output_nc"]
        nets = []
        for mod in (ref, mine):
            torch.manual_seed(seeds[0])
            d = mod.build_discriminator_module(opt["input_nc"] + opt["output_nc"], opt["first_layer_dis_filters"],
                                               opt["n_layers_D"], opt["norm"], opt["num_D"], not opt["no_ganFeat"])
            dts = []
            for s in range(opt["n_scales_temporal"]):
                torch.manual_seed(seeds[1] + s)
                dts.append(mod.build_discriminator_module(opt["output_nc"] * self.tD + 2 * (self.tD - 1),
                                                          opt["first_layer_dis_filters"], opt["n_layers_D"], opt["norm"],
                                                          opt["num_D"], not opt["no_ganFeat"]))
            nets.append([d] + dts)
        for a, b in zip(*nets):
            assert_same_init(a, b)
        self.netD, self.netD_T = nets[0][0].train(), [d.train() for d in nets[0][1:]]
        self.criterionGAN = ref_loss.GANLoss(opt["gan_mode"])
        self.criterionFlow = ref_loss.MaskedL1Loss()
        self.criterionWarp = ref_loss.MaskedL1Loss()
        self.criterionFeat = torch.nn.L1Loss()
        adam = dict(lr=opt["lr"], betas=(opt["beta1"], 0.999))
        self.optimizer_D = torch.optim.Adam(list(self.netD.parameters()), **adam)
        self.optimizer_D_T = [torch.optim.Adam(list(d.parameters()), **adam) for d in self.netD_T]

    def forward(self, scale_T, tensors_list):                               # discriminator.py:90-152
        lambda_feat, lambda_F, lambda_T = self.opt["lambda_feat"], self.opt["lambda_F"], self.opt["lambda_T"]
        scale_S = self.opt["n_scales_spatial"]
        if scale_T > 0:
            real_B, fake_B, flow_ref, conf_ref = tensors_list
            _, _, _, self.height, self.width = real_B.size()
            loss_D_T_real, loss_D_T_fake, loss_G_T_GAN, loss_G_T_GAN_Feat = self.compute_loss_D_T(
                real_B, fake_B, flow_ref / 20, conf_ref, scale_T - 1)
            loss_G_T_Warp = torch.zeros_like(loss_G_T_GAN)
            loss_list = [loss_G_T_GAN, loss_G_T_GAN_Feat, loss_D_T_real, loss_D_T_fake, loss_G_T_Warp]
            return [loss.view(-1, 1) for loss in loss_list]
        real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref = tensors_list
        _, _, self.height, self.width = real_B.size()
        loss_F_Flow = self.criterionFlow(flow, flow_ref, conf_ref) * lambda_F / (2 ** (scale_S - 1))
        real_B_warp = ref_resample(real_B_prev, flow)
        loss_F_Warp = self.criterionFlow(real_B_warp, real_B, conf_ref) * lambda_T
        loss_W = torch.zeros_like(weight)
        if self.opt["no_first_img"]:
            dummy0 = torch.zeros_like(weight)
            loss_W = self.criterionFlow(weight, dummy0, conf_ref)
        loss_G_VGG = torch.zeros_like(loss_W)                                 # no_vgg
        loss_D_real, loss_D_fake, loss_G_GAN, loss_G_GAN_Feat = self.compute_loss_D(self.netD, real_A, real_B, fake_B)
        fake_B_warp_ref = ref_resample(fake_B_prev, flow_ref)
        loss_G_Warp = self.criterionWarp(fake_B, fake_B_warp_ref.detach(), conf_ref) * lambda_T
        if fake_B_raw is not None:
            l_D_real, l_D_fake, l_G_GAN, l_G_GAN_Feat = self.compute_loss_D(self.netD, real_A, real_B, fake_B_raw)
            loss_G_GAN += l_G_GAN
            loss_G_GAN_Feat += l_G_GAN_Feat
            loss_D_real += l_D_real
            loss_D_fake += l_D_fake
        loss_list = [loss_G_VGG, loss_G_GAN, loss_G_GAN_Feat, loss_D_real, loss_D_fake, loss_G_Warp, loss_F_Flow, loss_F_Warp,
                     loss_W]
        return [loss.view(-1, 1) for loss in loss_list]

    def compute_loss_D(self, netD, real_A, real_B, fake_B):                 # discriminator.py:154-166
        real_AB = torch.cat((real_A, real_B), dim=1)
        fake_AB = torch.cat((real_A, fake_B), dim=1)
        pred_real = self._D(netD, real_AB)
        pred_fake = self._D(netD, fake_AB.detach())
        loss_D_real = self.criterionGAN(pred_real, True)
        loss_D_fake = self.criterionGAN(pred_fake, False)
        pred_fake = self._D(netD, fake_AB)
        loss_G_GAN, loss_G_GAN_Feat = self.GAN_and_FM_loss(pred_real, pred_fake)
        return loss_D_real, loss_D_fake, loss_G_GAN, loss_G_GAN_Feat

    def compute_loss_D_T(self, real_B, fake_B, flow_ref, conf_ref, scale_T):   # discriminator.py:168-184
        netD_T = self.netD_T[scale_T]
        real_B = real_B.view(-1, self.output_nc * self.tD, self.height, self.width)
        fake_B = fake_B.view(-1, self.output_nc * self.tD, self.height, self.width)
        if flow_ref is not None:
            flow_ref = flow_ref.view(-1, 2 * (self.tD - 1), self.height, self.width)
            real_B = torch.cat([real_B, flow_ref], dim=1)
            fake_B = torch.cat([fake_B, flow_ref], dim=1)
        pred_real = self._D(netD_T, real_B)
        pred_fake = self._D(netD_T, fake_B.detach())
        loss_D_T_real = self.criterionGAN(pred_real, True)
        loss_D_T_fake = self.criterionGAN(pred_fake, False)
        pred_fake = self._D(netD_T, fake_B)
        loss_G_T_GAN, loss_G_T_GAN_Feat = self.GAN_and_FM_loss(pred_real, pred_fake)
        return loss_D_T_real, loss_D_T_fake, loss_G_T_GAN, loss_G_T_GAN_Feat

    def GAN_and_FM_loss(self, pred_real, pred_fake):                        # discriminator.py:186-200
        loss_G_GAN = self.criterionGAN(pred_fake, True)
        loss_G_GAN_Feat = torch.zeros_like(loss_G_GAN)
        if not self.opt["no_ganFeat"]:
            feat_weights = 4.0 / (self.opt["n_layers_D"] + 1)
            D_weights = 1.0 / self.opt["num_D"]
            for i in range(min(len(pred_fake), self.opt["num_D"])):
                for j in range(len(pred_fake[i]) - 1):
                    loss_G_GAN_Feat += D_weights * feat_weights * \
                        self._feat(pred_fake[i][j], pred_real[i][j].detach()) * self.opt["lambda_feat"]
        return loss_G_GAN, loss_G_GAN_Feat

    def get_all_skipped_frames(self, frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, flowNet):   # :219-234
        real_B_all, fake_B_all, flow_ref_all, conf_ref_all = frames_all
        real_B_all, real_B_skipped = get_skipped_frames(real_B_all, real_B, t_scales, tD)
        fake_B_all, fake_B_skipped = get_skipped_frames(fake_B_all, fake_B, t_scales, tD)
        flow_ref_all, conf_ref_all, flow_ref_skipped, conf_ref_skipped = get_skipped_flows(
            flowNet, flow_ref_all, conf_ref_all, real_B_skipped, flow_ref, conf_ref, t_scales, tD)
        return (real_B_all, fake_B_all, flow_ref_all, conf_ref_all), (real_B_skipped, fake_B_skipped, flow_ref_skipped, conf_ref_skipped)

    def get_losses(self, loss_dict, loss_dict_T, t_scales):                 # discriminator.py:236-248
        loss_D = (loss_dict["D_fake"] + loss_dict["D_real"]) * 0.5
        loss_G = loss_dict["G_GAN"] + loss_dict["G_GAN_Feat"] + loss_dict["G_VGG"]
        loss_G += loss_dict["G_Warp"] + loss_dict["F_Flow"] + loss_dict["F_Warp"] + loss_dict["W"]
        loss_D_T = []
        t_scales_act = min(t_scales, len(loss_dict_T))
        for s in range(t_scales_act):
            loss_G += loss_dict_T[s]["G_T_GAN"] + loss_dict_T[s]["G_T_GAN_Feat"] + loss_dict_T[s]["G_T_Warp"]
            loss_D_T.append((loss_dict_T[s]["D_T_fake"] + loss_dict_T[s]["D_T_real"]) * 0.5)
        return loss_G, loss_D, loss_D_T, t_scales_act


def get_skipped_frames(B_all, B, t_scales, tD):                             # discriminator.py:257-271
    B_all = torch.cat([B_all.detach(), B], dim=1) if B_all is not None else B
    B_skipped = [None] * t_scales
    for s in range(t_scales):
        tDs = tD ** s
        span = tDs * (tD - 1)
        n_groups = min(B_all.size()[1] - span, B.size()[1])
        if n_groups > 0:
            for t in range(0, n_groups, tD):
                skip = B_all[:, (-span - t - 1):-t:tDs].contiguous() if t != 0 else B_all[:, -span - 1::tDs].contiguous()
                B_skipped[s] = torch.cat([B_skipped[s], skip]) if B_skipped[s] is not None else skip
    max_prev_frames = tD ** (t_scales - 1) * (tD - 1)
    if B_all.size()[1] > max_prev_frames:
        B_all = B_all[:, -max_prev_frames:]
    return B_all, B_skipped


def get_skipped_flows(flowNet, flow_ref_all, conf_ref_all, real_B, flow_ref, conf_ref, t_scales, tD):    # :274-284
    flow_ref_skipped, conf_ref_skipped = [None] * t_scales, [None] * t_scales
    flow_ref_all, flow = get_skipped_frames(flow_ref_all, flow_ref, 1, tD)
    conf_ref_all, conf = get_skipped_frames(conf_ref_all, conf_ref, 1, tD)
    if flow[0] is not None:
        flow_ref_skipped[0], conf_ref_skipped[0] = flow[0][:, 1:], conf[0][:, 1:]
    for s in range(1, t_scales):
        if real_B[s] is not None and real_B[s].size()[1] == tD:
            flow_ref_skipped[s], conf_ref_skipped[s] = flowNet(real_B[s][:, 1:], real_B[s][:, :-1])
    return flow_ref_all, conf_ref_all, flow_ref_skipped, conf_ref_skipped


def reshape(tensors):                                                       # train_vid2vid.py:172-178
    if tensors is None:
        return None
    if isinstance(tensors, list):
        return [reshape(tensor) for tensor in tensors]
    _, _, ch, h, w = tensors.size()
    return tensors.contiguous().view(-1, ch, h, w)


BASE_OPT = dict(input_nc=3, output_nc=3, n_input_gen_frames=3, first_layer_gen_filters=64, gen_network="composite",
                gen_ds_layers=3, norm="batch", n_scales_spatial=1, first_layer_dis_filters=64, num_D=2, n_layers_D=3,
                no_ganFeat=False, n_frames_D=3, n_scales_temporal=2, lr=2e-4, beta1=0.5, lambda_feat=10.0, lambda_T=10.0,
                lambda_F=10.0, no_first_img=False, gan_mode="ls", no_vgg=True, **G_OPT)


def _grad_record(prefix, named_params, keep):
    """Per-parameter gradient L2 norms (all) + full gradients of the tensors named in ``keep``."""
    names, norms, full = [], [], {}
    for k, p in named_params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(k)
        norms.append(g.double().norm().item())
        if k in keep:
            full[f"{prefix}grad/{k}"] = g.numpy().copy()
    return {f"{prefix}grad_names": np.array(names), f"{prefix}grad_norms": np.array(norms), **full}


D_KEEP = ("scale0_layer0.0.weight", "scale0_layer0.0.bias", "scale0_layer2.1.weight", "scale0_layer2.1.bias",
          "scale1_layer1.0.weight", "scale1_layer4.0.weight", "scale1_layer4.0.bias", "scale0_layer3.0.bias")


def loss_case(no_first_img, n_scales_spatial, tag):
    opt = dict(BASE_OPT, no_first_img=no_first_img, n_scales_spatial=n_scales_spatial)
    D = RefModelD(opt, seeds=(51, 52))
    H, W = 64, 128
    t = {}
    for i, k in enumerate(("real_B", "fake_B", "fake_B_raw", "real_A", "real_B_prev", "fake_B_prev")):
        t[k] = smooth((1, 3, H, W), 60 + i)
    t["fake_B"] = (0.7 * t["fake_B"] + 0.3 * t["real_B"]).contiguous()
    t["flow"] = smooth((1, 2, H, W), 70, blur=9) * 4.0
    t["weight"] = torch.sigmoid(smooth((1, 1, H, W), 71) * 2.0)
    t["flow_ref"] = smooth((1, 2, H, W), 72, blur=9) * 4.0
    t["conf_ref"] = (smooth((1, 1, H, W), 73) > -0.3).float()
    leaves = ("fake_B", "fake_B_raw", "flow", "weight")
    for k in leaves:
        t[k].requires_grad_()
    order = ("real_B", "fake_B", "fake_B_raw", "real_A", "real_B_prev", "fake_B_prev", "flow", "weight", "flow_ref", "conf_ref")
    losses = D.forward(0, [t[k] for k in order])
    losses = [torch.mean(x) if x is not None else 0 for x in losses]              # train_vid2vid.py:75
    loss_dict = dict(zip(D.loss_names, losses))
    # temporal scale 0 on a skipped triplet
    tt = dict(real_B_s=smooth((1, 3, 3, H, W), 80), fake_B_s=smooth((1, 3, 3, H, W), 81),
              flow_ref_s=smooth((1, 2, 2, H, W), 82, blur=9) * 4.0, conf_ref_s=(smooth((1, 2, 1, H, W), 83) > -0.3).float())
    tt["fake_B_s"].requires_grad_()
    lossesT = D.forward(1, [tt["real_B_s"], tt["fake_B_s"], tt["flow_ref_s"], tt["conf_ref_s"]])
    lossesT = [torch.mean(x) if not isinstance(x, int) else x for x in lossesT]   # train_vid2vid.py:97
    loss_dict_T = [dict(zip(D.loss_names_T, lossesT))]
    loss_G, loss_D, loss_D_T, _ = D.get_losses(loss_dict, loss_dict_T, opt["n_scales_temporal"])
    out = {f"in/{k}": v.detach().numpy() for k, v in {**t, **tt}.items()}
    out.update({f"loss/{k}": np.float64(v.item()) for k, v in loss_dict.items()})
    out.update({f"loss/{k}": np.float64(v.item()) for k, v in loss_dict_T[0].items()})
    out.update({"loss/G": loss_G.item(), "loss/D": loss_D.item(), "loss/D_T0": loss_D_T[0].item()})
    # loss_backward x3 (train_vid2vid.py:104-111, :166-169); optimizer_g has no parameters here: the generator's pass
    # is represented by the gradients that reach the generated tensors
    loss_G.backward()
    for k in leaves:
        out[f"gradG/{k}"] = (t[k].grad if t[k].grad is not None else torch.zeros_like(t[k])).numpy().copy()
    out["gradG/fake_B_s"] = tt["fake_B_s"].grad.numpy().copy()
    D.optimizer_D.zero_grad()
    loss_D.backward()
    out.update(_grad_record("D/", D.netD.named_parameters(), D_KEEP))
    D.optimizer_D_T[0].zero_grad()
    loss_D_T[0].backward()
    out.update(_grad_record("DT0/", D.netD_T[0].named_parameters(), D_KEEP))
    np.savez_compressed(os.path.join(OUT, f"losses_{tag}.npz"), seeds=np.array([51, 52]), no_first_img=no_first_img,
                        n_scales_spatial=n_scales_spatial, **out)
    print("losses", tag, {k: round(float(v), 5) for k, v in out.items() if k.startswith("loss/")})


# ------------------------------------------------------------------------------------------------
# a8 + a13 + a14: whole windows
# ------------------------------------------------------------------------------------------------
def build_pyr(tensor, n_scales):                                            # base_model.py:64-82
    tensor = [tensor]
    downsample = torch.nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
    for s in range(1, n_scales):
        b, t, c, h, w = tensor[-1].size()
        down = downsample(tensor[-1].view(-1, h, w)).view(b, t, c, h // 2, w // 2)
        tensor.append(down)
    return tensor


class RefModelG:
    """Vid2VidGenerator restated around the reference's generator modules (generator.py:16-80)."""

    def __init__(self, opt, seed):
        self.opt, self.n_scales = opt, opt["n_scales_spatial"]
        g_in = opt["input_nc"] * opt["n_input_gen_frames"]
        prev_nc = (opt["n_input_gen_frames"] - 1) * opt["output_nc"]
        kw = {k: opt[k] for k in G_OPT}
        nets = []
        for mod in (ref, mine):
            torch.manual_seed(seed)
            gs = [mod.build_generator_module(g_in, opt["output_nc"], prev_nc, opt["first_layer_gen_filters"], opt["gen_network"],
                                             opt["gen_ds_layers"], opt["norm"], 0, **kw)]
            for s in range(1, self.n_scales):
                gs.append(mod.build_generator_module(g_in, opt["output_nc"], prev_nc, opt["first_layer_gen_filters"] // (2 ** s),
                                                     opt["gen_network"] + "-local", opt["gen_ds_layers"], opt["norm"], s, **kw))
            nets.append(gs)
        for a, b in zip(*nets):
            assert_same_init(a, b)
        self.netG = [g.train() for g in nets[0]]
        del nets[1]
        self.n_frames_load, self.n_frames_bp = 1, 1                          # generator.py:52-55 (one GPU, max_frames_per_gpu=1)
        params = list(self.netG[self.n_scales - 1].parameters())             # generator.py:67-70 (niter_fix_global = 0)
        for s in range(self.n_scales - 1):
            params += list(self.netG[s].parameters())
        self.optimizer_G = torch.optim.Adam(params, lr=opt["lr"], betas=(opt["beta1"], 0.999))

    emulate = None      # as RefModelD.emulate

    def _net(self, s, A, prev, feat, flow_feat, use_raw_only):
        """netG.forward (networks.py:191-220 / :288-317); the warp + blend of :207-209 / :305-307 applied outside
        because BaseCompositeGeneratorModule.resample calls .cuda() (networks.py:93-100)."""
        if self.emulate is None:
            _, flow, weight, raw, feat, flow_feat, _ = self.netG[s](A, prev, None, feat, flow_feat, None, True)
        else:
            _, flow, weight, raw, feat, flow_feat, _ = emulated.generator_forward(self.netG[s], A, prev, feat, flow_feat, True,
                                                                                  dtype=self.emulate)
        if use_raw_only:
            return raw, flow, weight, raw, feat, flow_feat
        b, _, h, w = raw.shape
        grid = ref.get_grid(b, h, w, device="cpu", dtype=flow.dtype)
        fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
        warp = F.grid_sample(prev[:, -3:], (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border")
        return raw * weight + warp * (1 - weight), flow, weight, raw, feat, flow_feat

    def forward(self, input_A, input_B, fake_B_prev):                       # generator.py:99-123
        tG = self.opt["n_input_gen_frames"]
        real_A_all, real_B_all = input_A, input_B
        self.bs, _, _, self.height, self.width = input_A.size()
        is_first_frame = fake_B_prev is None
        if is_first_frame:                                                   # generate_first_frame, generator.py:217-235
            if self.opt["no_first_img"]:
                fake_B_prev = torch.zeros(self.bs, tG - 1, self.opt["output_nc"], self.height, self.width)
            else:
                fake_B_prev = real_B_all[:, :(tG - 1), ...]
            fake_B_prev = build_pyr(fake_B_prev, self.n_scales)
        fake_B, fake_B_raw, flow, weight = self.generate_frame_train(real_A_all, fake_B_prev, is_first_frame)
        fake_B_prev = [B[:, -tG + 1:].detach() for B in fake_B]
        fake_B = [B[:, tG - 1:] for B in fake_B]
        return fake_B[0], fake_B_raw, flow, weight, real_A_all[:, tG - 1:], real_B_all[:, tG - 2:], fake_B_prev

    def generate_frame_train(self, real_A_all, fake_B_pyr, is_first_frame):   # generator.py:125-182
        tG, n_scales = self.opt["n_input_gen_frames"], self.n_scales
        real_A_pyr = build_pyr(real_A_all, n_scales)
        fake_Bs_raw, flows, weights = None, None, None
        for t in range(self.n_frames_load):
            fake_B_feat = flow_feat = None
            for s in range(n_scales):
                si = n_scales - 1 - s
                real_As = real_A_pyr[si]
                _, _, _, h, w = real_As.size()
                real_As_reshaped = real_As[:, t:t + tG, ...].reshape(self.bs, -1, h, w)
                fake_B_prevs = fake_B_pyr[si][:, t:t + tG - 1, ...]
                if (t % self.n_frames_bp) == 0:
                    fake_B_prevs = fake_B_prevs.detach()
                fake_B_prevs_reshaped = fake_B_prevs.reshape(self.bs, -1, h, w)
                use_raw_only = self.opt["no_first_img"] and is_first_frame
                fake_B, flow, weight, fake_B_raw, fake_B_feat, flow_feat = self._net(
                    s, real_As_reshaped, fake_B_prevs_reshaped, fake_B_feat, flow_feat, use_raw_only)
                fake_B_pyr[si] = concat([fake_B_pyr[si], fake_B.unsqueeze(1)], dim=1)
                if s == n_scales - 1:
                    fake_Bs_raw = concat([fake_Bs_raw, fake_B_raw.unsqueeze(1)], dim=1)
                    if flow is not None:
                        flows = concat([flows, flow.unsqueeze(1)], dim=1)
                        weights = concat([weights, weight.unsqueeze(1)], dim=1)
        return fake_B_pyr, fake_Bs_raw, flows, weights

    @staticmethod
    def compute_fake_B_prev(real_B_prev, fake_B_last, fake_B):              # generator.py:283-287
        fake_B_prev = real_B_prev[:, 0:1] if fake_B_last is None else fake_B_last[0][:, -1:]
        if fake_B.size()[1] > 1:
            fake_B_prev = torch.cat([fake_B_prev, fake_B[:, :-1].detach()], dim=1)
        return fake_B_prev


def loss_backward(loss, optimizer):                                         # train_vid2vid.py:166-169
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()


G_KEEP = ("model_down_seg.1.weight", "model_down_seg.2.weight", "model_down_img.1.bias", "model_final_img.1.weight",
          "model_final_flow.1.weight", "model_final_w.1.weight", "model_final_w.1.bias", "model_up_img.7.bias",
          "model_up_flow.1.weight", "model_down_seg.11.bias")


def run_windows(model_g, model_d, seq_A, seq_B, n_windows, opt, record_grads=True, tag=""):
    """train_vid2vid.py:54-111 over 3-frame windows.  -> per-window records {loss terms, outputs}, window-0 gradients."""
    tG, tD, t_scales = opt["n_input_gen_frames"], opt["n_frames_D"], opt["n_scales_temporal"]
    fake_B_prev_last = None
    frames_all = (None, None, None, None)
    recs, grads = [], {}
    for i in range(n_windows):
        input_A, input_B = seq_A[:, i:i + tG], seq_B[:, i:i + tG]
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = model_g.forward(input_A, input_B, fake_B_prev_last)
        real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
        flow_ref, conf_ref = stub_flow_and_conf(real_B, real_B_prev)
        fake_B_prev = model_g.compute_fake_B_prev(real_B_prev, fake_B_prev_last, fake_B)
        fake_B_prev_last = fake_B_last
        losses = model_d.forward(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight,
                                             flow_ref, conf_ref]))
        losses = [torch.mean(x) if x is not None else 0 for x in losses]
        loss_dict = dict(zip(model_d.loss_names, losses))
        frames_all, frames_skipped = model_d.get_all_skipped_frames(frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales,
                                                                    tD, stub_flow_and_conf)
        loss_dict_T = []
        for s in range(t_scales):
            if frames_skipped[0][s] is not None:
                losses = model_d.forward(s + 1, [frame_skipped[s] for frame_skipped in frames_skipped])
                losses = [torch.mean(x) if not isinstance(x, int) else x for x in losses]
                loss_dict_T.append(dict(zip(model_d.loss_names_T, losses)))
        loss_G, loss_D, loss_D_T, t_scales_act = model_d.get_losses(loss_dict, loss_dict_T, t_scales)
        rec = {k: v.item() for k, v in loss_dict.items()}
        for s, d in enumerate(loss_dict_T):
            rec.update({f"{k}{s}": v.item() for k, v in d.items()})
        rec.update({"G": loss_G.item(), "D": loss_D.item()})
        rec.update({f"D_T{s}": v.item() for s, v in enumerate(loss_D_T)})
        outs = dict(fake_B=fake_B.detach().clone(), fake_B_raw=fake_B_raw.detach().clone(), flow=flow.detach().clone(),
                    weight=weight.detach().clone())
        recs.append((rec, outs))
        # the three loss_backward calls, with the gradients recorded between backward() and step() for window 0
        model_g.optimizer_G.zero_grad()
        loss_G.backward()
        if i == 0 and record_grads:
            for s, g in enumerate(model_g.netG):
                grads[f"G{s}"] = {k: p.grad.clone() for k, p in g.named_parameters() if p.grad is not None}
        model_g.optimizer_G.step()
        model_d.optimizer_D.zero_grad()
        loss_D.backward()
        if i == 0 and record_grads:
            grads["D"] = {k: p.grad.clone() for k, p in model_d.netD.named_parameters() if p.grad is not None}
        model_d.optimizer_D.step()
        for s in range(t_scales_act):
            loss_backward(loss_D_T[s], model_d.optimizer_D_T[s])
        print(f"window {tag} {i}:", {k: round(v, 5) for k, v in rec.items()}, flush=True)
    return recs, grads


def window_case(tag, n_windows, H, W, no_first_img=False, n_scales_spatial=1, ngf=64, seed=90, lr=2e-4, floor_windows=1):
    """One sequence of training windows by the reference's statements, plus -- for the first ``floor_windows`` windows --
    the ROUNDING FLOOR: the same statements evaluated by oracle/emulated.py (fp32 torch operators, values rounded where
    the HIP path stores a half tensor; same seeds, its own optimizers) and compared with the fp32 run, per loss term,
    output and stored gradient tensor (relative error / relative L2 / |projection - 1|).  The GPU test holds the HIP path
    to a multiple of this independently measured floor instead of to a guessed tolerance.  ``lr=0``: the weights stay
    put, so every later window checks the recurrence and the temporal bookkeeping without the feedback of Adam steps
    (whose first step is lr * sign(gradient): one flipped sign moves a weight by 2 lr whatever the gradient's size)."""
    import copy
    opt = dict(BASE_OPT, no_first_img=no_first_img, n_scales_spatial=n_scales_spatial, first_layer_gen_filters=ngf, lr=lr)
    model_g = RefModelG(opt, seed)
    model_d = RefModelD(opt, seeds=(seed + 1, seed + 2))
    tG = opt["n_input_gen_frames"]
    n_frames = n_windows + tG - 1
    seq_A, seq_B = smooth((1, n_frames, 3, H, W), seed + 5), smooth((1, n_frames, 3, H, W), seed + 6)
    out = {"seq_A": seq_A.numpy(), "seq_B": seq_B.numpy()}
    twins = {}
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):     # copies made before anything is stepped
        tg, td = RefModelG.__new__(RefModelG), RefModelD.__new__(RefModelD)
        tg.__dict__.update({k: v for k, v in model_g.__dict__.items() if k not in ("netG", "optimizer_G")})
        td.__dict__.update({k: v for k, v in model_d.__dict__.items() if k not in ("netD", "netD_T", "optimizer_D", "optimizer_D_T")})
        tg.netG = copy.deepcopy(model_g.netG)
        td.netD, td.netD_T = copy.deepcopy(model_d.netD), copy.deepcopy(model_d.netD_T)
        adam = dict(lr=opt["lr"], betas=(opt["beta1"], 0.999))
        params = list(tg.netG[tg.n_scales - 1].parameters())
        for s in range(tg.n_scales - 1):
            params += list(tg.netG[s].parameters())
        tg.optimizer_G = torch.optim.Adam(params, **adam)
        td.optimizer_D = torch.optim.Adam(list(td.netD.parameters()), **adam)
        td.optimizer_D_T = [torch.optim.Adam(list(d.parameters()), **adam) for d in td.netD_T]
        tg.emulate = td.emulate = dt
        twins[name] = (tg, td)
    recs, grads = run_windows(model_g, model_d, seq_A, seq_B, n_windows, opt, tag=tag)
    for i, (rec, outs) in enumerate(recs):
        out.update({f"w{i}/loss/{k}": np.float64(v) for k, v in rec.items()})
        out.update({f"w{i}/{k}": v.numpy().astype(np.float16) for k, v in outs.items()})
    for s, g in enumerate(model_g.netG):
        norms = {k: grads[f"G{s}"][k].double().norm().item() if k in grads[f"G{s}"] else 0.0 for k, _ in g.named_parameters()}
        out[f"w0/G{s}/grad_names"], out[f"w0/G{s}/grad_norms"] = np.array(list(norms)), np.array(list(norms.values()))
        out.update({f"w0/G{s}/grad/{k}": grads[f"G{s}"][k].numpy() for k in G_KEEP if k in grads[f"G{s}"]})
    norms = {k: grads["D"][k].double().norm().item() if k in grads["D"] else 0.0 for k, _ in model_d.netD.named_parameters()}
    out["w0/D/grad_names"], out["w0/D/grad_norms"] = np.array(list(norms)), np.array(list(norms.values()))
    out.update({f"w0/D/grad/{k}": grads["D"][k].numpy() for k in D_KEEP if k in grads["D"]})
    # the rounding floor
    for name, (tg, td) in twins.items():
        erecs, egrads = run_windows(tg, td, seq_A, seq_B, floor_windows, opt, tag=f"{tag} emulated {name}")
        fl = {}
        for i in range(floor_windows):
            for k, v in erecs[i][0].items():
                fl[f"floor/{name}/w{i}/loss/{k}"] = abs(v - recs[i][0][k]) / max(abs(recs[i][0][k]), 0.05)
            for k, v in erecs[i][1].items():
                fl[f"floor/{name}/w{i}/out/{k}"] = ((v - recs[i][1][k]).norm() / recs[i][1][k].norm()).item()
        for prefix, keep in [(f"G{s}", G_KEEP) for s in range(len(model_g.netG))] + [("D", D_KEEP)]:
            got, want = egrads[prefix], grads[prefix]
            for k in keep:
                if k in want and k in got and want[k].norm() > 0:
                    a, b = got[k].double().flatten(), want[k].double().flatten()
                    fl[f"floor/{name}/{prefix}/l2/{k}"] = ((a - b).norm() / b.norm()).item()
                    fl[f"floor/{name}/{prefix}/proj/{k}"] = abs((a @ b / (b @ b)).item() - 1.0)
            tot_a = torch.sqrt(sum(v.double().pow(2).sum() for v in got.values()))
            tot_b = torch.sqrt(sum(v.double().pow(2).sum() for v in want.values()))
            fl[f"floor/{name}/{prefix}/total_norm"] = abs(tot_a / tot_b - 1).item()
        print("rounding floor", tag, name, {k.split("/", 2)[2]: round(v, 4) for k, v in fl.items() if "/out/" in k or "/loss/G" in k}, flush=True)
        out.update({k: np.float64(v) for k, v in fl.items()})
    # what the optimizers did to a few tensors (Adam, beta1 0.5)
    sd = model_g.netG[-1].state_dict()
    for k in ("model_final_img.1.weight", "model_final_img.1.bias", "model_down_seg.2.weight"):
        out[f"after/G/{k}"] = sd[k].numpy().copy()
    sd = model_d.netD.state_dict()
    for k in ("scale0_layer0.0.bias", "scale1_layer4.0.weight"):
        out[f"after/D/{k}"] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"window_{tag}.npz"), seed=seed, n_windows=n_windows, ngf=ngf, lr=lr,
                        floor_windows=floor_windows, no_first_img=no_first_img, n_scales_spatial=n_scales_spatial, **out)


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"flownet", "losses", "window"}
    torch.set_num_threads(8)
    if "flownet" in what:
        flownet_cases()
    if "losses" in what:
        loss_case(False, 1, "s1")
        loss_case(True, 2, "s2_nofirst")
    if "window" in what:
        window_case("ngf64_64x128_lr0", 8, 64, 128, lr=0.0, floor_windows=3)               # recurrence + temporal bookkeeping, both temporal scales
        window_case("ngf64_64x128", 3, 64, 128, floor_windows=3)            # with the three Adam steps per window
        # no window golden with no_first_img: its first window feeds all-zero previous frames to model_down_img, whose
        # BatchNorm layers then normalise a constant (variance 0, scale 1/sqrt(eps) = 316): in fp32 the result is the
        # amplified rounding residue of mean(bias) - bias, re-shaped by the zero padding of the next convolution and
        # normalised to O(1) again -- not a function of the weights that another summation order reproduces.  The
        # no_first_img terms of the objective (weight loss, lambda_F / 2^(scales-1)) are pinned by losses_s2_nofirst.
        window_case("2scale_ngf128_64x128", 2, 64, 128, n_scales_spatial=2, ngf=128, seed=97, floor_windows=2)
    print("ok")
