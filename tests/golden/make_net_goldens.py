"""Generates tests/golden/net_*.npz from the REFERENCE's own network definitions.

Imports /root/reference/models/networks.py (importable in the authoring container; it never
travels to the GPU box), builds the generator / discriminator with a fixed seed, runs them on CPU
in fp32 (training-mode BatchNorm, as in the reference's training loop) and stores inputs and
outputs.  Weights are NOT stored: ir2rgb_amd.networks reproduces the reference's construction
order, so the same seed gives bit-identical parameters -- this script asserts that equality
tensor by tensor and stores per-tensor checksums for the CPU test-suite.

The reference's warp (networks.py:93-100) calls .cuda() and cannot run on CPU; img_final is
therefore produced with use_raw_only=True for the convolutional part and the a7 formula
(get_grid + grid_sample, same arithmetic) applied outside the module, as SURVEY section 8c says.

    python tests/golden/make_net_goldens.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from models import networks as ref  # noqa: E402  (the reference)

from ir2rgb_amd import networks as mine  # noqa: E402

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


def checks(sd):
    return {k: np.array([v.double().sum().item(), v.double().abs().sum().item()]) for k, v in sd.items()
            if v.dtype.is_floating_point}


def assert_same_init(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys()), "state_dict keys differ"
    for k in sa:
        assert torch.equal(sa[k], sb[k]), f"init differs at {k}"


def smooth(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    x = F.avg_pool2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), 7, stride=1)
    return torch.tanh(x * 3)


def warp_blend_ref(raw, prev, flow, weight):
    b, c, h, w = raw.shape
    grid = ref.get_grid(b, h, w, device="cpu", dtype=flow.dtype)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    warp = F.grid_sample(prev[:, -3:], (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border")
    return raw * weight + warp * (1 - weight), warp


def gen_case(name, model_name, ngf, H, W, seed, scale=0):
    torch.manual_seed(seed)
    g_ref = ref.build_generator_module(9, 3, 6, ngf, model_name, 3, "batch", scale, **OPT)
    torch.manual_seed(seed)
    g_mine = mine.build_generator_module(9, 3, 6, ngf, model_name, 3, "batch", scale, **OPT)
    assert_same_init(g_ref, g_mine)
    A, prev = smooth((1, 9, H, W), seed + 1), smooth((1, 6, H, W), seed + 2)
    extra = {}
    args = [A, prev, None, None, None, None, True]
    if model_name == "composite-local":
        fi = smooth((1, ngf * 2, H // 2, W // 2), seed + 3).abs()
        ff = smooth((1, ngf * 2, H // 2, W // 2), seed + 4).abs()
        args[3], args[4] = fi, ff
        extra = dict(img_feat_coarse=fi.numpy(), flow_feat_coarse=ff.numpy())
    g_ref.train()
    with torch.no_grad():
        _, flow, weight, raw, img_feat, flow_feat, _ = g_ref(*args)
        final, warp = warp_blend_ref(raw, prev, flow, weight)
    sd = g_ref.state_dict()
    bn_key = "model_down_seg.2.running_mean"
    ck = checks(g_mine.state_dict())
    np.savez_compressed(
        os.path.join(OUT, f"net_{name}.npz"), seed=seed, ngf=ngf, scale=scale, model_name=model_name, A=A.numpy(),
        prev=prev.numpy(), img_final=final.numpy(), img_warp=warp.numpy(), flow=flow.numpy(), weight=weight.numpy(),
        img_raw=raw.numpy(), img_feat=img_feat.numpy().astype(np.float16), flow_feat=flow_feat.numpy().astype(np.float16),
        running_mean_after=sd[bn_key].numpy(), running_var_after=sd["model_down_seg.2.running_var"].numpy(),
        check_keys=np.array(list(ck.keys())), check_vals=np.stack(list(ck.values())), **extra)
    print(name, "params", sum(p.numel() for p in g_ref.parameters()))


def global_case(name, ngf, H, W, seed):
    """BASELINE config 1: the pix2pixHD global generator (networks.py:320-352) on a single 3-channel frame."""
    torch.manual_seed(seed)
    g_ref = ref.build_generator_module(3, 3, 0, ngf, "global", 3, "batch", 0, **OPT)
    torch.manual_seed(seed)
    g_mine = mine.build_generator_module(3, 3, 0, ngf, "global", 3, "batch", 0, **OPT)
    assert_same_init(g_ref, g_mine)
    x = smooth((1, 3, H, W), seed + 1)
    g_ref.train()
    with torch.no_grad():
        out = g_ref(x)
    ck = checks(g_mine.state_dict())
    np.savez_compressed(os.path.join(OUT, f"net_{name}.npz"), seed=seed, ngf=ngf, x=x.numpy(), out=out.numpy(),
                        check_keys=np.array(list(ck.keys())), check_vals=np.stack(list(ck.values())))
    print(name, "params", sum(p.numel() for p in g_ref.parameters()))


def dis_case(name, input_nc, H, W, seed, ndf=64, num_D=2):
    torch.manual_seed(seed)
    d_ref = ref.build_discriminator_module(input_nc, ndf, 3, "batch", num_D, True)
    torch.manual_seed(seed)
    d_mine = mine.build_discriminator_module(input_nc, ndf, 3, "batch", num_D, True)
    assert_same_init(d_ref, d_mine)
    x = smooth((2, input_nc, H, W), seed + 1)
    d_ref.train()
    with torch.no_grad():
        out = d_ref(x)
    ck = checks(d_mine.state_dict())
    arrs = {f"out{i}_{j}": (o.numpy().astype(np.float16) if o.shape[1] > 1 else o.numpy())
            for i, sc in enumerate(out) for j, o in enumerate(sc)}
    np.savez_compressed(os.path.join(OUT, f"net_{name}.npz"), seed=seed, input_nc=input_nc, ndf=ndf, num_D=num_D, x=x.numpy(),
                        check_keys=np.array(list(ck.keys())), check_vals=np.stack(list(ck.values())), **arrs)
    print(name, "params", sum(p.numel() for p in d_ref.parameters()))


if __name__ == "__main__":
    import sys
    new_only = "new" in sys.argv[1:]          # round-2 additions only (the round-1 files stay byte-identical)
    if not new_only:
        gen_case("G0_ngf64_32x64", "composite", 64, 32, 64, seed=11)
        gen_case("G1_ngf64_32x64", "composite-local", 64, 32, 64, seed=12, scale=1)
        dis_case("D_nc6_64x96", 6, 64, 96, seed=13)
        dis_case("DT_nc13_48x80", 13, 48, 80, seed=14)
    # widths that are not multiples of 64 (generator.py:36 halves ngf per spatial scale: n_scales_spatial=3 gives 32, 16)
    gen_case("G0_ngf32_32x64", "composite", 32, 32, 64, seed=15)
    gen_case("G2_ngf16_32x64", "composite-local", 16, 32, 64, seed=16, scale=2)
    dis_case("D_ndf32_nc6_64x96", 6, 64, 96, seed=17, ndf=16, num_D=3)
    global_case("GG_ngf128_256x256", 128, 256, 256, seed=18)      # BASELINE.json configs[0]
    print("ok")
