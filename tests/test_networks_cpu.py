"""CPU tests of the factory boundary: signatures, state_dict keys, init parity with the reference
(through checksums frozen by tests/golden/make_net_goldens.py), loud failure without a GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from ir2rgb_amd import networks as N

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


def _check_init(module, g):
    sd = module.state_dict()
    keys = [str(k) for k in g["check_keys"]]
    assert [k for k in sd if sd[k].dtype.is_floating_point] == keys
    for k, (s, a) in zip(keys, g["check_vals"]):
        v = sd[k].double()
        assert abs(v.sum().item() - s) <= 1e-9 * max(1.0, abs(a)), k
        assert abs(v.abs().sum().item() - a) <= 1e-9 * max(1.0, abs(a)), k


def test_generator_init_matches_reference(golden_dir):
    for f in sorted(glob.glob(os.path.join(golden_dir, "net_G*.npz"))):
        g = np.load(f)
        torch.manual_seed(int(g["seed"]))
        if "model_name" in g:
            m = N.build_generator_module(9, 3, 6, int(g["ngf"]), str(g["model_name"]), 3, "batch", int(g["scale"]), **OPT)
        else:       # the 'global' generator of BASELINE config 1
            m = N.build_generator_module(3, 3, 0, int(g["ngf"]), "global", 3, "batch", 0, **OPT)
        _check_init(m, g)


def test_discriminator_init_matches_reference(golden_dir):
    for f in sorted(glob.glob(os.path.join(golden_dir, "net_D*.npz"))):
        g = np.load(f)
        ndf, num_D = (int(g["ndf"]), int(g["num_D"])) if "ndf" in g else (64, 2)
        torch.manual_seed(int(g["seed"]))
        _check_init(N.build_discriminator_module(int(g["input_nc"]), ndf, 3, "batch", num_D, True), g)


def test_generator_state_dict_keys():
    m = N.build_generator_module(9, 3, 6, 64, "composite", 3, "batch", 0, **OPT)
    keys = set(m.state_dict())
    # conv/BN pairs at 1,2,4,5,7,8,10,11; ResnetBlocks 13..17 with conv_block 1,2,5,6
    for i in (1, 4, 7, 10):
        assert f"model_down_seg.{i}.weight" in keys and f"model_down_seg.{i + 1}.running_var" in keys
        assert f"model_down_img.{i}.bias" in keys
    for b in range(13, 18):
        for j in (1, 2, 5, 6):
            assert f"model_down_seg.{b}.conv_block.{j}.weight" in keys
    for b in range(4):
        assert f"model_res_img.{b}.conv_block.5.bias" in keys and f"model_res_flow.{b}.conv_block.1.weight" in keys
    for i in (0, 1, 3, 4, 6, 7):
        assert f"model_up_img.{i}.weight" in keys and f"model_up_flow.{i}.weight" in keys
    for k in ("model_final_img.1.weight", "model_final_flow.1.bias", "model_final_w.1.weight"):
        assert k in keys
    assert m.model_final_flow[1].out_channels == 2 and m.model_final_w[1].out_channels == 1
    # deep copies are independent parameters; weights_init (applied after construction, reference
    # networks.py:74) redraws them, so they differ at init
    a, b = m.model_down_seg[4].weight, m.model_down_img[4].weight
    assert a is not b and a.shape == b.shape and not torch.equal(a, b)


def test_discriminator_state_dict_keys_and_widths():
    d = N.build_discriminator_module(6, 64, 3, "batch", 2, True)
    keys = set(d.state_dict())
    for i in range(2):
        for j in range(5):
            assert f"scale{i}_layer{j}.0.weight" in keys
        for j in (1, 2, 3):
            assert f"scale{i}_layer{j}.1.running_mean" in keys
    assert d.scale0_layer0[0].weight.shape == (64, 6, 4, 4) and d.scale0_layer4[0].weight.shape == (1, 512, 4, 4)
    assert sum(p.numel() for p in d.parameters()) == 5539202
    d2 = N.build_discriminator_module(6, 64, 3, "batch", 1, False)
    assert "layer0.0.weight" in d2.state_dict()


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        N.build_generator_module(9, 3, 6, 64, "composite", 3, "instance", 0, **OPT)
    with pytest.raises(NotImplementedError):
        N.build_generator_module(9, 3, 6, 64, "nonsense", 3, "batch", 0, **OPT)
    with pytest.raises(NotImplementedError):
        N.build_generator_module(9, 3, 6, 64, "local", 3, "batch", 0, **OPT)      # pix2pixHD LocalEnhancer: not on the path


def test_any_width_builds_with_reference_shapes():
    """generator.py:36 halves ngf per spatial scale, networks.py:634-637 scales ndf per discriminator: no 64-multiple rule."""
    g = N.build_generator_module(9, 3, 6, 48, "composite", 3, "batch", 0, **OPT)
    assert g.model_down_seg[1].weight.shape == (48, 9, 7, 7) and g.model_res_img[0].conv_block[1].weight.shape == (384, 384, 3, 3)
    gl = N.build_generator_module(3, 3, 0, 128, "global", 3, "batch", 0, **OPT)
    assert list(gl.state_dict())[0] == "model.1.weight" and gl.model[-2].weight.shape == (3, 128, 7, 7)
    assert sum(p.numel() for p in gl.parameters()) == 182356995
    d = N.build_discriminator_module(6, 16, 3, "batch", 3, True)
    assert [getattr(d, f"scale{i}_layer0")[0].out_channels for i in range(3)] == [64, 32, 16]


def test_cpu_forward_is_refused():
    m = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **OPT)
    with pytest.raises(ValueError, match="no CPU fallback"):
        m(torch.zeros(1, 9, 16, 16), torch.zeros(1, 6, 16, 16), None, None, None, None, False)
    d = N.build_discriminator_module(6, 64, 3, "batch", 2, True)
    with pytest.raises(ValueError, match="no CPU fallback"):
        d(torch.zeros(1, 6, 32, 32))


def test_get_grid_matches_linspace_lattice():
    g = N.get_grid(2, 5, 7, device="cpu")
    assert g.shape == (2, 2, 5, 7)
    assert torch.equal(g[0, 0, 0], torch.linspace(-1, 1, 7)) and torch.equal(g[1, 1, :, 3], torch.linspace(-1, 1, 5))
