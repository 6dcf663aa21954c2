"""CPU tests: the plain-torch restatement (oracle/networks_oracle.py) against golden outputs of the
reference's own models/networks.py.  fp32 vs fp32: max|delta| <= 1e-5 (feature maps are stored as
fp16 in the fixtures, so those are compared at fp16 resolution)."""
import os

import numpy as np
import pytest
import torch

from ir2rgb_amd import networks as N
from oracle import networks_oracle as O

OPT = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)


@pytest.mark.parametrize("name", ["G0_ngf64_32x64", "G1_ngf64_32x64"])
def test_generator_restatement(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
    torch.manual_seed(int(g["seed"]))
    m = N.build_generator_module(9, 3, 6, int(g["ngf"]), str(g["model_name"]), 3, "batch", int(g["scale"]), **OPT).train()
    A, prev = torch.from_numpy(g["A"]), torch.from_numpy(g["prev"])
    fi = torch.from_numpy(g["img_feat_coarse"]) if "img_feat_coarse" in g else None
    ff = torch.from_numpy(g["flow_feat_coarse"]) if "flow_feat_coarse" in g else None
    with torch.no_grad():
        final, flow, weight, raw, img_feat, flow_feat, _ = O.generator_forward(m, A, prev, fi, ff)
    for got, key in ((final, "img_final"), (flow, "flow"), (weight, "weight"), (raw, "img_raw")):
        np.testing.assert_allclose(got.numpy(), g[key], atol=1e-5 * max(1.0, np.abs(g[key]).max()), rtol=0, err_msg=key)
    for got, key in ((img_feat, "img_feat"), (flow_feat, "flow_feat")):
        np.testing.assert_allclose(got.numpy(), g[key].astype(np.float32), atol=2e-3, rtol=2e-3, err_msg=key)
    np.testing.assert_allclose(m.state_dict()["model_down_seg.2.running_mean"].numpy(), g["running_mean_after"], atol=1e-6)


@pytest.mark.parametrize("name", ["D_nc6_64x96", "DT_nc13_48x80"])
def test_discriminator_restatement(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
    torch.manual_seed(int(g["seed"]))
    d = N.build_discriminator_module(int(g["input_nc"]), 64, 3, "batch", 2, True).train()
    with torch.no_grad():
        out = O.discriminator_forward(d, torch.from_numpy(g["x"]))
    for i, sc in enumerate(out):
        for j, o in enumerate(sc):
            ref = g[f"out{i}_{j}"]
            tol = 1e-5 if ref.dtype == np.float32 else 2e-3
            np.testing.assert_allclose(o.numpy(), ref.astype(np.float32), atol=tol, rtol=tol)


@pytest.mark.parametrize("name", ["G0_ngf64_32x64", "G1_ngf64_32x64"])
def test_emulated_oracle_is_the_same_graph(golden_dir, name):
    """oracle/emulated.py with dtype=float32 (every rounding a no-op) must reproduce the reference goldens: this pins its
    graph (separable heads, bias-free convolutions in front of BatchNorm, fused residual adds) to the reference.  With
    dtype=bfloat16 it must land at the distance from the fp32 golden that half-precision storage costs (1e-3 .. 4e-2)."""
    from oracle import emulated as E
    g = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
    torch.manual_seed(int(g["seed"]))
    m = N.build_generator_module(9, 3, 6, int(g["ngf"]), str(g["model_name"]), 3, "batch", int(g["scale"]), **OPT).train()
    A, prev = torch.from_numpy(g["A"]), torch.from_numpy(g["prev"])
    fi = torch.from_numpy(g["img_feat_coarse"]) if "img_feat_coarse" in g else None
    ff = torch.from_numpy(g["flow_feat_coarse"]) if "flow_feat_coarse" in g else None
    with torch.no_grad():
        out = E.generator_forward(m, A, prev, fi, ff, dtype=torch.float32)
        out_bf = E.generator_forward(m, A, prev, fi, ff, dtype=torch.bfloat16)
    for i, key in ((0, "img_final"), (1, "flow"), (2, "weight"), (3, "img_raw")):
        ref = torch.from_numpy(g[key])
        assert ((out[i] - ref).norm() / ref.norm()).item() <= 5e-5, key
        if i:
            e = ((out_bf[i] - ref).norm() / ref.norm()).item()
            assert 1e-3 <= e <= 4e-2, (key, e)


def test_emulated_discriminator_is_the_same_graph(golden_dir):
    from oracle import emulated as E
    g = np.load(os.path.join(golden_dir, "net_D_nc6_64x96.npz"))
    torch.manual_seed(int(g["seed"]))
    d = N.build_discriminator_module(6, 64, 3, "batch", 2, True).train()
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        out, want = E.discriminator_forward(d, x, torch.float32), O.discriminator_forward(d, x)
    for a, b in zip(out, want):
        for u, v in zip(a, b):
            assert ((u - v).norm() / v.norm()).item() <= 2e-5
