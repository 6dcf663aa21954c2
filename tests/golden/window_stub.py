"""Deterministic stand-ins shared by tests/golden/make_window_goldens.py (which runs beside the REFERENCE in
the authoring container) and the GPU parity tests (which run beside the HIP build) -- TEST INFRASTRUCTURE.

They replace the two pieces of the training window whose reference code cannot be executed offline:

* ``stub_flow_and_conf``  stands in for ``FlowNet.forward`` (models/flownet.py:20-37): the pretrained
  FlowNet2 checkpoint is a download and FlowNetC needs the CUDA correlation extension.  Any fixed function
  of the two frame stacks serves to pin the HARNESS (what is fed where, which tensors are detached, how the
  losses are weighted); this one is cheap, smooth and data dependent.
* ``stub_flownetc``       stands in for ``FlowNetC.forward`` inside the FlowNet2 composition
  (models/flownet2_pytorch/models.py:104): same reason.

Plain torch, device agnostic, no randomness.
"""
import torch
import torch.nn.functional as F


def smooth(shape, seed, blur=7, gain=3.0):
    """Seeded smooth field in (-1, 1) (Gaussian noise, box blur, tanh) -- same recipe as make_net_goldens."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    lead = x.shape[:-3]
    x = x.reshape((-1,) + tuple(x.shape[-3:]))
    p = blur // 2
    x = F.avg_pool2d(F.pad(x, (p, p, p, p), mode="reflect"), blur, stride=1)
    return torch.tanh(x * gain).reshape(lead + tuple(x.shape[-3:]))


def _flow_conf_4d(im1, im2):
    d = im1 - im2
    flow = 6.0 * F.avg_pool2d(F.pad(d[:, 0:2], (2, 2, 2, 2), mode="replicate"), 5, stride=1)
    conf = ((d * d).sum(1, keepdim=True) < 0.35).float()
    return flow, conf


def stub_flow_and_conf(input_A, input_B):
    """[B,n,3,H,W] or [N,3,H,W] frame stacks -> (flow [..,2,H,W], conf [..,1,H,W]), fp32, no grad."""
    with torch.no_grad():
        if input_A.dim() == 5:
            b, n, c, h, w = input_A.shape
            flow, conf = _flow_conf_4d(input_A.reshape(-1, c, h, w).float(), input_B.reshape(-1, c, h, w).float())
            return flow.view(b, n, 2, h, w), conf.view(b, n, 1, h, w)
        return _flow_conf_4d(input_A.float(), input_B.float())


def stub_flownetc(x):
    """[N,6,H,W] (two mean-subtracted images) -> 'flow2' [N,2,H/4,W/4]."""
    return 0.05 * F.avg_pool2d(x[:, 0:2] - x[:, 3:5], 4)
