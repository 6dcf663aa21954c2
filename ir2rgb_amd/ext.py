"""Stand-ins for the reference's three pybind11 extension modules.

``correlation_cuda``, ``resample2d_cuda`` and ``channelnorm_cuda`` expose ``forward`` /
``backward`` with the argument lists of the reference bindings
(correlation_cuda.cc:10-20,:89-97,:169-172; resample2d_cuda.cc:6-31; channelnorm_cuda.cc:6-30)
so that the reference's own ``*_package/*.py`` files work against them unchanged (see
INTEGRATION.md).  They marshal ``at::Tensor`` arguments into the C ABI of
include/ir2rgb_hip.h: raw device pointers, sizes, and the current HIP stream.
"""
import ctypes

import torch

from . import _lib

_f32 = torch.float32


def _p(t):

    return t.data_ptr()


class correlation_cuda:
    @staticmethod
    def out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
        oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().ir2rgb_correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                                           stride2, ctypes.byref(oc), ctypes.byref(oh),
                                                           ctypes.byref(ow)), "correlation_out_shape")
        return oc.value, oh.value, ow.value

    @staticmethod
    def forward(input1, input2, rInput1, rInput2, output, pad_size, kernel_size, max_displacement, stride1, stride2,
                corr_type_multiply):
        """output is resized to [N, D*D, outH, outW] and fully written.  rInput1/rInput2 (the
        reference's padded channels-last scratch) are accepted and left empty: this
        implementation needs no scratch."""
        _lib.require_device(input1, input2, dtype=_f32)
        if input1.dim() != 4 or input1.shape != input2.shape:
            raise ValueError(f"correlation: expected two equal 4-d inputs, got {tuple(input1.shape)} and "
                             f"{tuple(input2.shape)}")
        N, C, H, W = input1.shape
        oc, oh, ow = correlation_cuda.out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
        if oh <= 0 or ow <= 0:
            raise ValueError("correlation: empty output for these parameters")
        output.resize_(N, oc, oh, ow)
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_correlation_fwd(input1, input2, output, N, C, H, W, pad_size,
                                                   kernel_size, max_displacement, stride1, stride2,
                                                   _lib.current_stream(input1))
        _lib.check(rc, "correlation_cuda.forward")
        return 1

    @staticmethod
    def backward(input1, input2, rInput1, rInput2, gradOutput, gradInput1, gradInput2, pad_size, kernel_size,
                 max_displacement, stride1, stride2, corr_type_multiply):
        _lib.require_device(input1, input2, dtype=_f32)
        gradOutput = gradOutput.contiguous()
        _lib.require_device(gradOutput, dtype=_f32)
        N, C, H, W = input1.shape
        oc, oh, ow = correlation_cuda.out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
        if tuple(gradOutput.shape) != (N, oc, oh, ow):
            raise ValueError(f"correlation backward: gradOutput {tuple(gradOutput.shape)} != {(N, oc, oh, ow)}")
        gradInput1.resize_(N, C, H, W)
        gradInput2.resize_(N, C, H, W)
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_correlation_bwd(input1, input2, gradOutput, gradInput1,
                                                   gradInput2, N, C, H, W, pad_size, kernel_size,
                                                   max_displacement, stride1, stride2, _lib.current_stream(input1))
        _lib.check(rc, "correlation_cuda.backward")
        return 1


class resample2d_cuda:
    @staticmethod
    def _shapes(input1, input2):
        _lib.require_device(input1, input2, dtype=_f32)
        if input1.dim() != 4 or input2.dim() != 4 or input2.shape[1] != 2:
            raise ValueError("resample2d: expected img [N,C,H,W] and flow [N,2,H,W]")
        N, C, H, W = input1.shape
        if tuple(input2.shape) != (N, 2, H, W):
            raise ValueError(f"resample2d: flow {tuple(input2.shape)} does not match image {tuple(input1.shape)}")
        return N, C, H, W

    @staticmethod
    def forward(input1, input2, output, kernel_size):
        N, C, H, W = resample2d_cuda._shapes(input1, input2)
        _lib.require_device(output, dtype=_f32)
        if tuple(output.shape) != (N, C, H, W):
            raise ValueError("resample2d: output has the wrong shape")
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_resample2d_fwd(input1, input2, output, N, C, H, W, kernel_size,
                                                  _lib.current_stream(input1))
        _lib.check(rc, "resample2d_cuda.forward")
        return 1

    @staticmethod
    def backward(input1, input2, gradOutput, gradInput1, gradInput2, kernel_size):
        N, C, H, W = resample2d_cuda._shapes(input1, input2)
        _lib.require_device(gradOutput, gradInput1, gradInput2, dtype=_f32)
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_resample2d_bwd(input1, input2, gradOutput, gradInput1,
                                                  gradInput2, N, C, H, W, kernel_size,
                                                  _lib.current_stream(input1))
        _lib.check(rc, "resample2d_cuda.backward")
        return 1


class channelnorm_cuda:
    @staticmethod
    def forward(input1, output, norm_deg):
        _lib.require_device(input1, output, dtype=_f32)
        N, C, H, W = input1.shape
        if tuple(output.shape) != (N, 1, H, W):
            raise ValueError("channelnorm: output has the wrong shape")
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_channelnorm_fwd(input1, output, N, C, H, W, norm_deg,
                                                   _lib.current_stream(input1))
        _lib.check(rc, "channelnorm_cuda.forward")
        return 1

    @staticmethod
    def backward(input1, output, gradOutput, gradInput1, norm_deg):
        _lib.require_device(input1, output, gradOutput, gradInput1, dtype=_f32)
        N, C, H, W = input1.shape
        with _lib.on_device(input1):
            rc = _lib.lib().ir2rgb_channelnorm_bwd(input1, output, gradOutput, gradInput1, N, C, H, W,
                                                   norm_deg, _lib.current_stream(input1))
        _lib.check(rc, "channelnorm_cuda.backward")
        return 1


def warp_diff_norm(img1, img2, flow, want_warped=True, want_diff=True, want_norm=True):
    """Fused ``w = Resample2d(img2, flow); d = img1 - w; n = ChannelNorm(d)`` -- one launch for
    the three-launch sequence of reference models/flownet2_pytorch/models.py:109-111."""
    _lib.require_device(img1, img2, flow, dtype=_f32)
    N, C, H, W = img2.shape
    if tuple(img1.shape) != (N, C, H, W) or tuple(flow.shape) != (N, 2, H, W):
        raise ValueError("warp_diff_norm: shape mismatch")
    warped = torch.empty_like(img2) if want_warped else None
    diff = torch.empty_like(img2) if want_diff else None
    norm = img2.new_empty(N, 1, H, W) if want_norm else None
    null = 0
    with _lib.on_device(img1):
        rc = _lib.lib().ir2rgb_warp_diff_norm_fwd(img1, img2, flow, warped if want_warped else null,
                                                  diff if want_diff else null, norm if want_norm else null,
                                                  N, C, H, W, _lib.current_stream(img1))
    _lib.check(rc, "warp_diff_norm")
    return warped, diff, norm
