"""``Resample2d`` / ``Resample2dFunction`` -- pixel-space flow warp backed by libir2rgb_hip.so.

Mirrors reference models/flownet2_pytorch/networks/resample2d_package/resample2d.py:5-46:
``Resample2d(kernel_size=1)(input1, input2)`` makes ``input1`` contiguous and requires
``input2`` (the flow, in pixels) to be contiguous already (the reference asserts, :9-10;
here a ValueError).  fp32 only, like the reference kernel (resample2d_kernel.cu:209-226).
"""
import torch
from torch.autograd import Function
from torch.nn import Module

from ....ext import resample2d_cuda


class Resample2dFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1):
        if not input1.is_contiguous() or not input2.is_contiguous():
            raise ValueError("Resample2dFunction: input1 and input2 must be contiguous")
        ctx.save_for_backward(input1, input2)
        ctx.kernel_size = kernel_size
        channels = input1.size(1)
        b, _, h, w = input2.size()
        output = input1.new_zeros(b, channels, h, w)
        resample2d_cuda.forward(input1, input2, output, kernel_size)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        input1, input2 = ctx.saved_tensors
        grad_input1 = torch.zeros_like(input1)
        grad_input2 = torch.zeros_like(input2)
        resample2d_cuda.backward(input1, input2, grad_output, grad_input1, grad_input2, ctx.kernel_size)
        return grad_input1, grad_input2, None


class Resample2d(Module):
    def __init__(self, kernel_size=1):
        super().__init__()
        self.kernel_size = kernel_size

    def forward(self, input1, input2):
        return Resample2dFunction.apply(input1.contiguous(), input2, self.kernel_size)
