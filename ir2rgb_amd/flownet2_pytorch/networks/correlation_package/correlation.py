"""``Correlation`` / ``CorrelationFunction`` -- cost-volume operator backed by libir2rgb_hip.so.

Mirrors the public surface of the reference's
models/flownet2_pytorch/networks/correlation_package/correlation.py:7-55 (same class names,
constructor defaults and positional argument order) so FlowNetC (FlowNetC.py:31,87-89) can
import it unchanged.  Differences, all deliberate:
  * scalar hyper-parameters live on ``ctx`` as attributes (the reference hands ints to
    ``save_for_backward``, correlation.py:13, which raises on current torch);
  * inputs are validated (device, dtype, contiguity) instead of being reinterpreted;
  * no padded channels-last scratch tensors are allocated.
"""
import torch
from torch.autograd import Function
from torch.nn import Module

from ....ext import correlation_cuda


class CorrelationFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1):
        input1, input2 = input1.contiguous(), input2.contiguous()
        ctx.save_for_backward(input1, input2)
        ctx.hyper = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        scratch1, scratch2, output = input1.new_empty(0), input2.new_empty(0), input1.new_empty(0)
        correlation_cuda.forward(input1, input2, scratch1, scratch2, output, *ctx.hyper)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        scratch1, scratch2 = input1.new_empty(0), input2.new_empty(0)
        grad1, grad2 = input1.new_empty(0), input2.new_empty(0)
        correlation_cuda.backward(input1, input2, scratch1, scratch2, grad_output.contiguous(), grad1, grad2,
                                  *ctx.hyper)
        return (grad1, grad2) + (None,) * 6


class Correlation(Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                         self.stride1, self.stride2, self.corr_multiply)

    def extra_repr(self):
        return (f"pad_size={self.pad_size}, kernel_size={self.kernel_size}, max_displacement={self.max_displacement}, "
                f"stride1={self.stride1}, stride2={self.stride2}")
