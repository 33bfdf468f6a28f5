"""trunc_exp(x): exp(x) computed in fp32 whose gradient is taken at x clamped to [-15, 15]
(semantics of reconstruction/activation.py:5-17; the fused field kernels implement the same rule, field_bwd.hip).

exp is monotone, so exp(clamp(x, -15, 15)) == clamp(exp(x), e^-15, e^15): the backward needs only the forward's
OUTPUT, which is what is kept for it (no second exponential, no copy of the input).
"""
import math

import torch

_LO, _HI = math.exp(-15.0), math.exp(15.0)


class TruncExp(torch.autograd.Function):
    """y = exp(float32(x));  dL/dx = dL/dy * clamp(y, e^-15, e^15)."""

    @staticmethod
    def forward(ctx, x):
        y = torch.exp(x.float())
        ctx.save_for_backward(y)
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        return (grad_y * y.clamp(_LO, _HI)).to(ctx.in_dtype)


def trunc_exp(x):
    return TruncExp.apply(x)
