"""Mirror of aux_libs/shencoder/sphere_harmonics.py (reference) on libtrinerflet_hip.so.

SHEncoder(input_dim=3, degree=4).forward(inputs, size=1) -> [..., degree^2]; degrees 1..8 as in the
reference (the hot path uses 4, reconstruction/nerf/network.py:58).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib as L


class _sh_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        # reference: sphere_harmonics.py:13-38 (custom_fwd casts to float32)
        L.require_cuda(inputs)
        inputs = inputs.to(torch.float32).contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * output_dim, dtype=torch.float32, device=inputs.device) \
            if calc_grad_inputs else None
        err = L.lib().tnl_sh_encode_forward(L.ptr(inputs), L.ptr(outputs), L.u32(B), L.u32(input_dim), L.u32(degree),
                                            L.ptr(dy_dx), L.stream())
        if err == 1:  # hipErrorInvalidValue
            raise ValueError("SH encoder: input_dim must be 3 and degree in [1, 8]")
        L.check(err, "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        return outputs

    @staticmethod
    def backward(ctx, grad):
        # reference: sphere_harmonics.py:40-56
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        grad = grad.to(torch.float32).contiguous()
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        L.check(L.lib().tnl_sh_encode_backward(L.ptr(grad), L.ptr(inputs), L.u32(B), L.u32(input_dim), L.u32(degree),
                                               L.ptr(dy_dx), L.ptr(grad_inputs), L.stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
