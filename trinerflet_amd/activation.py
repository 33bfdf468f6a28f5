"""trunc_exp -- mirror of reconstruction/activation.py:5-17 (forward exp in fp32, backward clamps to +-15)."""
import torch
from torch.autograd import Function


class _trunc_exp(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.to(torch.float32)
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply
