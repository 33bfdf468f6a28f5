"""`_shencoder` -- the native module name the reference binds (aux_libs/shencoder/sphere_harmonics.py:9-12,
`import _shencoder as _backend`), served by libtrinerflet_hip.so.  The two functions of
aux_libs/shencoder/src/shencoder.h with their argument order; validation as shencoder.cu:401-411,420-433 does it
(device, contiguous, floating point -> RuntimeError), fp32 only."""
import torch

from trinerflet_amd import _lib as L

__all__ = ["sh_encode_forward", "sh_encode_backward"]


def _check(name, t):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")         # CHECK_CUDA, shencoder.cu:12
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")   # CHECK_CONTIGUOUS
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor (the reference's wrapper casts to float32)")


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx=None):
    """shencoder.h: inputs [B, D] -> outputs [B, C*C]; dy_dx [B, D*C*C] optional."""
    _check("inputs", inputs)
    _check("outputs", outputs)
    if dy_dx is not None:
        _check("dy_dx", dy_dx)
    err = L.lib().tnl_sh_encode_forward(L.ptr(inputs), L.ptr(outputs), L.u32(B), L.u32(D), L.u32(C), L.ptr(dy_dx),
                                        L.stream())
    if err:
        raise RuntimeError(f"sh_encode_forward: hipError {err} (D must be 3, degree 1..8)")


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    for n, t in (("grad", grad), ("inputs", inputs), ("dy_dx", dy_dx), ("grad_inputs", grad_inputs)):
        _check(n, t)
    err = L.lib().tnl_sh_encode_backward(L.ptr(grad), L.ptr(inputs), L.u32(B), L.u32(D), L.u32(C), L.ptr(dy_dx),
                                         L.ptr(grad_inputs), L.stream())
    if err:
        raise RuntimeError(f"sh_encode_backward: hipError {err}")
