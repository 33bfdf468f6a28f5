"""FusedAdamL1 -- torch.optim.Adam(betas, eps, weight_decay) as ONE HIP pass per parameter (csrc/adam.hip,
tnl_adam_l1_step_dev: read p, g, m, v, write p, m, v = 28 B per element at the HBM rate), with GradScaler's unscale
folded in and, optionally, the wavelet L1 regulariser's gradient (reconstruction/nerf/utils.py:639-655) added inside
the pass instead of through autograd.

The one-line change for the reference's own training loop (reconstruction/main_nerf.py:119):

    optimizer = lambda model: trinerflet_amd.optim.FusedAdamL1(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)

The loop of utils.py:1134-1175 (scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()) and its
checkpoints stay as they are: state_dict() has torch.optim.Adam's layout (per parameter 'step', 'exp_avg',
'exp_avg_sq'; the same param_groups keys), so a checkpoint written with either optimiser loads into the other.
torch.optim.Adam runs ~13 multi-tensor kernels over the parameters per step (6.9 ms at the base configuration's 403 M
coefficients) plus GradScaler's unscale pass (0.9 ms); this pass takes ~2 ms.

Arithmetic: the kernel's (m, v, p) update is torch's single-tensor Adam in fp32 with the bias corrections evaluated in
double on the device from the parameter's own `step` (tests/test_optim_gpu.py holds it against torch.optim.Adam).
Not supported (ValueError): amsgrad, maximize, differentiable, sparse gradients, non-fp32 or CPU parameters.
"""
import torch

from . import _lib as L


class FusedAdamL1(torch.optim.Optimizer):
    # torch.amp.GradScaler: do not unscale the gradients in a pass of their own -- step() receives the scale
    # (self.grad_scale) and the non-finite flag (self.found_inf) and folds both into the update
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *,
                 maximize=False, l1=0.0):
        if amsgrad or maximize:
            raise ValueError("FusedAdamL1: amsgrad / maximize are not built")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("FusedAdamL1: invalid hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False, l1=l1)
        super().__init__(params, defaults)

    def __setstate__(self, state):
        super().__setstate__(state)
        for g in self.param_groups:
            g.setdefault("l1", 0.0)
            g.setdefault("weight_decay", 0.0)

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not torch.is_tensor(st["step"]) or st["step"].device != p.device or st["step"].dtype != torch.float32:
            # a state loaded from torch.optim.Adam keeps `step` as a CPU tensor (or a number in old checkpoints)
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device).reshape(())
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        # set by GradScaler.step around this call (and deleted after it); absent when the optimiser is stepped directly
        found_inf, grad_scale, inv_scale = getattr(self, "found_inf", None), getattr(self, "grad_scale", None), None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr) if not torch.is_tensor(lr) else float(lr.item())
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("FusedAdamL1: dense contiguous fp32 device parameters only")
                g = g if (g.is_contiguous() and g.dtype == torch.float32) else g.to(torch.float32).contiguous()
                if found_inf is None:
                    found_inf = torch.zeros(1, dtype=torch.float32, device=p.device)
                elif found_inf.device != p.device:
                    found_inf = found_inf.to(p.device)
                if grad_scale is not None and inv_scale is None:
                    inv_scale = grad_scale.to(device=p.device, dtype=torch.float32).reciprocal().reshape(1)
                st = self._state_of(p)
                if group["weight_decay"] != 0:        # Adam's L2 term enters the gradient (torch: grad.add(param, alpha=wd))
                    wd_g = g * (inv_scale if inv_scale is not None else 1.0) + group["weight_decay"] * p
                    use_g, use_inv = wd_g, None
                else:
                    use_g, use_inv = g, inv_scale
                L.check(lib.tnl_adam_l1_step_dev(
                    L.ptr(p), L.ptr(use_g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.u64(p.numel()), L.f32(lr),
                    L.ptr(st["step"].reshape(1)), L.f32(b1), L.f32(b2), L.f32(group["eps"]), L.f32(1.0), L.ptr(use_inv),
                    L.f32(group["l1"]), L.ptr(found_inf.reshape(-1)), L.ptr(None), L.i32(0), L.stream()), "adam_l1_step_dev")
                # torch: `step` advances only when the update is applied (GradScaler skips the whole step() otherwise)
                st["step"].add_(1.0 - found_inf.reshape(()).to(torch.float32))
        return loss
