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

fold_l1 (default on): the reference's regulariser is `sum_k w_k * coef_k.abs().mean()` over the tensors
TriPlaneVolume.get_wavelet_features() returns.  Through autograd its gradient costs a pass that writes sign(p) * s for
every coefficient and a second one that adds it to the data gradient (0.63 + 0.95 ms at base).  With fold_l1 the
backward of `.abs().mean()` (triplane_encoder._AbsMean) only adds its scalar s = grad_output / numel to this optimiser's
per-parameter "sink" and returns no gradient; step() hands the sink to the kernel, which adds s * sign(p) to the
gradient in registers (tnl_adam_l1_step_sink).  Same numbers up to one rounding (g * inv + (s * inv) * sign(p) instead of
(g + s * sign(p)) * inv).  Only parameters of a live FusedAdamL1 are folded; with any other optimiser the regulariser's
gradient is materialised as before.  A parameter whose .grad is None at step() is skipped as torch.optim.Adam skips it,
its folded term dropped (Trainer.clear_grad(), utils.py:1105-1114, freezes levels that way); l1_without_grad=True steps a
parameter that only the regulariser reached.  Not compatible with GradScaler.unscale_() before step() (the sink is in scaled
units): step() raises; construct with fold_l1=False for such loops.

GradScaler's inf check: for an optimiser with _step_supports_amp_scaling, GradScaler.step runs
_amp_foreach_non_finite_check_and_unscale_ over all gradients with a scale of 1 -- a pass that reads AND rewrites every
element (0.9 ms per step at base).  step() therefore takes the scaler itself (the `grad_scaler` keyword GradScaler passes
to optimisers that declare it; torch announces its removal with a FutureWarning, silenced here; without it the attributes
path above is used) and checks the gradients with a read-only pass (tnl_nonfinite_check, 0.3 ms), recording the result
where GradScaler.update() looks for it.

Live / deferred split (defer=True, the default; round 5).  With install_dropin()'s windowed rebuild a wavelet level's
gradient is zero outside a rectangle and nothing outside a slightly larger "live" rectangle is read by the renderer until
the occupancy window changes (TriPlaneVolume leaves both on the parameter after backward: `_tnl_live`).  A coefficient out
there still has to take Adam's step -- m and v decay, p coasts, the L1 term pulls -- but that update depends on nothing
except its own three numbers and the step's scalars, so it is not done every step: step() updates the live rectangle only
(tnl_adam_l1_step_live) and records the step's scalars (learning rate, bias corrections, skip flag, folded L1 coefficient)
in a 16-slot ring per parameter; the rest is replayed in registers, all pending steps in one pass
(tnl_adam_l1_catchup), when the ring is full, when the rectangle changes, before a whole-plane rebuild
(TriPlaneVolume.build_planes, e.g. the density-grid refresh every 16 steps), before state_dict() of the optimiser or of
the encoder, before load_state_dict() of either, before any method of a guarded EMA helper touches the parameters
(guard_ema_class: applied to torch_ema.ExponentialMovingAverage, the reference loop's EMA, automatically), and on
flush_deferred().  p, m, v after a flush are the bits the undeferred pass leaves
(tests/test_optim_gpu.py).  Until then the deferred coefficients' VALUES lag: code that reads the parameter tensors
directly (not through the encoder or a state_dict) calls trinerflet_amd.optim.flush_deferred() first; the regulariser's
reported VALUE (`.abs().mean()` over whole levels) lags by up to a period for them, its gradient does not.  1.95 -> ~1 ms of
the reference loop's 7.0 ms per step at the base configuration.

Arithmetic: the kernel's (m, v, p) update is torch's single-tensor Adam in fp32 with the bias corrections evaluated in
double on the device from the parameter's own `step` (tests/test_optim_gpu.py holds it against torch.optim.Adam).
Not supported (ValueError): amsgrad, maximize, differentiable, sparse gradients, non-fp32 or CPU parameters.
"""
import ctypes as C_
import warnings
import weakref

import torch

from . import _lib as L


def _i32x8(v):
    return (C_.c_int32 * 8)(*[int(x) for x in v[:8]])


class _L1Sink:
    """One float per parameter of a FusedAdamL1 on one device: the sum of d(scaled loss)/d(sum |p|) since the last step."""

    def __init__(self, owner, device, n):
        self.owner = weakref.ref(owner)
        self.vec = torch.zeros(n, dtype=torch.float32, device=device)
        self.touched = set()          # indices of the parameters a folded term was added for since the last step

    @property
    def used(self):
        return bool(self.touched)

    def clear(self):
        if self.touched:
            self.vec.zero_()
            self.touched.clear()

    def add(self, idx, grad_output, numel):
        """Called from the backward of coef.abs().mean(); False if the optimiser is gone (the caller then materialises)."""
        opt = self.owner()
        if opt is None or not opt.fold_l1:
            return False
        self.vec[idx:idx + 1].add_(grad_output.reshape(1).to(torch.float32), alpha=1.0 / numel)
        self.touched.add(idx)
        return True


_DEFERRING = weakref.WeakSet()       # live FusedAdamL1 instances that may hold deferred updates


def flush_deferred(params=None):
    """Replays the deferred part of every live FusedAdamL1's pending steps (for the given parameters, or all)."""
    for opt in list(_DEFERRING):
        opt.flush_deferred(params)


_EMA_METHODS = ("update", "store", "copy_to", "restore", "state_dict", "load_state_dict")


def guard_ema_class(cls):
    """Make an EMA helper safe beside the live / deferred split: every method of `cls` that reads or writes the parameter
    tensors directly (torch_ema.ExponentialMovingAverage's update / store / copy_to / restore / state_dict /
    load_state_dict) first replays the deferred steps of every live FusedAdamL1.  The reference's loop runs such an EMA by
    default (run_utils.py:93 `--ema_decay 0.95`; utils.py:1204-1207 update() after each epoch, :838-841 / :890 and
    :984-994 store() / copy_to() / restore() around every evaluation).  Without the guard update() would average
    coefficients that are up to 15 steps behind, and store() would save them -- the evaluation's whole-plane rebuild then
    replays the pending steps onto the EMA's shadow values and restore() writes the stale ones back: the steps are lost.
    Idempotent; returns cls.  FusedAdamL1(defer=True) applies it to `torch_ema` on its own when that module is loaded."""
    if getattr(cls, "_tnl_flush_guard", False):
        return cls
    import functools
    for name in _EMA_METHODS:
        fn = getattr(cls, name, None)          # (inherited methods too: the wrapper lands on cls itself)
        if fn is None or not callable(fn):
            continue

        def make(fn):
            @functools.wraps(fn)
            def guarded(self, *a, **kw):
                flush_deferred()
                return fn(self, *a, **kw)
            return guarded
        setattr(cls, name, make(fn))
    cls._tnl_flush_guard = True
    return cls


def _guard_loaded_emas():
    import sys
    mod = sys.modules.get("torch_ema")
    cls = getattr(mod, "ExponentialMovingAverage", None)
    if isinstance(cls, type):
        guard_ema_class(cls)
        return True
    return False


class FusedAdamL1(torch.optim.Optimizer):
    # torch.amp.GradScaler: do not unscale the gradients in a pass of their own -- step() receives the scale
    # (self.grad_scale) and the non-finite flag (self.found_inf) and folds both into the update
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *,
                 maximize=False, l1=0.0, fold_l1=True, l1_without_grad=False, defer=True):
        if amsgrad or maximize:
            raise ValueError("FusedAdamL1: amsgrad / maximize are not built")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("FusedAdamL1: invalid hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False, l1=l1)
        self.fold_l1 = bool(fold_l1)
        self.l1_without_grad = bool(l1_without_grad)
        self.defer = bool(defer)
        self._deferred = {}            # parameter -> {"ring", "pending", "live", "ctx"}: see the module docstring
        self._ema_guarded = bool(defer) and _guard_loaded_emas()      # (again at the first deferred step: import order)
        self.deferred_steps = self.deferred_flushes = 0
        _DEFERRING.add(self)
        self._sinks = {}
        warnings.filterwarnings("ignore", message="GradScaler is going to stop passing itself", category=FutureWarning)
        super().__init__(params, defaults)
        self._attach_sinks()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_sinks"):
            self._attach_sinks()

    def _attach_sinks(self):
        """(Re)builds the per-device sinks and tags every fp32 device parameter with (sink, index)."""
        by_dev = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.is_cuda and p.dtype == torch.float32:
                    by_dev.setdefault(p.device, []).append(p)
        self._sinks = {}
        for dev, ps in by_dev.items():
            sink = _L1Sink(self, dev, len(ps))
            self._sinks[dev] = sink
            for i, p in enumerate(ps):
                p._tnl_l1_sink = (sink, i)

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none)
        for group in self.param_groups:      # what a windowed backward left on the parameters belongs to that gradient
            for p in group["params"]:
                if getattr(p, "_tnl_live", None) is not None:
                    p._tnl_live = None
        for sink in self._sinks.values():
            sink.clear()

    # ---- live / deferred split ---------------------------------------------------------------------------------------
    def flush_deferred(self, params=None):
        """Replay the pending steps outside the live rectangles (all parameters, or the given ones)."""
        if not self._deferred:
            return
        want = None if params is None else {id(p) for p in params}
        lib = L.lib()
        for p, d in list(self._deferred.items()):
            if d["pending"] == 0 or (want is not None and id(p) not in want):
                continue
            st = self.state[p]
            b1, b2, eps, l1 = d["ctx"]
            n, C = p.shape[-1], p.shape[1]
            L.check(lib.tnl_adam_l1_catchup(L.ptr(p), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.u32(3 * C), L.u32(3),
                                            L.u32(n), L.u32(C), L.u32(0), _i32x8(d["live"]),
                                            L.ptr(d["ring"]), L.i32(d["pending"]), L.f32(b1), L.f32(b2), L.f32(eps), L.f32(l1),
                                            L.ptr(None), L.stream()), "adam_l1_catchup")
            d["pending"] = 0
            self.deferred_flushes += 1

    def state_dict(self):
        self.flush_deferred()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self.flush_deferred()
        return super().load_state_dict(state_dict)

    def _deferrable(self, p, g, group):
        """The live rectangle of this step if p's update can be split, else None."""
        info = getattr(p, "_tnl_live", None)
        if not self.defer or info is None or info[0] is None or p.dim() != 5 or p.shape[0] != 3 or p.shape[2] != 3:
            return None
        if p.grad is None or p.grad.data_ptr() != info[3] or g.data_ptr() != info[3] or p.grad._version != info[4]:
            return None        # not (only) the windowed chain's gradient (another tensor, or this one added into): dense
        n = p.shape[-1]
        if p.shape[-2] != n or n & (n - 1) or n % 4 or group["weight_decay"] != 0 or g.shape != p.shape or not g.is_contiguous():
            return None
        return info

    def _step_deferred(self, p, g, st, group, lr, found_inf, inv_scale, sink, sidx, info):
        live, rect = info[0], info[1]
        lib = L.lib()
        b1, b2 = group["betas"]
        ctx = (float(b1), float(b2), float(group["eps"]), float(group["l1"]))
        if not self._ema_guarded:
            self._ema_guarded = _guard_loaded_emas()
        d = self._deferred.get(p)
        if d is None:
            d = self._deferred[p] = {"ring": torch.zeros(16 * 4, dtype=torch.float32, device=p.device), "pending": 0,
                                     "live": None, "ctx": ctx}
        if d["pending"] and (d["live"] != list(live) or d["ctx"] != ctx or d["pending"] == 16):
            self.flush_deferred([p])
        d["live"], d["ctx"] = list(live), ctx
        slot = d["pending"]
        C = p.shape[1]
        n = p.shape[-1]
        L.check(lib.tnl_adam_record_step_l1(L.ptr(d["ring"]), L.i32(slot), L.f32(lr), L.ptr(st["step"].reshape(1)), L.f32(b1),
                                            L.f32(b2), L.ptr(found_inf.reshape(-1)),
                                            L.ptr(sink.vec[sidx:sidx + 1] if sink is not None else None), L.ptr(inv_scale),
                                            L.stream()), "adam_record_step_l1")
        L.check(lib.tnl_adam_l1_step_live(
            L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.u32(3 * C), L.u32(C), L.u32(0), L.u32(1),
            (C_.c_uint64 * 1)(0), (C_.c_uint32 * 1)(n), (C_.c_uint32 * 1)(3), _i32x8(live), _i32x8(rect),
            (C_.c_float * 1)(group["l1"]), L.f32(lr), L.ptr(st["step"].reshape(1)), L.ptr(d["ring"][4 * slot:]), L.f32(b1),
            L.f32(b2), L.f32(group["eps"]), L.f32(1.0), L.ptr(inv_scale), L.ptr(found_inf.reshape(-1)), L.ptr(None),
            L.stream()), "adam_l1_step_live")
        d["pending"] += 1
        self.deferred_steps += 1

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

    def _check_grads(self, device):
        """GradScaler's found_inf over this optimiser's gradients: [1] float on `device`, 1 if any is inf / nan."""
        lib = L.lib()
        found = torch.zeros(1, dtype=torch.float32, device=device)
        rest = []
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if (g.device == device and g.dtype == torch.float32 and not g.is_sparse and g.is_contiguous()
                        and g.numel() >= 65536 and g.data_ptr() % 16 == 0):
                    L.check(lib.tnl_nonfinite_check(L.ptr(g), L.u64(g.numel()), L.ptr(found), L.stream()), "nonfinite_check")
                else:
                    rest.append(g)
        if rest:
            if any(g.device != device for g in rest):
                raise ValueError("FusedAdamL1: parameters on more than one device")
            torch._amp_foreach_non_finite_check_and_unscale_(rest, found, torch.ones((), dtype=torch.float32, device=device))
        return found

    def _amp_state(self, scaler):
        """(found_inf, grad_scale) when GradScaler.step hands itself over (see the module docstring)."""
        from torch.amp.grad_scaler import OptState
        state = scaler._per_optimizer_states[id(self)]
        scale = scaler._get_scale_async()
        if state["stage"] is OptState.READY:
            found = self._check_grads(scale.device)
            state["found_inf_per_device"] = {found.device: found}
            return found, scale
        # scaler.unscale_(self) ran: its verdict stands and the gradients are in true units
        return sum(t.to(scale.device, non_blocking=True) for t in state["found_inf_per_device"].values()), None

    @torch.no_grad()
    def step(self, closure=None, grad_scaler=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        # set by GradScaler.step around this call (and deleted after it); absent when the optimiser is stepped directly
        found_inf, grad_scale, inv_scale = getattr(self, "found_inf", None), getattr(self, "grad_scale", None), None
        if grad_scaler is not None:
            found_inf, grad_scale = self._amp_state(grad_scaler)
        if found_inf is not None and grad_scale is None and any(s_.used for s_ in self._sinks.values()):
            raise RuntimeError("FusedAdamL1(fold_l1=True): the folded L1 term is in loss-scaled units; GradScaler.unscale_() "
                               "before step() is not supported -- construct with fold_l1=False")
        steps = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr) if not torch.is_tensor(lr) else float(lr.item())
            for p in group["params"]:
                sink, sidx = getattr(p, "_tnl_l1_sink", (None, 0))
                if sink is not None and not (sidx in sink.touched and sink.owner() is self):
                    sink = None
                if p.grad is None:
                    # torch.optim.Adam skips a parameter without a gradient, and so does this pass -- its folded L1 term
                    # included: the reference's Trainer.clear_grad() (nerf/utils.py:1105-1114) freezes coarse levels by
                    # setting .grad = None AFTER backward, and the term sitting in the sink must not outlive that.  A
                    # parameter that only the regulariser reaches is stepped when asked for (l1_without_grad=True).
                    p._tnl_live = None
                    if sink is None or not self.l1_without_grad:
                        continue
                    p.grad = torch.zeros_like(p)
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
                if sink is not None and use_inv is None and inv_scale is not None:
                    raise ValueError("FusedAdamL1: weight_decay together with a folded L1 term under GradScaler is not built")
                info = self._deferrable(p, g, group)
                if info is not None:
                    self._step_deferred(p, g, st, group, lr, found_inf, inv_scale, sink, sidx, info)
                    p._tnl_live = None
                    steps.append(st["step"])
                    continue
                if p in self._deferred and self._deferred[p]["pending"]:
                    self.flush_deferred([p])          # this step takes the whole-array pass: the pending ones first
                L.check(lib.tnl_adam_l1_step_sink(
                    L.ptr(p), L.ptr(use_g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.u64(p.numel()), L.f32(lr),
                    L.ptr(st["step"].reshape(1)), L.f32(b1), L.f32(b2), L.f32(group["eps"]), L.ptr(use_inv),
                    L.f32(group["l1"]), L.ptr(sink.vec[sidx:sidx + 1] if sink is not None else None),
                    L.ptr(found_inf.reshape(-1)), L.stream()), "adam_l1_step_sink")
                # torch: `step` advances only when the update is applied (GradScaler skips the whole step() otherwise)
                steps.append(st["step"])
        if steps:
            torch._foreach_add_(steps, 1.0 - found_inf.reshape(()).to(torch.float32))
        for sink in self._sinks.values():
            sink.clear()
        return loss


_ORIG_ADAM = None


def patch_torch_adam():
    """install_dropin(fused_adam=True): `torch.optim.Adam(...)` returns a FusedAdamL1 when every parameter is a dense fp32
    device tensor and no option outside FusedAdamL1's set is asked for -- reconstruction/main_nerf.py:119 then needs no edit
    at all -- and the real torch.optim.Adam in every other case.  `torch.optim.Adam` stays a class; isinstance(opt,
    torch.optim.Optimizer) holds for both results, isinstance(opt, torch.optim.Adam) only for the real one.
    unpatch_torch_adam() restores it."""
    global _ORIG_ADAM
    if _ORIG_ADAM is not None:
        return
    orig = torch.optim.Adam
    _ORIG_ADAM = orig

    class Adam(orig):
        def __new__(cls, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
            params = list(params)
            groups = params if (params and isinstance(params[0], dict)) else [{"params": params}]
            flat = [p for g in groups for p in (g["params"] if isinstance(g["params"], (list, tuple)) else [g["params"]])]
            plain = (not amsgrad and not any(kw.get(k) for k in ("maximize", "foreach", "capturable", "differentiable", "fused"))
                     and not any(g.get("amsgrad") or g.get("maximize") for g in groups)
                     and not torch.is_tensor(lr) and len(flat) > 0
                     and all(torch.is_tensor(p) and p.is_cuda and p.dtype == torch.float32 and not p.is_sparse for p in flat))
            if plain:
                return FusedAdamL1(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
            return orig(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)

    Adam.__name__ = Adam.__qualname__ = "Adam"
    torch.optim.Adam = Adam


def unpatch_torch_adam():
    global _ORIG_ADAM
    if _ORIG_ADAM is not None:
        torch.optim.Adam = _ORIG_ADAM
        _ORIG_ADAM = None
