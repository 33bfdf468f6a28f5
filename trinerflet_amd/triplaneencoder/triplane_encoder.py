"""TriPlaneVolume -- mirror of reconstruction/triplaneencoder/triplane_encoder.py (reference) whose
arithmetic runs in libtrinerflet_hip.so.

Same constructor signature, parameter names/shapes (state-dict compatible: `planes_features`,
`planes_features_wavelet_coefs.{i}`, `plane_axes`, `plane_normals`, `idwt.*`) and public methods:
forward(coordinates, bound), get_planes(max_res, max_scale, get_all_resolutions), reset_cahce() [sic],
get_wavelet_features(), get_wavelet_features_upscaled(), get_lbound_scale(), sample_from_planes(),
get_params().

What differs underneath (MI355X-first):
  * build_planes (:364-405) = one LDS-tiled HIP kernel per wavelet level (csrc/wavelet.hip) instead of
    3 conv_transpose2d pairs + 2 F.pad copies per level; its autograd is the adjoint kernel.
  * sampling reads a texel-major [3,R,R,C] copy of the planes (fp16 by default = the "e=2" fast mode of
    SURVEY.md 8(d); `plane_dtype=torch.float32` is the train-parity mode) so a texel's channels are one
    contiguous segment; get_planes() still returns the reference-shaped (3,C,R,R) fp32 tensor.
Options no README configuration uses -- learn_rotation_axis, lbound_auto_scale,
upscale_ratio_bound / upscale_levels (nested zoom planes), apply_activation_on_features,
inner_multi_res_scale_current != 1, get_grid_features, get_params2 -- follow the reference line by line on top of
the same kernels (the IDWT levels and, where the coordinates are not re-mapped per plane, the lookup); the two
re-mapped lookups (per-plane coordinate scale, per-channel rotated axes) use torch's grid_sample on the GPU as the
reference does.  They switch the fused field / TrainStep fast path off (`is_plain()` is False).  Pinned by
tests/golden/triplane_options_reference.npz (the reference class run here).  Only wavelet_base_resolution > 0 raises
NotImplementedError.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib as L
from . import utils

WAVELET_IDS = {'haar': 0, 'bior2.2': 1, 'bior4.4': 2, 'bior2.6': 3, 'bior6.8': 4}
# triplane_encoder.py:174-180
PAD_DICT = {'bior6.8': 4, 'bior2.6': 3, 'bior4.4': 2, 'bior2.2': 1, 'haar': 0}
# pywt.Wavelet(w).rec_lo / rec_hi, only used to fill the `idwt.*` buffers for state-dict compatibility
_REC = {
    'haar': ([0.7071067811865476, 0.7071067811865476], [0.7071067811865476, -0.7071067811865476]),
    'bior2.2': ([0, 0.3535533905932738, 0.7071067811865476, 0.3535533905932738, 0, 0],
                [0, 0.1767766952966369, 0.3535533905932738, -1.0606601717798212, 0.3535533905932738,
                 0.1767766952966369]),
    'bior4.4': ([0, -0.06453888262869706, -0.04068941760916406, 0.41809227322161724, 0.7884856164055829,
                 0.41809227322161724, -0.04068941760916406, -0.06453888262869706, 0, 0],
                [0, -0.03782845550726404, -0.023849465019556843, 0.11062440441843718, 0.37740285561283066,
                 -0.8526986790088938, 0.37740285561283066, 0.11062440441843718, -0.023849465019556843,
                 -0.03782845550726404]),
    'bior2.6': ([0, 0, 0, 0, 0, 0.3535533905932738, 0.7071067811865476, 0.3535533905932738, 0, 0, 0, 0, 0, 0],
                [0, 0.006905339660024878, 0.013810679320049757, -0.04695630968816917, -0.1077232986963881,
                 0.16987135563661201, 0.4474660099696121, -0.966747552403483, 0.4474660099696121,
                 0.16987135563661201, -0.1077232986963881, -0.04695630968816917, 0.013810679320049757,
                 0.006905339660024878]),
    'bior6.8': ([0, 0, 0, 0.014426282505624435, 0.014467504896790148, -0.07872200106262882,
                 -0.04036797903033992, 0.41784910915027457, 0.7589077294536541, 0.41784910915027457,
                 -0.04036797903033992, -0.07872200106262882, 0.014467504896790148, 0.014426282505624435,
                 0, 0, 0, 0],
                [0, -0.0019088317364812906, -0.0019142861290887667, 0.016990639867602342, 0.01193456527972926,
                 -0.04973290349094079, -0.07726317316720414, 0.09405920349573646, 0.4207962846098268,
                 -0.8259229974584023, 0.4207962846098268, 0.09405920349573646, -0.07726317316720414,
                 -0.04973290349094079, 0.01193456527972926, 0.016990639867602342, -0.0019142861290887667,
                 -0.0019088317364812906]),
}


# ------------------------------------------------------------------------------------------------
# autograd wrappers of the C ABI
# ------------------------------------------------------------------------------------------------
class _IDWTLevel(Function):
    """x:(3,C,n,n), yh:(3,C,3,n,n) -> (3,C,2n,2n) = idwt((pad(2x),[pad(yh)])) (triplane_encoder.py:379-394)."""

    @staticmethod
    def forward(ctx, x, yh, wave_id):
        L.require_cuda(x, yh)
        x = x.to(torch.float32).contiguous()
        yh = yh.to(torch.float32).contiguous()
        P, C, n = x.shape[0], x.shape[1], x.shape[-1]
        out = torch.empty(P, C, 2 * n, 2 * n, dtype=torch.float32, device=x.device)
        L.check(L.lib().tnl_idwt_level_forward(L.ptr(x), L.ptr(yh), L.u32(P * C), L.u32(n), L.i32(wave_id),
                                               L.ptr(out), L.stream()), "idwt_level_forward")
        ctx.wave_id = wave_id
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.to(torch.float32).contiguous()
        P, C, n = g.shape[0], g.shape[1], g.shape[-1] // 2
        dx = torch.empty(P, C, n, n, dtype=torch.float32, device=g.device)
        dyh = torch.empty(P, C, 3, n, n, dtype=torch.float32, device=g.device)
        L.check(L.lib().tnl_idwt_level_backward(L.ptr(g), L.u32(P * C), L.u32(n), L.i32(ctx.wave_id), L.ptr(dx),
                                                L.ptr(dyh), L.stream()), "idwt_level_backward")
        return dx, dyh, None


class _IDWTChainWin(Function):
    """All J levels of build_planes() (triplane_encoder.py:364-405, plain geometry) with the result restricted to the
    occupancy window `roi` (8 ints, occupancy.window_from_bounds) of the finest grid: every level computes only the window
    of its output the next level needs (occupancy.level_windows); the rest of the returned (3,C,R,R) array is
    UNINITIALISED.  backward: the incoming gradient is read inside `roi` only (what the fused field's backward writes
    when it is given the same window); each level's coefficient gradient is written inside the rectangle of coarse
    tiles the window reaches and is zero elsewhere (tnl_idwt_level_backward_win's support chain).  Inside the window /
    rectangles both directions equal the whole-plane levels bit for bit (the same kernels, fewer workgroups)."""

    @staticmethod
    def forward(ctx, wave_id, roi, ll, *coefs):
        from .. import occupancy
        L.require_cuda(ll)
        lib = L.lib()
        J, C = len(coefs), ll.shape[1]
        R = ll.shape[-1] << J
        wins = occupancy.level_windows(list(roi), J, R)
        x = ll.detach().to(torch.float32).contiguous()
        for lvl in range(J):
            yh = coefs[lvl].detach().to(torch.float32).contiguous()
            n = x.shape[-1]
            out = torch.empty(3, C, 2 * n, 2 * n, dtype=torch.float32, device=x.device)
            if wins[lvl] is None:
                L.check(lib.tnl_idwt_level_forward(L.ptr(x), L.ptr(yh), L.u32(3 * C), L.u32(n), L.i32(wave_id), L.ptr(out),
                                                   L.stream()), "idwt_level_forward")
            else:
                L.check(lib.tnl_idwt_level_forward_win(L.ptr(x), L.ptr(yh), L.u32(3 * C), L.u32(n), L.i32(wave_id),
                                                       L.ptr(out), L.roi_array(list(wins[lvl]) + [C, 0]), L.stream()),
                        "idwt_level_forward_win")
            x = out
        ctx.meta = (wave_id, tuple(int(v) for v in roi), J, C, R)
        # for optim.FusedAdamL1's live / deferred split: the coefficient parameters this chain was built from (weak) and the
        # forward windows -- backward leaves each level's live rectangle on its parameter
        import weakref
        ctx.coef_refs = [weakref.ref(c) for c in coefs]
        ctx.fwd_wins = wins
        return x

    @staticmethod
    def backward(ctx, g):
        import ctypes
        wave_id, roi, J, C, R = ctx.meta
        lib = L.lib()
        g = g.to(torch.float32).contiguous()
        dev = g.device
        win = list(roi)
        grads = [None] * J
        rects = [None] * J
        for lvl in reversed(range(J)):
            n = R >> (J - lvl)
            # the coarsest dx is the LL parameter's gradient: like the band gradients it must be zero outside the rectangle
            dx = (torch.zeros if lvl == 0 else torch.empty)(3, C, n, n, dtype=torch.float32, device=dev)
            dyh = torch.zeros(3, C, 3, n, n, dtype=torch.float32, device=dev)
            rect = (ctypes.c_int32 * 8)()
            L.check(lib.tnl_idwt_level_backward_win(L.ptr(g), L.u32(3 * C), L.u32(n), L.i32(wave_id), L.ptr(dx), L.ptr(dyh),
                                                    L.roi_array(win + [C, 0]), L.i32(1), rect, L.stream()),
                    "idwt_level_backward_win")
            win = list(rect)
            grads[lvl] = dyh
            rects[lvl] = list(rect)
            g = dx
        # what a step's optimiser pass has to touch per level: everything the windowed rebuild reads united with the
        # gradient's rectangle (occupancy.live_rects) -- left on the parameter together with the rectangle and the window
        from .. import occupancy
        live = occupancy.live_rects(ctx.fwd_wins, rects, [R >> (J - lvl) for lvl in range(J)])
        for lvl, ref in enumerate(ctx.coef_refs):
            prm = ref()
            if prm is not None:
                # (+ the address AND the version counter of the gradient tensor this backward returns: the optimiser takes
                #  the split only while .grad is exactly that tensor, untouched.  The address alone does not tell: when
                #  another recorded path reaches the coefficient -- a materialised regulariser, an extra loss term -- the
                #  engine sums the two gradients IN PLACE in the leaf's input buffer, which may be this very tensor; an
                #  in-place add bumps the version, a steal by AccumulateGrad (a detach) shares it)
                prm._tnl_live = (live[lvl], rects[lvl], roi, grads[lvl].data_ptr(), grads[lvl]._version)
        return (None, None, g, *grads)


def idwt_level_half(x, yh, wave_id):
    """Finest level straight to fp16 (no autograd): x (3,C,n,n), yh (3,C,3,n,n) fp32 -> (3,C,2n,2n) fp16."""
    x = x.detach().to(torch.float32).contiguous()
    yh = yh.detach().to(torch.float32).contiguous()
    P, C, n = x.shape[0], x.shape[1], x.shape[-1]
    out = torch.empty(P, C, 2 * n, 2 * n, dtype=torch.float16, device=x.device)
    L.check(L.lib().tnl_idwt_level_forward_half(L.ptr(x), L.ptr(yh), L.u32(P * C), L.u32(n), L.i32(wave_id),
                                                L.ptr(out), L.stream()), "idwt_level_forward_half")
    return out


def idwt_level_half_roi(x, yh, wave_id, roi, spans=None):
    """Finest level over the ROI window only: -> compact fp16 (P*C, rh, rw).  roi = 8 ints, see
    include/trinerflet_hip.h (window of the 2n x 2n grid, multiples of 64).  spans: device table of the pieces of the
    window anything reads (tnl_idwt_level_forward_spans); the rest of the result is then undefined."""
    x = x.detach().to(torch.float32).contiguous()
    yh = yh.detach().to(torch.float32).contiguous()
    P, C, n = x.shape[0], x.shape[1], x.shape[-1]
    out = torch.empty(P * C, roi[7], roi[6], dtype=torch.float16, device=x.device)
    L.check(L.lib().tnl_idwt_level_forward_spans(L.ptr(x), L.ptr(yh), L.u32(P * C), L.u32(n), L.i32(wave_id),
                                                 L.ptr(out), L.i32(1), L.roi_array(roi), L.i32(0), L.ptr(spans),
                                                 L.stream()),
            "idwt_level_forward_spans")
    return out


def half_roi_into_texel_major(planes_roi_half, tm, roi, spans=None):
    """Compact fp16 window (3C, rh, rw) -> the same window of the full fp16 [3,R,R,C] array `tm`, in place (spans: only
    the 64-texel row pieces that meet the table's pieces of the plane grid)."""
    _, R, _, C = tm.shape
    L.check(L.lib().tnl_planes_half_to_texel_major_spans(L.ptr(planes_roi_half), L.u32(C), L.u32(R), L.ptr(tm),
                                                         L.roi_array(roi), L.ptr(spans), L.stream()),
            "planes_half_to_texel_major_spans")
    return tm


def half_to_texel_major(planes_cm_half):
    """fp16 (3,C,R,R) -> fp16 [3,R,R,C] (no autograd)."""
    _, C, R, _ = planes_cm_half.shape
    tm = torch.empty(3, R, R, C, dtype=torch.float16, device=planes_cm_half.device)
    L.check(L.lib().tnl_planes_half_to_texel_major(L.ptr(planes_cm_half), L.u32(C), L.u32(R), L.ptr(tm), L.stream()),
            "planes_half_to_texel_major")
    return tm


class _ToTexelMajor(Function):
    """(3,C,R,R) fp32 -> [3,R,R,C] fp16|fp32 ; backward: fp32 [3,R,R,C] -> (3,C,R,R)."""

    @staticmethod
    def forward(ctx, planes_cm, half, window=None):
        """window (8 ints, occupancy.window_from_bounds): only that part of each plane is converted; the rest of the
        returned array is uninitialised (csrc: tnl_planes_to_texel_major_win)."""
        L.require_cuda(planes_cm)
        planes_cm = planes_cm.to(torch.float32).contiguous()
        _, C, R, _ = planes_cm.shape
        tm = torch.empty(3, R, R, C, dtype=torch.float16 if half else torch.float32, device=planes_cm.device)
        if window is not None:
            L.check(L.lib().tnl_planes_to_texel_major_win(L.ptr(planes_cm), L.u32(C), L.u32(R), L.i32(1 if half else 0),
                                                          L.ptr(tm), L.roi_array(list(window) + [C, 0]), L.stream()),
                    "planes_to_texel_major_win")
            return tm
        L.check(L.lib().tnl_planes_to_texel_major(L.ptr(planes_cm), L.u32(C), L.u32(R), L.i32(1 if half else 0),
                                                  L.ptr(tm), L.stream()), "planes_to_texel_major")
        return tm

    @staticmethod
    def backward(ctx, g_tm):
        g_tm = g_tm.to(torch.float32).contiguous()
        _, R, _, C = g_tm.shape
        g_cm = torch.empty(3, C, R, R, dtype=torch.float32, device=g_tm.device)
        L.check(L.lib().tnl_planes_to_channel_major(L.ptr(g_tm), L.u32(C), L.u32(R), L.ptr(g_cm), L.stream()),
                "planes_to_channel_major")
        return g_cm, None, None


class _Sample(Function):
    """planes_tm [3,R,R,C], xyz [N,3] -> feats [N,3C] (triplane_encoder.py:314-332)."""

    @staticmethod
    def forward(ctx, planes_tm, xyz, bound):
        L.require_cuda(planes_tm, xyz)
        xyz = xyz.to(torch.float32).contiguous()
        _, R, _, C = planes_tm.shape
        N = xyz.shape[0]
        feats = torch.empty(N, 3 * C, dtype=torch.float32, device=xyz.device)
        L.check(L.lib().tnl_triplane_sample_forward(L.ptr(planes_tm), L.i32(int(planes_tm.dtype == torch.float16)),
                                                    L.ptr(xyz), L.f32(bound), L.u32(N), L.u32(C), L.u32(R),
                                                    L.ptr(feats), L.stream()), "triplane_sample_forward")
        ctx.save_for_backward(xyz)
        ctx.dims = (C, R, float(bound))
        return feats

    @staticmethod
    def backward(ctx, g):
        (xyz,) = ctx.saved_tensors
        C, R, bound = ctx.dims
        g = g.to(torch.float32).contiguous()
        g_tm = torch.zeros(3, R, R, C, dtype=torch.float32, device=g.device)
        L.check(L.lib().tnl_triplane_sample_backward(L.ptr(g), L.ptr(xyz), L.f32(bound), L.u32(xyz.shape[0]),
                                                     L.u32(C), L.u32(R), L.ptr(g_tm), L.stream()),
                "triplane_sample_backward")
        return g_tm, None, None


class _GridSample(Function):
    """planes_tm [3,R,R,C], grid [N,3,CG,2] (normalised, gx -> W) -> feats [N,3C]: F.grid_sample(bilinear, border,
    align_corners=True) of the reference's optional lookups (triplane_encoder.py:323-329,335-362) on the texel-major
    planes, differentiable w.r.t. the planes and the grid (csrc/triplane.hip)."""

    @staticmethod
    def forward(ctx, planes_tm, grid):
        L.require_cuda(planes_tm, grid)
        grid = grid.to(torch.float32).contiguous()
        _, R, _, C = planes_tm.shape
        N, _, CG, _ = grid.shape
        feats = torch.empty(N, 3 * C, dtype=torch.float32, device=grid.device)
        L.check(L.lib().tnl_grid_sample_tm_forward(L.ptr(planes_tm), L.i32(int(planes_tm.dtype == torch.float16)),
                                                   L.ptr(grid), L.u32(N), L.u32(C), L.u32(CG), L.u32(R), L.ptr(feats),
                                                   L.stream()), "grid_sample_tm_forward")
        ctx.save_for_backward(planes_tm, grid)
        return feats

    @staticmethod
    def backward(ctx, g):
        planes_tm, grid = ctx.saved_tensors
        _, R, _, C = planes_tm.shape
        N, _, CG, _ = grid.shape
        g = g.to(torch.float32).contiguous()
        g_tm = torch.zeros(3, R, R, C, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        g_grid = (torch.zeros_like(grid) if CG == 1 else torch.empty_like(grid)) if ctx.needs_input_grad[1] else None
        L.check(L.lib().tnl_grid_sample_tm_backward(L.ptr(planes_tm), L.i32(int(planes_tm.dtype == torch.float16)),
                                                    L.ptr(grid), L.ptr(g), L.u32(N), L.u32(C), L.u32(CG), L.u32(R),
                                                    L.ptr(g_tm), L.ptr(g_grid), L.stream()), "grid_sample_tm_backward")
        return g_tm, g_grid


class _AbsMean(Function):
    """mean |x| as one reduction pass, its gradient sign(x) * g / n as one pass (csrc/loss.hip)."""

    @staticmethod
    def usable(x):
        return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() > 0
                and x.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, sink=None):
        lib = L.lib()
        ws = torch.empty(int(lib.tnl_abs_mean_workspace()), dtype=torch.uint8, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(lib.tnl_abs_mean_forward(L.ptr(x), L.u64(x.numel()), L.ptr(ws), L.ptr(out), L.stream()), "abs_mean_forward")
        ctx.save_for_backward(x)
        # sink: (optim._L1Sink, index) of the PARAMETER x aliases, when it belongs to a live optim.FusedAdamL1(fold_l1=True):
        # the gradient sign(x) * g / n is then applied inside the optimiser's pass from the scalar alone, not materialised
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if ctx.sink is not None and ctx.sink[0].add(ctx.sink[1], g, x.numel()):
            return None, None
        gx = torch.empty_like(x)
        g = g.to(torch.float32).contiguous()
        L.check(L.lib().tnl_abs_mean_backward(L.ptr(x), L.u64(x.numel()), L.ptr(g), L.ptr(gx), L.stream()),
                "abs_mean_backward")
        return gx, None


class _LazyAbs:
    """What `coef.abs()` returns for the tensors get_wavelet_features() hands out: |coef| not yet evaluated.  `.mean()`
    with no arguments -- the reference's regulariser, utils.py:639-655 -- runs the fused reduction; anything else
    (another method, an operator, a torch function) evaluates torch.abs first and carries on with an ordinary tensor."""
    __slots__ = ("_src", "_val", "_sink")

    def __init__(self, src, sink=None):
        self._src, self._val, self._sink = src, None, sink

    def _tensor(self):
        if self._val is None:
            self._val = torch.abs(self._src)
        return self._val

    def mean(self, *args, **kwargs):
        if not args and not kwargs and self._val is None and _AbsMean.usable(self._src):
            return _AbsMean.apply(self._src, self._sink)
        return self._tensor().mean(*args, **kwargs)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.mean, torch.Tensor.mean) and len(args) == 1 and not kwargs and isinstance(args[0], _LazyAbs):
            return args[0].mean()
        conv = lambda a: a._tensor() if isinstance(a, _LazyAbs) else a
        return func(*torch.utils._pytree.tree_map(conv, args), **torch.utils._pytree.tree_map(conv, kwargs))

    def __getattr__(self, name):
        return getattr(self._tensor(), name)

    def __repr__(self):
        return repr(self._tensor())

    def __len__(self):
        return len(self._tensor())

    def __getitem__(self, k):
        return self._tensor()[k]

    def __neg__(self):
        return -self._tensor()


for _op in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "rpow", "matmul", "lt", "le", "gt",
            "ge", "eq", "ne"):
    setattr(_LazyAbs, f"__{_op}__", (lambda name: lambda self, o: getattr(self._tensor(), name)(o))(f"__{_op}__"))


class _CoefView(torch.Tensor):
    """A coefficient Parameter as get_wavelet_features() hands it out (Tensor.as_subclass: same storage, same place in
    the autograd graph).  Behaves like the parameter, except that `.abs()` is deferred (see _LazyAbs) so that the
    regulariser's `val.abs().mean()` is one fused pass forward and one backward."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.abs, torch.Tensor.abs) and len(args) == 1 and not kwargs and type(args[0]) is _CoefView:
            return _LazyAbs(args[0].as_subclass(torch.Tensor), getattr(args[0], "_tnl_l1_sink", None))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return torch.utils._pytree.tree_map(lambda t: t.as_subclass(torch.Tensor) if type(t) is _CoefView else t, out)


def _flush_deferred_optimisers(params):
    """A reader of whole coefficient arrays is about to run (whole-plane rebuild, state_dict): optimisers that keep part of
    these parameters' updates deferred (optim.FusedAdamL1(defer=True)) replay them now."""
    from .. import optim
    optim.flush_deferred(params)


class _IDWTBuffers(nn.Module):
    """Holds pytorch_wavelets.DWTInverse's filter buffers (g0_col, g1_col, g0_row, g1_row) so that
    reference checkpoints (`encoder.idwt.*` keys) load and save unchanged.  Not used for compute."""

    def __init__(self, wave):
        super().__init__()
        lo, hi = _REC[wave]
        g0 = torch.tensor(lo, dtype=torch.float32)
        g1 = torch.tensor(hi, dtype=torch.float32)
        self.register_buffer('g0_col', g0.reshape(1, 1, -1, 1))
        self.register_buffer('g1_col', g1.reshape(1, 1, -1, 1))
        self.register_buffer('g0_row', g0.reshape(1, 1, 1, -1))
        self.register_buffer('g1_row', g1.reshape(1, 1, 1, -1))


# default of TriPlaneVolume.windowed_autograd for encoders constructed from now on (install_dropin() sets it)
WINDOWED_AUTOGRAD = False


class TriPlaneVolume(torch.nn.Module):
    _upgrading = False

    def __init__(self, number_of_features=3, plane_resolution=224, init_sigma=0.1, lbound=1,
                 viewdir_plane_resolution=32,
                 two_planes_per_axis=False,
                 planes_features=None, viewdir_plane=None, apply_activation_on_features=False,
                 inner_multi_res_scale=1, inner_multi_res_viewdir_scale=1, viewdir_mode='plane',
                 inner_multi_res_scale_current=1,
                 learn_rotation_axis=False,
                 dropout=0,
                 wavelet_type='bior6.8',
                 lbound_auto_scale=False,
                 upscale_ratio_bound=-1,
                 upscale_levels=2,
                 wavelet_base_resolution=0,
                 plane_dtype=torch.float16,
                 ):
        super().__init__()
        # reference: triplane_encoder.py:27-94
        self.number_of_features = number_of_features
        self.plane_resolution = plane_resolution
        self.init_sigma = init_sigma
        self.lbound = lbound
        self.lbound_viewdir = 1
        self.output_dim = 3 * self.number_of_features
        self.viewdir_plane_resolution = viewdir_plane_resolution
        self.two_planes_per_axis = two_planes_per_axis
        self.apply_activation_on_features = apply_activation_on_features
        self.plane_dtype = plane_dtype

        # create_subplanes_trivial_base (:250-289): up (x,z | y), front (x,y | z), right (y,z | x)
        eye = torch.eye(3)
        plane_axes = torch.stack([torch.cat([eye[:, 0:1], eye[:, 2:]], 1), torch.cat([eye[:, 0:1], eye[:, 1:2]], 1),
                                  torch.cat([eye[:, 1:2], eye[:, 2:]], 1)], 0)
        plane_normals = torch.stack([eye[:, 1:2], eye[:, 2:], eye[:, 0:1]], 0)
        self.register_buffer('plane_axes', plane_axes.clone().detach())
        self.register_buffer('plane_normals', plane_normals.clone().detach())
        self.plane_direction = ['up', 'front', 'right']

        self.wavelet_type = wavelet_type
        self.inner_wavelet_scale = inner_multi_res_scale
        self.inner_wavelet_viewdir_scale = inner_multi_res_viewdir_scale
        self.inner_multi_res_scale_current = inner_multi_res_scale_current
        self.wavelet_base_resolution = wavelet_base_resolution
        assert self.inner_wavelet_scale >= self.inner_multi_res_scale_current

        self.init_plane_features(planes_features)

        # :72-94 (parameters are created in the reference's order: state-dict / optimiser layout)
        self.learn_rotation_axis = learn_rotation_axis
        self.rotation_matrix = None
        if self.learn_rotation_axis:
            self.rotation_matrix = nn.Parameter(torch.randn(number_of_features, 3, 3))
            self.register_buffer('eye_matrix', torch.eye(3).unsqueeze(0))
        self.dropout = None
        if (dropout > 0) and (dropout < 1):
            self.dropout = nn.Dropout(dropout)
        self.lbound_auto_scale = lbound_auto_scale
        self.lbound_scale = None
        if self.lbound_auto_scale:
            self.lbound_scale = nn.Parameter(0.5 * torch.ones(3))
        self.upscale_ratio_bound = upscale_ratio_bound
        self.upscale_levels = upscale_levels
        self.init_upscale()

    def init_upscale(self):
        # reference: triplane_encoder.py:98-129 -- nested zoom planes: level k refines the central
        # upscale_ratio_bound^(k+1) part of the volume with one more wavelet level of its own
        self.upscale_enabled = False
        if 0 < self.upscale_ratio_bound < 1:
            assert self.upscale_levels > 0
            self.upscale_enabled = True
            plane_resolution = self.plane_resolution
            wavelets, self.upscale_base_resolution_lst = [], []
            self.upscale_base_corner_lst, self.upscale_bound_ratio_lst = [], []
            for level in range(self.upscale_levels):
                base = round(plane_resolution * self.upscale_ratio_bound)
                assert plane_resolution % base == 0
                corner = round(plane_resolution / 2 - base / 2)
                plane_resolution = 2 * base
                self.upscale_bound_ratio_lst.append(self.upscale_ratio_bound ** (level + 1))
                self.upscale_base_resolution_lst.append(base)
                self.upscale_base_corner_lst.append(corner)
                wavelets.append(nn.Parameter(torch.zeros(3, self.number_of_features, 3, base, base)))
            self.upscale_wavelet_lst = nn.ParameterList(wavelets)

    def is_plain(self):
        """True when the lookup is the plain three-plane bilinear one over a single set of planes: the case the fused
        field kernels and TrainStep are built for."""
        return not (self.learn_rotation_axis or self.lbound_auto_scale or self.upscale_enabled
                    or self.apply_activation_on_features)

    def get_params2(self, lr):
        # reference: triplane_encoder.py:135-151 (10x learning rate for lbound_scale)
        res_1 = [p for n, p in self.named_parameters() if 'lbound_scale' in n]
        res_2 = [p for n, p in self.named_parameters() if 'lbound_scale' not in n]
        return [{'params': res_1, 'lr': 10 * lr}, {'params': res_2, 'lr': lr}]

    def init_plane_features(self, planes_features):
        # reference: triplane_encoder.py:155-231.  The reference runs a real forward DWT of a ones tensor only
        # to learn the coefficient shapes (:188-203); they are (3,C,3,n_i,n_i) with n_i = base * 2^i,
        # base = plane_resolution / inner_multi_res_scale.
        R, C = self.plane_resolution, self.number_of_features
        self.last_used_planes = None
        self._planes_tm = None
        self._planes_tm_window = None
        # a checkpoint reads every coefficient: deferred optimiser passes over them are replayed first
        self.register_state_dict_pre_hook(
            lambda module, prefix, keep_vars: _flush_deferred_optimisers(module.planes_features_wavelet_coefs))
        # ... and load_state_dict() overwrites them: steps still pending would otherwise be replayed later, with the old
        # run's moments, on top of the loaded values (the reference loads the model before the optimiser, utils.py:1482-1510)
        self._register_load_state_dict_pre_hook(
            lambda *a, _self=weakref.ref(self): (_self() is not None and
                                                 _flush_deferred_optimisers(_self().planes_features_wavelet_coefs)) and None)
        self.window_provider = None      # callable -> occupancy window or None (see _autograd_window); set by NeRFNetwork
        self.windowed_autograd = WINDOWED_AUTOGRAD     # opt-in: see _autograd_window
        if self.inner_wavelet_scale <= 1:
            if planes_features is None:
                planes_features = self.init_sigma * torch.randn(3, C, R, R)
            self.planes_features = nn.Parameter(planes_features.clone().detach())
            self.planes_features_wavelet_all_level = 0
            self.planes_features_wavelet_coefs = nn.ParameterList([])
            self.planes_features_wavelet_pad = 0
            self.wave_id = -1
            return
        if self.wavelet_type not in WAVELET_IDS:
            raise KeyError(self.wavelet_type)
        levels = utils.get_levels(self.inner_wavelet_scale)
        if R % (2 ** levels) != 0:
            raise ValueError('plane_resolution must be divisible by inner_multi_res_scale')
        self.wave_id = WAVELET_IDS[self.wavelet_type]
        self.idwt = _IDWTBuffers(self.wavelet_type)
        pad = self.planes_features_wavelet_pad = PAD_DICT[self.wavelet_type]
        # Level sizes, fine -> coarse, as the reference's shape probe finds them (:188-203): a zero-mode analysis step
        # of an n-sample axis gives n/2 + 2*pad samples, cropped back to n/2 -- except at levels whose uncropped size
        # does not exceed wavelet_base_resolution (default 0: every level is cropped, n_i = base * 2^i).
        sizes, n = [], R
        for _ in range(levels):
            n = (n + 4 * pad + 1) // 2          # floor((n + L - 1) / 2) with L = 4 * pad + 2
            if pad > 0 and n > self.wavelet_base_resolution:
                n -= 2 * pad
            sizes.append(n)
        sizes = sizes[::-1]
        base = sizes[0]
        self.planes_features_wavelet_yh_shapes = [torch.Size((3, C, 3, m, m)) for m in sizes]
        if planes_features is None:
            planes_features = self.init_sigma * torch.randn(3, C, base, base)
        self.planes_features = nn.Parameter(planes_features.clone().detach())
        # :220-227: the finest get_levels(inner_multi_res_scale_current) levels are not learnable (zero detail)
        self.planes_features_wavelet_current_level = utils.get_levels(self.inner_multi_res_scale_current)
        self.planes_features_wavelet_all_level = levels
        n_learn = levels - self.planes_features_wavelet_current_level
        self.planes_features_wavelet_coefs = nn.ParameterList(
            [nn.Parameter(torch.zeros(s)) for s in self.planes_features_wavelet_yh_shapes[:n_learn]])

    fused_l1_views = True    # get_wavelet_features(): views whose .abs().mean() is one fused pass (False: the bare parameters)

    def get_wavelet_features(self):
        if self.inner_wavelet_scale <= 1:
            return []
        coefs = list(self.planes_features_wavelet_coefs)
        if self.fused_l1_views and torch.is_grad_enabled() and all(_AbsMean.usable(c) for c in coefs):
            views = [c.as_subclass(_CoefView) for c in coefs]
            for v, c in zip(views, coefs):
                v._tnl_l1_sink = getattr(c, "_tnl_l1_sink", None)     # set by optim.FusedAdamL1 on its parameters
            return views
        return coefs

    def get_wavelet_features_upscaled(self):
        return self.upscale_wavelet_lst if self.upscale_enabled else []

    def get_lbound_scale(self):
        # reference: triplane_encoder.py:304-312
        if self.lbound_scale is None:
            return None
        return torch.exp(self.lbound_scale.abs())

    def _autograd_window(self, max_res=-1, max_scale=-1, get_all_resolutions=False):
        """The occupancy window get_planes() may restrict a DIFFERENTIABLE rebuild to, or None (whole planes).
        Opt-in (self.windowed_autograd; trinerflet_amd.install_dropin() turns the module default WINDOWED_AUTOGRAD on):
        the contract is then "while autograd records, get_planes() is valid, and differentiable, inside the occupancy
        window only" -- which is all the reference's training loop asks of it (reconstruction/nerf/utils.py:1138-1140
        discards the result; the renderer samples marched positions, all inside the window).  A caller that reads or
        differentiates the whole array under autograd must leave the option off or use get_planes_whole().
        window_provider (set by NeRFNetwork: its density grid's window at this plane resolution) is only consulted while
        autograd records; under no_grad (evaluation, save_triplane, the density-grid refresh) planes are always whole,
        and this module's own readers (forward(), get_planes_texel_major()) upgrade a windowed cache when they need more.
        The windowed result is UNINITIALISED outside the window (see _IDWTChainWin)."""
        prov = self.window_provider
        if (prov is None or not self.windowed_autograd or not torch.is_grad_enabled() or not self.planes_features.requires_grad or get_all_resolutions
                or max_res > 0 or max_scale > 0 or not self.is_plain() or self.upscale_enabled
                or self.apply_activation_on_features or self.inner_wavelet_scale <= 1
                or self.planes_features_wavelet_all_level != len(self.planes_features_wavelet_coefs)
                or (self.wavelet_base_resolution and self.planes_features_wavelet_pad > 0)):
            return None
        n0, J = self.planes_features.shape[-1], len(self.planes_features_wavelet_coefs)
        if n0 < 32 or (n0 & (n0 - 1)) != 0 or (n0 << J) % 64 != 0 or not self.planes_features.is_cuda:
            return None
        return prov()

    def get_planes_whole(self):
        """get_planes() with every texel valid: a windowed cached result is replaced by a whole, still differentiable
        rebuild (the caller may be inside no_grad -- the density-grid refresh -- while the iteration's render, which
        follows, must reach the parameters through the cached planes)."""
        self._upgrading = True
        try:
            planes = self.get_planes()
            if getattr(planes, "_tnl_window", None) is None:
                return planes
            prov, self.window_provider = self.window_provider, None
            try:
                self.reset_cahce()
                with torch.enable_grad():
                    return self.get_planes()
            finally:
                self.window_provider = prov
        finally:
            self._upgrading = False

    def build_planes(self, get_all_resolutions=False, max_res=-1, max_scale=-1, planes_features=None, coefs=None,
                     all_level=None, inner_wavelet_scale=None):
        # reference: triplane_encoder.py:364-405
        all_res = []
        current_scale = 1
        _flush_deferred_optimisers(self.planes_features_wavelet_coefs)      # whole planes read every coefficient
        x = self.planes_features if planes_features is None else planes_features
        coefs = self.planes_features_wavelet_coefs if coefs is None else coefs
        all_level = self.planes_features_wavelet_all_level if all_level is None else all_level
        inner = self.inner_wavelet_scale if inner_wavelet_scale is None else inner_wavelet_scale
        if inner > 1:
            for level_idx in range(all_level):
                if get_all_resolutions:
                    all_res.append(x)
                if ((max_res > 0) and (min(x.shape[2:]) >= max_res)) or ((max_scale > 0) and (current_scale >= max_scale)):
                    break
                if level_idx < len(coefs):
                    yh = coefs[level_idx]
                else:   # a level that is not learnable (yet): zero detail coefficients (:387-389)
                    n = x.shape[-1]
                    yh = torch.zeros(x.shape[0], x.shape[1], 3, n, n, dtype=x.dtype, device=x.device)
                n_in = x.shape[-1]
                x = _IDWTLevel.apply(x, yh, self.wave_id)
                if n_in < self.wavelet_base_resolution and self.planes_features_wavelet_pad > 0:
                    # :391-393: below wavelet_base_resolution the level is synthesised WITHOUT the zero halo of `pad`
                    # samples: the unpadded transposed convolution is the padded one minus K = (L - 2) / 2 = 2 * pad
                    # output samples per side (2n - L + 2 instead of 2n), i.e. a crop of the same kernel's result
                    k = 2 * self.planes_features_wavelet_pad
                    x = x[..., k:-k, k:-k]
                current_scale *= 2
            if get_all_resolutions:
                all_res.append(x)
        if self.apply_activation_on_features:
            x = torch.tanh(x)
        return x, all_res

    def get_planes(self, max_res=-1, max_scale=-1, get_all_resolutions=False):
        # reference: triplane_encoder.py:407-439 (note: like the reference, a cached result is returned
        # regardless of the arguments)
        if self.last_used_planes is not None:
            if (getattr(self.last_used_planes, "_tnl_window", None) is not None and not torch.is_grad_enabled()
                    and not self._upgrading):
                return self.get_planes_whole()       # a reader outside autograd gets every texel
            return self.last_used_planes
        # what an earlier windowed backward left on the parameters describes THAT gradient: a rebuild starts a new one (the
        # caching allocator hands a freed gradient's address to the next one, so the address check alone would not tell
        # after model.zero_grad(), which -- unlike FusedAdamL1.zero_grad -- does not clear the mark)
        for prm in self.planes_features_wavelet_coefs:
            if getattr(prm, "_tnl_live", None) is not None:
                prm._tnl_live = None
        window = self._autograd_window(max_res, max_scale, get_all_resolutions)
        if window is not None:
            key = tuple(int(v) for v in window)
            if key != getattr(self, "_last_autograd_window", None):
                # a new window reads coefficients the previous one did not: an optimiser that kept their updates deferred
                # (optim.FusedAdamL1) replays them BEFORE this rebuild
                _flush_deferred_optimisers(self.planes_features_wavelet_coefs)
                self._last_autograd_window = key
            planes = _IDWTChainWin.apply(self.wave_id, window, self.planes_features, *self.planes_features_wavelet_coefs)
            planes._tnl_window = tuple(int(v) for v in window)      # read by nerf/network.py and get_planes_texel_major
            self.last_used_planes = planes
            self._planes_tm = None
            self._planes_tm_window = None
            return planes
        planes, all_res = self.build_planes(get_all_resolutions, max_res, max_scale)
        self.last_used_planes = planes
        self._planes_tm = None
        self._planes_tm_window = None
        if self.upscale_enabled:
            # :417-436: level k = the central crop of level k-1, refined by one IDWT level with its own wavelets
            up = planes
            planes, all_res = [planes], [all_res]
            for level in range(self.upscale_levels):
                c0, w = self.upscale_base_corner_lst[level], self.upscale_base_resolution_lst[level]
                up = up[:, :, c0:c0 + w, c0:c0 + w]
                up, all_up = self.build_planes(get_all_resolutions, max_res, max_scale, planes_features=up,
                                               coefs=[self.upscale_wavelet_lst[level]], all_level=1,
                                               inner_wavelet_scale=2)
                planes.append(up)
                all_res.append(all_up)
            self.last_used_planes = planes
        if get_all_resolutions:
            return all_res
        return planes

    def get_planes_texel_major(self, window=None):
        """[3,R,R,C] copy of get_planes() in `plane_dtype`, cached with it (what the samplers read).
        window (8 ints, occupancy.window_from_bounds; only honoured without autograd through the copy): the caller
        promises to read nothing outside it -- a cached whole copy is served as it is, otherwise only the window is
        converted and the copy is remembered as partial (a later request for more rebuilds it)."""
        if not self.is_plain():
            raise RuntimeError("the texel-major fast path only exists for the plain three-plane lookup")
        # A copy made under no_grad (the density-grid refresh queries the field inside @torch.no_grad; TrainStep installs
        # its own) must not be served to a later differentiable lookup: the planes would silently get no gradient.
        want_grad = torch.is_grad_enabled() and self.planes_features.requires_grad
        if want_grad:
            window = None
        window = tuple(int(v) for v in window) if window is not None else None
        have = self._planes_tm is not None and (self._planes_tm_window is None or self._planes_tm_window == window)
        if not have or (want_grad and not self._planes_tm.requires_grad):
            self._upgrading = True               # (a windowed cache is fine here when it is the window asked for)
            try:
                planes = self.get_planes()
            finally:
                self._upgrading = False
            if getattr(planes, "_tnl_window", None) is not None and planes._tnl_window != window:
                planes = self.get_planes_whole()     # only a window of the cached planes exists, and not the one asked for
            self._planes_tm = _ToTexelMajor.apply(planes, self.plane_dtype == torch.float16, window)
            self._planes_tm_window = window
        return self._planes_tm

    def reset_cahce(self):
        self.last_used_planes = None
        self._planes_tm = None
        self._planes_tm_window = None

    def _project(self, plane_axes, coords):
        # project_into_planes (:293-300): [N,dim] -> [N,Np,dim-1]
        return torch.matmul(plane_axes.transpose(-1, -2).unsqueeze(0), coords.unsqueeze(-1).unsqueeze(1)).squeeze(-1)

    def sample_from_planes_aux(self, coordinates, plane_features, plane_axes, lbound=1):
        # reference: triplane_encoder.py:314-332 -> [N, Np, C]
        if not self.lbound_auto_scale:
            tm = _ToTexelMajor.apply(plane_features, self.plane_dtype == torch.float16)
            return _Sample.apply(tm, coordinates, float(lbound)).view(coordinates.shape[0], 3, -1)
        # per-plane learnable zoom of the projected coordinates (:323-326), then the general HIP lookup
        proj = self._project(plane_axes, coordinates / lbound)                                   # N,Np,2
        proj = (proj * self.get_lbound_scale().view(1, -1, 1)).clamp(-1, 1)
        tm = _ToTexelMajor.apply(plane_features, self.plane_dtype == torch.float16)
        return _GridSample.apply(tm, proj.unsqueeze(2)).view(coordinates.shape[0], 3, -1)

    def sample_from_planes_aux_rotation(self, coordinates, plane_features, plane_axes, lbound=1):
        # reference: triplane_encoder.py:335-362: every channel samples its own rotated copy of the three axes
        Np, C, H, W = plane_features.shape
        dim = plane_axes.shape[1]
        rot = torch.matmul(self.rotation_matrix.transpose(1, 2), self.rotation_matrix) + 1e-6 * self.eye_matrix
        rot, _ = torch.linalg.qr(rot)                                                    # C,dim,dim
        axes = torch.matmul(rot.unsqueeze(1), plane_axes.unsqueeze(0)).transpose(0, 1)   # Np,C,dim,dim-1
        axes = axes.reshape(-1, dim, dim - 1)
        proj = self._project(axes, coordinates / lbound)                                 # N,Np*C,2
        tm = _ToTexelMajor.apply(plane_features, self.plane_dtype == torch.float16)
        return _GridSample.apply(tm, proj.view(coordinates.shape[0], Np, C, 2)).view(coordinates.shape[0], Np, C)

    def sample_from_planes(self, coordinates, plane_features=None, lbound=None):
        # reference: triplane_encoder.py:443-484 -> [N, 3, C]
        if lbound is None:
            lbound = self.lbound
        if self.is_plain():
            if plane_features is None:
                tm = self.get_planes_texel_major()
            else:
                tm = _ToTexelMajor.apply(plane_features, self.plane_dtype == torch.float16)
            feats = _Sample.apply(tm, coordinates, float(lbound))
            return feats.view(coordinates.shape[0], 3, -1)
        if plane_features is None:
            plane_features = self.get_planes()
        N = coordinates.shape[0]
        if self.learn_rotation_axis:
            return self.sample_from_planes_aux_rotation(coordinates, plane_features, self.plane_axes, lbound)
        if not self.upscale_enabled:
            return self.sample_from_planes_aux(coordinates, plane_features, self.plane_axes, lbound)
        # nested zoom planes (:454-483): a point uses the finest level whose box contains it
        all_planes, base = plane_features, plane_features[0]
        cmax = coordinates.abs().max(dim=-1).values
        res = torch.zeros(N, base.shape[0], base.shape[1], device=base.device, dtype=base.dtype)
        used = None
        for level in range(self.upscale_levels):
            lb = self.upscale_bound_ratio_lst[level] * lbound
            if level < self.upscale_levels - 1:
                flag = torch.logical_and(cmax <= lb, cmax > self.upscale_bound_ratio_lst[level + 1] * lbound)
            else:
                flag = cmax <= lb
            res[flag] = self.sample_from_planes_aux(coordinates[flag], all_planes[level + 1], self.plane_axes, lb) \
                .to(res.dtype)
            used = flag if used is None else torch.logical_or(used, flag)
        rest = ~used
        res[rest] = self.sample_from_planes_aux(coordinates[rest], base, self.plane_axes, lbound).to(res.dtype)
        return res

    def get_grid_features(self, grid_res, plane_features=None, grid=None):
        # reference: triplane_encoder.py:486-510
        if grid is None:
            ax = torch.arange(grid_res)
            gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing='xy')
            grid = torch.stack([gx, gy, gz], dim=-1) / (grid_res - 1)
        assert grid.max() <= 1 and grid.min() >= 0
        grid = 2 * self.lbound * grid - self.lbound
        grid = grid[..., [2, 0, 1]]
        if plane_features is None:
            plane_features = self.get_planes(2 * grid_res)
        ref = plane_features[0] if isinstance(plane_features, (list, tuple)) else plane_features
        grid = grid.to(device=ref.device, dtype=ref.dtype)
        shape = grid.shape
        feats = self.sample_from_planes(grid.view(-1, shape[-1]), plane_features=plane_features)
        return self.lbound, feats.view(*shape[:-1], -1), grid

    def get_params(self, opt_cfg):
        return self.parameters()

    def forward(self, coordinates, bound):
        # reference: triplane_encoder.py:523-530
        sampled_vals = self.sample_from_planes(coordinates, lbound=bound)
        res = sampled_vals.view(sampled_vals.shape[0], -1)
        if self.dropout is not None:
            res = self.dropout(res)
        return res
