"""Autograd wrapper of the fused field kernels (csrc/field.hip, csrc/field_bwd.hip).

sigma, rgb = fused_field(planes_tm, xyz, dirs, W0..W4, bound) == NeRFNetwork.forward
(reconstruction/nerf/network.py:118-147) with the layers in fp16 MFMA / fp32 accumulate.
"""
import ctypes as C

import torch
from torch.autograd import Function

from .. import _lib as L

SUPPORTED = {(16, 64), (32, 64), (48, 128)}


def supported(C, Hd, Hc):
    return Hd == Hc and (C, Hd) in SUPPORTED


def pack_weights(W0, W1, W2, W3, W4, C, H):
    """fp32 nn.Linear weights -> fp16 MFMA fragment buffer (tnl_field_pack)."""
    lib = L.lib()
    nbytes = lib.tnl_field_packed_bytes(L.u32(C), L.u32(H), L.u32(H))
    if nbytes == 0:
        raise NotImplementedError(f"fused field: unsupported (channels={C}, hidden={H})")
    packed = torch.empty(nbytes // 2, dtype=torch.float16, device=W0.device)
    ws = [w.detach().to(torch.float32).contiguous() for w in (W0, W1, W2, W3, W4)]
    L.check(lib.tnl_field_pack(*[L.ptr(w) for w in ws], L.u32(C), L.u32(H), L.u32(H), L.ptr(packed), L.stream()),
            "field_pack")
    return packed


def field_forward(planes_tm, xyz, dirs, packed, bound, C, R, H, save_feats=False, geo_out=False, m_actual=None,
                  zero_tail=False):
    """Raw call.  dirs=None -> density only (returns sigma, geo[M,15] or None, feats).  m_actual (device int32): rows
    from m_actual[0] on are not computed; their outputs are zeros with zero_tail, undefined otherwise."""
    M = xyz.shape[0]
    dev = xyz.device
    new = torch.zeros if (zero_tail and m_actual is not None) else torch.empty
    sigma = new(M, dtype=torch.float32, device=dev)
    if dirs is None:
        second = new(M, 15, dtype=torch.float32, device=dev) if geo_out else None
    else:
        second = new(M, 3, dtype=torch.float32, device=dev)
    feats = None
    if save_feats:
        # [M,3C] features; for hidden 128 the 16 sigma-net outputs per sample follow in the same allocation (kept alive by
        # the view) for the colour half of the split backward
        nb = L.lib().tnl_field_feats_save_bytes(L.u32(M), L.u32(C), L.u32(H))
        feats = torch.empty(nb // 2, dtype=torch.float16, device=dev)   # opaque: blocked by 32-sample tiles (field_common.h)
    L.check(L.lib().tnl_field_forward(L.ptr(planes_tm), L.i32(int(planes_tm.dtype == torch.float16)), L.ptr(xyz),
                                      L.ptr(dirs), L.f32(bound), L.u32(M), L.u32(C), L.u32(R), L.u32(H), L.u32(H),
                                      L.ptr(packed), L.ptr(sigma), L.ptr(second), L.ptr(feats), L.ptr(m_actual), L.stream()),
            "field_forward")
    return sigma, second, feats


def field_backward(grad_sigma, grad_rgb, sigma, rgb, feats, xyz, dirs, packed, bound, C, R, H, grad_tm, gradW,
                   m_actual=None, dfeat=None):
    """Raw call: accumulates into grad_tm [3,R,R,C] fp32 and gradW (concatenated W0..W4, fp32)."""
    lib = L.lib()
    M = xyz.shape[0]
    nws = lib.tnl_field_backward_workspace(L.u32(M), L.u32(C), L.u32(H), L.u32(H))
    ws = torch.empty(max(nws, 4), dtype=torch.uint8, device=xyz.device)
    L.check(lib.tnl_field_backward(L.ptr(grad_sigma), L.ptr(grad_rgb), L.ptr(sigma), L.ptr(rgb), L.ptr(feats),
                                   L.ptr(xyz), L.ptr(dirs), L.f32(bound), L.u32(M), L.u32(C), L.u32(R), L.u32(H),
                                   L.u32(H), L.ptr(packed), L.ptr(grad_tm), L.ptr(gradW), L.ptr(ws), L.ptr(m_actual), L.ptr(dfeat),
                                   L.stream()),
            "field_backward")


_SIDE = {}


def _side_stream(device):
    """One side stream per device for work a forward starts on behalf of its backward (see _FusedField.forward)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _FusedField(Function):
    binned_backward = True     # False: the plane gradient by global float atomics (tests compare the two)
    early_sort = True          # the backward's tile sort is started by the forward, on a side stream (6.98 -> 6.87 ms per step)
    deterministic = False      # tile lists ordered by sample id before they are reduced (order_tile_lists): two runs on the
    #                            same inputs give the same bits (tests; the default order is the arrival order of atomics)

    @staticmethod
    def forward(ctx, planes_tm, xyz, dirs, W0, W1, W2, W3, W4, bound, planes_cm=None, m_actual=None, window=None):
        """planes_cm (optional): the (3,C,R,R) planes the texel-major copy `planes_tm` was made from.  When given, the
        planes' gradient is returned for IT, already in its layout (the tile reduction writes channel-major directly),
        and planes_tm is read as plain data -- the layout pass back to (3,C,R,R) (0.74 ms at base) disappears.
        m_actual (optional, device int32): the march's sample count; rows from there on (the zero padding up to the
        sample budget, raymarching.cu:312-480) are skipped -- outputs 0, no gradient -- in every kernel.  They are worth
        nothing to the result (no ray owns them) but all sit on the texel of the origin: one tile's list in the
        plane-gradient reduction then holds 2 % of the batch and its workgroup runs 1 ms after all others have finished.
        window (optional, 8 ints; needs planes_cm): the occupancy window planes_cm was built for
        (triplane_encoder._IDWTChainWin) -- every sample lies inside it.  The gradient returned for planes_cm is then
        written inside the window only (no zero fill of the whole planes; the rest is UNINITIALISED), which is all its
        producer's backward reads."""
        L.require_cuda(planes_tm, xyz, dirs, W0)
        _, R, _, C = planes_tm.shape
        H = W0.shape[0]
        xyz = xyz.detach().to(torch.float32).contiguous()
        dirs = dirs.detach().to(torch.float32).contiguous()
        packed = pack_weights(W0, W1, W2, W3, W4, C, H)
        need_grad = any(ctx.needs_input_grad)
        if m_actual is not None:
            m_actual = m_actual.reshape(-1)[:1].to(torch.int32).clone()    # the caller's counter is a ring slot
        ctx.m_actual = m_actual
        ctx.window = [int(v) for v in window] if (window is not None and planes_cm is not None) else None
        ctx.sort = None
        if (need_grad and _FusedField.early_sort and R % 32 == 0 and _FusedField.binned_backward and xyz.shape[0] > 0
                and ctx.needs_input_grad[0 if planes_cm is None else 9]):
            # the backward's tile sort needs the positions only: it runs NOW on a side stream, underneath this forward
            # and the compositing (0.5 ms of atomics-bound passes that the in-line backward would wait for)
            cur = torch.cuda.current_stream()
            side = _side_stream(xyz.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ws = plane_grad_sort(xyz, float(bound), R, m_actual)
                if _FusedField.deterministic:
                    order_tile_lists(ws, R, xyz.shape[0])
                ev = torch.cuda.Event()
                ev.record()
            ws.record_stream(cur)                       # allocated under the side stream, consumed by the backward
            for t_ in (xyz,) + ((m_actual,) if m_actual is not None else ()):
                t_.record_stream(side)
            ctx.sort = (ws, ev)
        sigma, rgb, feats = field_forward(planes_tm, xyz, dirs, packed, float(bound), C, R, H, save_feats=need_grad,
                                          m_actual=m_actual, zero_tail=True)
        ctx.save_for_backward(xyz, dirs, packed, sigma, rgb, feats)
        ctx.dims = (C, R, H, float(bound), [tuple(w.shape) for w in (W0, W1, W2, W3, W4)])
        ctx.cm = planes_cm is not None and R % 32 == 0 and _FusedField.binned_backward
        if planes_cm is not None and not ctx.cm:
            raise NotImplementedError("fused_field(planes_cm=...) needs plane_resolution % 32 == 0 (the tile reduction)")
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        xyz, dirs, packed, sigma, rgb, feats = ctx.saved_tensors
        C, R, H, bound, shapes = ctx.dims
        dev = xyz.device
        g_sigma = g_sigma.to(torch.float32).contiguous()
        g_rgb = g_rgb.to(torch.float32).contiguous()
        nw = sum(a * b for a, b in shapes)
        gradW = torch.zeros(nw, dtype=torch.float32, device=dev)
        if ctx.cm:
            win = ctx.window if xyz.shape[0] > 0 else None
            grad_cm = (torch.empty if win is not None else torch.zeros)(3, C, R, R, dtype=torch.float32, device=dev)
            if xyz.shape[0] > 0:
                dfeat = torch.empty(3, xyz.shape[0], C, dtype=torch.float16, device=dev)
                field_backward(g_sigma, g_rgb, sigma, rgb, feats, xyz, dirs, packed, bound, C, R, H, grad_cm, gradW, dfeat=dfeat,
                               m_actual=ctx.m_actual)
                kw = dict(roi=win + [C, 0], roi_in_place=True) if win is not None else dict(prezeroed=True)
                if ctx.sort is not None:
                    torch.cuda.current_stream().wait_event(ctx.sort[1])
                    plane_grad_reduce(ctx.sort[0], dfeat, xyz, bound, C, R, grad_cm, channel_major=True, **kw)
                elif _FusedField.deterministic:
                    ws = plane_grad_sort(xyz, bound, R, ctx.m_actual)
                    order_tile_lists(ws, R, xyz.shape[0])
                    plane_grad_reduce(ws, dfeat, xyz, bound, C, R, grad_cm, channel_major=True, **kw)
                else:
                    plane_grad_binned(dfeat, xyz, bound, C, R, grad_cm, m_actual=ctx.m_actual, channel_major=True, **kw)
            grad_tm = None
        elif R % 32 == 0 and xyz.shape[0] > 0 and _FusedField.binned_backward:
            # no global float atomics (round 4; before: 6.8 ms of a 24.5-ms step of the reference's loop at base, bound by
            # the memory-side atomic rate): dF leaves the field kernel as fp16, plane-major, and a tile-sorted matrix-core
            # reduction (csrc/scatter.hip) writes every texel of the plane gradient once -- what TrainStep does, here
            # with the counting sort in line.  Under GradScaler an overflow of the fp16 dF shows up as inf in the
            # gradient, which is what makes scaler.step() skip and back off.
            grad_tm = torch.zeros(3, R, R, C, dtype=torch.float32, device=dev)
            dfeat = torch.empty(3, xyz.shape[0], C, dtype=torch.float16, device=dev)
            field_backward(g_sigma, g_rgb, sigma, rgb, feats, xyz, dirs, packed, bound, C, R, H, grad_tm, gradW, dfeat=dfeat,
                           m_actual=ctx.m_actual)
            if ctx.sort is not None:
                torch.cuda.current_stream().wait_event(ctx.sort[1])
                plane_grad_reduce(ctx.sort[0], dfeat, xyz, bound, C, R, grad_tm, prezeroed=True)
            elif _FusedField.deterministic:
                ws = plane_grad_sort(xyz, bound, R, ctx.m_actual)
                order_tile_lists(ws, R, xyz.shape[0])
                plane_grad_reduce(ws, dfeat, xyz, bound, C, R, grad_tm, prezeroed=True)
            else:
                plane_grad_binned(dfeat, xyz, bound, C, R, grad_tm, m_actual=ctx.m_actual, prezeroed=True)
        else:
            grad_tm = torch.zeros(3, R, R, C, dtype=torch.float32, device=dev)
            field_backward(g_sigma, g_rgb, sigma, rgb, feats, xyz, dirs, packed, bound, C, R, H, grad_tm, gradW,
                           m_actual=ctx.m_actual)
        gws, off = [], 0
        for a, b in shapes:
            gws.append(gradW[off:off + a * b].view(a, b))
            off += a * b
        return (grad_tm, None, None, *gws, None, grad_cm if ctx.cm else None, None, None)


fused_field = _FusedField.apply


def plane_grad_sort(xyz, bound, R, m_actual=None):
    """First half of plane_grad_binned: counting sort of the samples by plane tile (positions only).  Returns the
    workspace tensor plane_grad_reduce consumes."""
    lib = L.lib()
    M = xyz.shape[0]
    nbytes = lib.tnl_plane_grad_binned_workspace(L.u32(M), L.u32(R))
    if nbytes == 0:
        raise NotImplementedError("binned plane gradient needs plane_resolution % 32 == 0")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
    L.check(lib.tnl_plane_grad_sort(L.ptr(xyz), L.f32(bound), L.u32(M), L.ptr(m_actual), L.u32(R), L.ptr(ws),
                                    L.stream()), "plane_grad_sort")
    return ws


def order_tile_lists(ws, R, M):
    """Sorts every tile's list of sample ids ascending, in place (the counting sort leaves them in the arrival order of
    its atomics, so the tile reduction's fp32 summation order -- hence the last bits of the plane gradient -- differs
    from run to run).  With this the plane gradient is a pure function of its inputs.  A test / debugging knob
    (TrainStep(deterministic=True)): a 64-bit torch.sort over all list entries, ~10 ms at the base workload."""
    lay = (C.c_int64 * 5)()
    L.check(L.lib().tnl_plane_grad_sort_layout(L.u32(M), L.u32(R), lay), "plane_grad_sort_layout")
    nb, off_idx, ent_idx, subs, pos_idx = (int(v) for v in lay)
    w32 = ws.view(torch.int32)
    offsets = w32[off_idx:off_idx + nb + 1]
    tile_off = offsets[::subs].long()                      # nb / subs + 1 boundaries (the last = total entries)
    if torch.cuda.is_current_stream_capturing():
        # no read-back inside a stream capture (TrainStep(graph=True, deterministic=True)): the whole capacity of 12 M
        # slots is sorted, the unused ones (index >= total, a device-side comparison) behind all lists
        total = 12 * M
        idx = torch.arange(total, device=ws.device)
        used = idx < tile_off[-1]
    else:
        total = int(tile_off[-1])
        if total == 0:
            return
        idx, used = torch.arange(total, device=ws.device), None
    entries = w32[ent_idx:ent_idx + total]
    pos = w32[pos_idx:pos_idx + 2 * total].view(total, 2)  # the entries' (fx, fy), moved with their ids
    seg = torch.searchsorted(tile_off[1:].contiguous(), idx, right=True)
    key = (seg << 32) | (entries.long() & 0xFFFFFFFF)
    if used is not None:
        key = torch.where(used, key, torch.full_like(key, 1 << 62))
    skey, perm = torch.sort(key)
    entries.copy_((skey & 0xFFFFFFFF).to(torch.int32))
    pos.copy_(pos[perm])


def plane_grad_sort_workspace(M, R, device):
    """Workspace of plane_grad_sort / plane_grad_reduce for M samples (uninitialised)."""
    nbytes = L.lib().tnl_plane_grad_binned_workspace(L.u32(M), L.u32(R))
    if nbytes == 0:
        raise NotImplementedError("binned plane gradient needs plane_resolution % 32 == 0")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def plane_grad_sort_counted(ws, xyz, bound, R, m_actual=None):
    """plane_grad_sort whose counting pass was done by march_rays_train(..., sort=(R, ws)): scan + fill."""
    L.check(L.lib().tnl_plane_grad_sort_counted(L.ptr(xyz), L.f32(bound), L.u32(xyz.shape[0]), L.ptr(m_actual), L.u32(R),
                                                L.ptr(ws), L.stream()), "plane_grad_sort_counted")
    return ws


def plane_grad_reduce(ws, dfeat, xyz, bound, C, R, grad_out, grad_scale=1.0, channel_major=False, nonfinite_flag=None,
                      roi=None, prezeroed=False, roi_in_place=False):
    """Second half: per-tile matrix-core reduction of dfeat over the sorted samples in `ws` (prezeroed / roi_in_place as
    in plane_grad_binned)."""
    layout = int(channel_major) | (2 if prezeroed else 0) | (4 if (roi_in_place and roi is not None) else 0)
    L.check(L.lib().tnl_plane_grad_reduce(L.ptr(dfeat), L.ptr(xyz), L.f32(bound), L.u32(xyz.shape[0]), L.u32(C), L.u32(R),
                                          L.f32(grad_scale), L.ptr(grad_out), L.i32(layout),
                                          L.ptr(nonfinite_flag), L.roi_array(roi), L.ptr(ws), L.stream()),
            "plane_grad_reduce")
    return grad_out


def plane_grad_binned(dfeat, xyz, bound, C, R, grad_out, m_actual=None, grad_scale=1.0, channel_major=False,
                      nonfinite_flag=None, roi=None, prezeroed=False, roi_in_place=False):
    """fp16 feature gradients (plane-major [3,M,C], as field_backward(dfeat=...) writes them) -> plane gradient fp32
    by tile-sorted matrix-core reduction
    (csrc/scatter.hip): [3,R,R,C], or (3,C,R,R) with channel_major=True; writes every tile of grad_out.
    roi (8 ints): only the window's tiles, grad_out compact (3C, rh, rw), channel_major required.
    prezeroed: grad_out was zero-filled by the caller; untouched tiles are skipped (whole planes: 1.39 -> 0.7 ms at base).
    roi_in_place (with roi): grad_out is the whole (3,C,R,R) array; only the window is written, every texel of it."""
    lib = L.lib()
    M = xyz.shape[0]
    nbytes = lib.tnl_plane_grad_binned_workspace(L.u32(M), L.u32(R))
    if nbytes == 0:
        raise NotImplementedError("binned plane gradient needs plane_resolution % 32 == 0")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
    L.check(lib.tnl_plane_grad_binned_roi(L.ptr(dfeat), L.ptr(xyz), L.f32(bound), L.u32(M), L.ptr(m_actual),
                                          L.u32(C), L.u32(R), L.f32(grad_scale), L.ptr(grad_out),
                                          L.i32(int(channel_major) | (2 if prezeroed else 0) | (4 if (roi_in_place and roi is not None) else 0)),
                                          L.ptr(nonfinite_flag), L.roi_array(roi),
                                          L.ptr(ws), L.stream()),
            "plane_grad_binned")
