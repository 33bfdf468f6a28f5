"""TrainStep: plane rebuild (IDWT) and its adjoint over whole planes, the window or this rank's slices; the banded plane-gradient exchange."""
from .common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                     half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                     torch, types)


class PlanesMixin:
    """Methods of TrainStep (trinerflet_amd/train.py): plane rebuild (IDWT) and its adjoint over whole planes, the
    window or this rank's slices; the banded plane-gradient exchange."""

    def _idwt_level_win(self, x, yh, win, s0=0, spans=None):
        """One non-finest level restricted to the window of its output (fp32, full-size array, rest undefined)."""
        x = x.detach().contiguous()
        yh = yh.detach().contiguous()
        P, Cc, n = x.shape[0], x.shape[1], x.shape[-1]
        out = torch.empty(P, Cc, 2 * n, 2 * n, dtype=torch.float32, device=x.device)
        L.check(L.lib().tnl_idwt_level_forward_spans(L.ptr(x), L.ptr(yh), L.u32(P * Cc), L.u32(n),
                                                     L.i32(self.enc.wave_id), L.ptr(out), L.i32(0),
                                                     L.roi_array(list(win) + [self.C, s0]), L.i32(1), L.ptr(spans),
                                                     L.stream()),
                "idwt_level_forward_spans")
        return out

    def _cropped(self, lvl):
        """Level lvl (input size n) is one of the uncropped-size levels of wavelet_base_resolution > 0."""
        return self.base_res > 0 and self.crop_k > 0 and self.coef.params[lvl].shape[-1] < self.base_res

    def _crop(self, x, lvl):
        if not self._cropped(lvl):
            return x
        k = self.crop_k
        return x[..., k:-k, k:-k].contiguous()

    def rebuild_planes(self, roi=False):
        """encoder.reset_cahce(); encoder.get_planes() of utils.py:1138-1140, outside autograd.
        roi=True (step() between grid refreshes): only the occupancy window of the finest level is rebuilt and
        written into the persistent texel-major array; the encoder's own plane cache is dropped."""
        enc = self.enc
        fast = (self.J > 0 and enc.plane_dtype == torch.float16 and self.C % 8 == 0 and self.R % 16 == 0)
        if self._roi_request is not None:       # a refresh step that did not reach its backward (it raised): take its window now
            self._roi = self._finish_roi()
        roi = roi and self._roi is not None and self._tm_full is not None
        if not roi:
            self.flush_deferred()     # whole planes read every coefficient
        with torch.no_grad():
            wins = self._forward_windows() if roi else [None] * self.J
            spans, plane_spans = self._forward_spans() if roi else ([None] * self.J, None)
            if self.dist_mode == "sharded":
                planes = self._rebuild_sharded(roi, wins, spans)
            else:
                x = enc.planes_features
                for lvl in range(self.J):
                    yh = enc.planes_features_wavelet_coefs[lvl]
                    if fast and lvl == self.J - 1:  # finest level written as fp16: the fp32 planes never exist
                        x = idwt_level_half_roi(x, yh, enc.wave_id, self._roi10(), spans[lvl]) if roi else \
                            idwt_level_half(x, yh, enc.wave_id)
                    elif wins[lvl] is not None:
                        x = self._idwt_level_win(x, yh, wins[lvl], spans=spans[lvl])
                    else:
                        x = _IDWTLevel.apply(x, yh, enc.wave_id)
                    x = self._crop(x, lvl)
                planes = x
            if roi:
                enc.last_used_planes = None
                enc._planes_tm = None
                enc._planes_tm_window = None
                if planes.dtype == torch.float16:
                    return half_roi_into_texel_major(planes, self._tm_full, self._roi10(), plane_spans)
                # fp32 planes: the finest level wrote its window of a full-size (3,C,R,R) array (_idwt_level_win); the
                # layout pass converts that window into the persistent fp32 [3,R,R,C] array
                L.check(L.lib().tnl_planes_to_texel_major_win(L.ptr(planes), L.u32(self.C), L.u32(self.R), L.i32(0),
                                                              L.ptr(self._tm_full), L.roi_array(self._roi10()), L.stream()),
                        "planes_to_texel_major_win")
                return self._tm_full
            if planes.dtype == torch.float16:
                # the (3,C,R,R) fp32 planes never exist on this path: only the sampler's copy is installed in the
                # encoder's cache (get_planes() rebuilds on demand; get_planes_texel_major() serves this copy)
                enc.last_used_planes = None
                enc._planes_tm = half_to_texel_major(planes)
            else:
                enc.last_used_planes = planes
                enc._planes_tm = _ToTexelMajor.apply(planes, enc.plane_dtype == torch.float16)
            enc._planes_tm_window = None          # whole copies
            self._tm_full = enc._planes_tm if self.use_roi else None
        return enc._planes_tm

    def _slice_range(self):
        return D.slice_range(3 * self.C, self.world, self.rank)

    def _rebuild_sharded(self, roi=False, wins=None, spans=None):
        """IDWT of this rank's (plane, channel) slices, then all-gather of the rebuilt slices -- in fp16 when the
        sampler's planes are fp16 (half the bytes on the wire); with roi only the occupancy window travels."""
        enc = self.enc
        spans = spans if spans is not None else [None] * self.J
        s0, s1 = self._slice_range()
        n0 = enc.planes_features.shape[-1]
        x = enc.planes_features.reshape(3 * self.C, n0, n0)[s0:s1].unsqueeze(0).contiguous()
        fast = (self.J > 0 and enc.plane_dtype == torch.float16 and self.C % 8 == 0 and self.R % 16 == 0)
        for lvl in range(self.J):
            n = x.shape[-1]
            yh = enc.planes_features_wavelet_coefs[lvl].reshape(3 * self.C, 3, n, n)[s0:s1].unsqueeze(0).contiguous()
            if fast and lvl == self.J - 1:
                x = idwt_level_half_roi(x, yh, enc.wave_id, self._roi10(s0), spans[lvl]) if roi else \
                    idwt_level_half(x, yh, enc.wave_id)
            elif wins is not None and wins[lvl] is not None:
                x = self._idwt_level_win(x, yh, wins[lvl], s0, spans[lvl])
            else:
                x = _IDWTLevel.apply(x, yh, enc.wave_id)
            x = self._crop(x, lvl)
        if roi and x.dtype == torch.float16:
            return D.all_gather_slices(x.reshape(s1 - s0, self._roi[7], self._roi[6]), self.pg)
        if roi:
            # fp32 planes: this rank's slices hold their window inside full-size arrays; the window travels compact and is
            # put back into a full-size array for the layout pass (plain copies: the fp32 planes under several GPUs are
            # the reference-precision check, not the fast path)
            r, C = self._roi, self.C
            xs = x.reshape(s1 - s0, self.R, self.R)
            mine = torch.empty(s1 - s0, r[7], r[6], dtype=torch.float32, device=x.device)
            for pl in range(3):
                a0, a1 = max(pl * C, s0), min((pl + 1) * C, s1)
                if a1 > a0:
                    mine[a0 - s0:a1 - s0] = xs[a0 - s0:a1 - s0, r[3 + pl]:r[3 + pl] + r[7], r[pl]:r[pl] + r[6]]
            allw = D.all_gather_slices(mine, self.pg)
            full = torch.empty(3, C, self.R, self.R, dtype=torch.float32, device=x.device)
            for pl in range(3):
                full[pl, :, r[3 + pl]:r[3 + pl] + r[7], r[pl]:r[pl] + r[6]] = allw[pl * C:(pl + 1) * C]
            return full
        mine = x.reshape(s1 - s0, self.R, self.R)
        return D.all_gather_slices(mine, self.pg).view(3, self.C, self.R, self.R)

    def _adjoint(self, grad_tm, g_cm=None, fuse=None, roi=None, scattered=False, live_adam=None):
        """plane gradient (texel-major [3,R,R,C], or already (3,C,R,R) in g_cm) -> coefficient / LL gradients.
        fuse=None: fills self.ll.grad / self.coef.grad (dense).  fuse=(lr_t, l1, found_inf, inv_scale): every
        level applies Adam(+L1) to its coefficients where their gradients are produced (no gradient buffer)."""
        lib = L.lib()
        C, R = self.C, self.R
        if g_cm is None:
            g_cm = torch.empty(3, C, R, R, dtype=torch.float32, device=self.dev)
            L.check(lib.tnl_planes_to_channel_major(L.ptr(grad_tm), L.u32(C), L.u32(R), L.ptr(g_cm), L.stream()),
                    "planes_to_channel_major")
        S = 3 * C
        s0, s1 = 0, S
        g = g_cm.view(S, R, R) if (roi is None and not scattered) else g_cm   # roi: compact (S, rh, rw) window of the gradient
        if scattered:            # already reduce-scattered (banded exchange): this rank's slices only
            s0, s1 = self._slice_range()
        elif self.dist_mode == "allreduce":
            dist.all_reduce(g, group=self.pg)
        elif self.dist_mode == "sharded":
            s0, s1 = self._slice_range()
            g = D.reduce_scatter_slices(g, self.pg, self.grad_transport)
        elif self.grad_transport == "bf16" and not self.multi:
            # one process: what a world of one would exchange -- its own contribution rounded to bf16 once (the form the
            # PSNR check of the transport runs in: tools/psnr_ci.py --arm bf16_transport)
            g = g.to(torch.bfloat16).to(torch.float32)
        ns = s1 - s0
        if fuse is not None:
            lr_t, l1, found_inf, inv_scale = fuse
            step_size, bias2_sqrt = self._adam_scalars(lr_t)
        adj_spans = self._adjoint_spans() if (roi is not None and fuse is None) else [None] * self.J
        # (min_wavelet_resolution_to_learn: the levels below the first one that learns -- and the LL plane -- take no step, so
        #  the adjoint stops there: their gradients are never formed)
        for lvl in reversed(range(self.frozen_levels, self.J)):
            if self._cropped(lvl):       # the level's output was cropped by k per side: its gradient is zero there
                g = torch.nn.functional.pad(g, (self.crop_k,) * 4)
            n = (R >> (self.J - lvl)) if roi is not None else g.shape[-1] // 2
            per = 3 * n * n
            dx = torch.empty(ns, n, n, dtype=torch.float32, device=self.dev) if lvl > 0 else None
            if fuse is not None:
                o = self.coef.offsets[lvl] + s0 * per
                cf = self.coef
                llp = [None, None, None]
                if lvl == 0:
                    lo = s0 * n * n
                    llp = [self.ll.data[lo:], self.ll.m[lo:], self.ll.v[lo:]]
                L.check(lib.tnl_idwt_level_backward_adam(
                    L.ptr(g), L.u32(ns), L.u32(n), L.i32(self.enc.wave_id), L.ptr(dx), L.ptr(cf.data[o:]),
                    L.ptr(cf.m[o:]), L.ptr(cf.v[o:]), L.ptr(llp[0]), L.ptr(llp[1]), L.ptr(llp[2]), L.f32(step_size),
                    L.f32(bias2_sqrt), L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0), L.ptr(inv_scale),
                    L.f32(l1), L.ptr(found_inf), L.ptr(self.abs_sum), L.stream()), "idwt_level_backward_adam")
            else:
                dyh = self.coef.grad_view(lvl).view(S, 3, n, n)[s0:s1]  # contiguous slice range of the flat buffer
                if lvl == 0:
                    dx = self.ll.grad_view(0).view(S, n, n)[s0:s1]
                if roi is not None and self._rect_ok and live_adam is not None and lvl in self._fused_levels:
                    # steady state: this level's live pieces are updated in the adjoint kernel's epilogue (fuse_live)
                    slot, l1_, found_inf_, inv_scale_ = live_adam
                    win = list(roi) if lvl == self.J - 1 else list(self._rects[lvl + 1])
                    lv, bt, cf = self._live[lvl], self._live_bands[lvl], self.coef
                    o = cf.offsets[lvl] + s0 * per
                    L.check(lib.tnl_idwt_level_backward_live_adam(
                        L.ptr(g), L.u32(ns), L.u32(n), L.i32(self.enc.wave_id), L.ptr(dx), L.roi_array(win + [C, s0]),
                        L.i32(0 if lvl == self.J - 1 else 1), (C_.c_int32 * 8)(*lv[:8]),
                        L.ptr(None if bt is None else bt[3]), L.ptr(cf.data[o:]), L.ptr(cf.m[o:]), L.ptr(cf.v[o:]),
                        L.ptr(None if bt is None else bt[0]), L.u32(0 if bt is None else lv[7] // 8),
                        L.ptr(self._ring[4 * slot:]), L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0),
                        L.ptr(inv_scale_), L.f32(l1_), L.ptr(found_inf_), L.ptr(self.abs_sum), L.stream()),
                        "idwt_level_backward_live_adam")
                elif roi is not None and self._rect_ok:
                    # gradient-support chain: the window of this level's input -> the rectangle of coarse tiles it
                    # reaches; nothing is stored outside it, the next level reads it as a strided window and the
                    # Adam pass of this level takes g = 0 outside (self._rects[lvl])
                    win = list(roi) if lvl == self.J - 1 else list(self._rects[lvl + 1])
                    rect = (C_.c_int32 * 8)()
                    L.check(lib.tnl_idwt_level_backward_spans(
                        L.ptr(g), L.u32(ns), L.u32(n), L.i32(self.enc.wave_id), L.ptr(dx), L.ptr(dyh),
                        L.roi_array(win + [C, s0]), L.i32(0 if lvl == self.J - 1 else 1), rect,
                        L.ptr(adj_spans[lvl] if lvl > 0 else None), L.stream()),
                        "idwt_level_backward_spans")
                    if adj_spans[lvl] is not None and list(rect) != self._rects[lvl]:
                        raise RuntimeError("the adjoint's rectangle changed under an unchanged occupancy window")
                    self._rects[lvl] = list(rect)
                else:
                    lvl_roi = L.roi_array(list(roi) + [C, s0]) if (roi is not None and lvl == self.J - 1) else None
                    L.check(lib.tnl_idwt_level_backward_roi(L.ptr(g), L.u32(ns), L.u32(n), L.i32(self.enc.wave_id),
                                                            L.ptr(dx), L.ptr(dyh), lvl_roi, L.stream()),
                            "idwt_level_backward")
            g = dx
        self._rects_roi = self._roi if roi is not None else None
        return s0, s1

    def _exchange_bands(self, roi):
        """[(first row, rows)] of the bands the plane-gradient window is exchanged in, or None (one piece): overlap_exchange
        K > 1, an occupancy window, the slice-sharded mode (or a single process, where only the banded reduction's own
        cost shows: the measurement of DESIGN.md section 5), not the Adam-fused adjoint."""
        K = self.overlap_exchange
        if K == "auto":      # from the cost model, for this window and sample budget (distributed.plan_exchange)
            key = (tuple(roi) if roi is not None else None, int(self.model.mean_count))
            if self._auto_plan is None or self._auto_plan[0] != key:
                tex = self.R * self.R if roi is None else roi[6] * roi[7]
                plan = D.plan_exchange(self.world, 3 * self.C, tex, max(int(self.model.mean_count), 1),
                                       plane_bytes=self.enc.plane_dtype.itemsize if hasattr(self.enc.plane_dtype, "itemsize")
                                       else 2, transports=(self.grad_transport,))
                self._auto_plan = (key, plan)
            plan = self._auto_plan[1]
            K = plan["overlap_exchange"] if plan["mode"] == "sharded" else 0
        if K <= 1 or roi is None or self.fuse_adam or (self.multi and self.dist_mode != "sharded"):
            return None
        n64 = roi[7] // 64
        K = min(K, n64)
        if K <= 1:
            return None
        sizes = [(n64 // K + (1 if b < n64 % K else 0)) * 64 for b in range(K)]
        out, y = [], 0
        for hb in sizes:
            out.append((y, hb))
            y += hb
        return out

    def sync_sharded_parameters(self, moments=False):
        """All-gather the slice-sharded coefficients ("sharded" mode: a rank's Adam pass only updates its own
        (plane, channel) slices, the others go stale until this runs).  A collective: every rank must call it, in the
        same order.  moments=True also gathers exp_avg / exp_avg_sq (needed for a full checkpoint).  No-op when
        nothing was stepped since the last call.  In every mode it first applies the deferred part of the coefficient
        pass (flush_deferred): after it the parameter and moment arrays are what a per-step pass would have left."""
        self.flush_deferred()
        if self.dist_mode != "sharded":
            return
        need_p = self._stale_params
        need_m = moments and self._stale_moments
        if not (need_p or need_m):
            return
        s0, s1 = self._slice_range()
        S = 3 * self.C
        for flat in (self.coef, self.ll):
            for k, p in enumerate(flat.params):
                o, n = flat.offsets[k], flat.sizes[k]
                bufs = ([flat.data] if need_p else []) + ([flat.m, flat.v] if need_m else [])
                for buf in bufs:
                    seg = buf[o:o + n].view(S, -1)
                    seg.copy_(D.all_gather_slices(seg[s0:s1], self.pg))
        self._stale_params = False
        if need_m:
            self._stale_moments = False
