"""TrainStep: the fused Adam(+L1) passes: whole arrays, gradient rectangles, the live / deferred split with its ring of step records and the replay."""
from .common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                     half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                     torch, types)


class OptimiserMixin:
    """Methods of TrainStep (trinerflet_amd/train.py): the fused Adam(+L1) passes: whole arrays, gradient rectangles,
    the live / deferred split with its ring of step records and the replay."""

    # ------------------------------------------------------------------------------------------
    def _adam(self, flat, lr_t, l1_coef, found_inf, inv_scale_dev, abs_sum=None, lo=0, hi=None):
        # bias corrections from the device-side count of steps actually taken (self.opt_steps): GradScaler.step does
        # not advance torch.optim.Adam's `step` on a skipped iteration, and the host never reads found_inf
        hi = flat.total if hi is None else hi
        n = hi - lo
        if n <= 0:
            return
        L.check(L.lib().tnl_adam_l1_step_dev(
            L.ptr(flat.data[lo:]), L.ptr(flat.grad[lo:]), L.ptr(flat.m[lo:]), L.ptr(flat.v[lo:]), L.u64(n),
            L.f32(lr_t), L.ptr(self.opt_steps), L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0),
            L.ptr(inv_scale_dev), L.f32(l1_coef), L.ptr(found_inf), L.ptr(abs_sum), L.i32(0), L.stream()),
            "adam_l1_step")

    def _learn_from(self):
        """Offset (in the flat coefficient buffer) of the first wavelet level that learns (min_wavelet_resolution_to_learn)."""
        F = self.frozen_levels
        return self.coef.total if F >= self.J else self.coef.offsets[F]

    def _frozen_abs_sum(self):
        """sum |coef| over the frozen levels (this rank's slices in the sharded mode, whose ranks' sums are all-reduced):
        constant while they are frozen; recomputed after model.load_state_dict()."""
        if self._frozen_abs is None:
            tot = torch.zeros(1, dtype=torch.float32, device=self.dev)
            for lvl in range(self.frozen_levels):
                q = self.coef.params[lvl].detach()
                if self.multi and self.dist_mode == "sharded":
                    s0, s1 = self._slice_range()
                    q = q.reshape(3 * self.C, -1)[s0:s1]
                tot += q.abs().sum(dtype=torch.float32)
            self._frozen_abs = tot
        return self._frozen_abs

    def _time_adam_pass(self, data, grad, m, v, reps=3):
        """Milliseconds of one k_adam_l1 pass over whole arrays with lr = 0 (nothing changes when g = m = v = 0).  The pass
        is asked to store every wavefront (zero_grad bit 1): candidate buffers and zero-initialised coefficient sets would
        otherwise take the zero fixed-point shortcut (no stores, twice as fast) and every placement would look perfect."""
        one = torch.ones(1, dtype=torch.float32, device=data.device)
        zero = torch.zeros(1, dtype=torch.float32, device=data.device)

        def run():
            L.check(L.lib().tnl_adam_l1_step_dev(
                L.ptr(data), L.ptr(grad), L.ptr(m), L.ptr(v), L.u64(data.numel()), L.f32(0.0), L.ptr(one),
                L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0), None, L.f32(0.0), L.ptr(zero), None,
                L.i32(2), L.stream()), "adam_l1_step (placement probe)")   # 2: the stores are not skipped for all-zero wavefronts
        run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps

    def _adam_scalars(self, lr_t):
        # fuse_adam path only: host-side bias correction from the iteration count (equal to the device count unless
        # GradScaler skipped a step)
        t = self.global_step + 1
        return lr_t / (1 - self.b1 ** t), math.sqrt(1 - self.b2 ** t)

    def _adam_levels(self, lr_t, l1, found_inf, inv_scale, s0, s1, rects):
        """Adam(+L1) over this rank's slices [s0, s1) of every wavelet level and of LL.  rects: per level the
        gradient-support rectangle from the windowed adjoint (None: gradients are dense)."""
        lib = L.lib()
        S, ns = 3 * self.C, s1 - s0

        def rect_step(flat, off, bands, n, rect, l1c, abs_sum):
            L.check(lib.tnl_adam_l1_step_rect(
                L.ptr(flat.data[off:]), L.ptr(flat.grad[off:]), L.ptr(flat.m[off:]), L.ptr(flat.v[off:]), L.u32(ns),
                L.u32(bands), L.u32(n), L.u32(self.C), L.u32(s0), (C_.c_int32 * 8)(*rect), L.f32(lr_t),
                L.ptr(self.opt_steps), L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0), L.ptr(inv_scale),
                L.f32(l1c), L.ptr(found_inf), L.ptr(abs_sum), L.stream()), "adam_l1_step_rect")

        if rects is None and ns == S:
            self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, lo=self._learn_from())
            if not self.freeze_ll:
                self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale)
            return
        for lvl in range(self.frozen_levels, self.J):
            n = self.coef.params[lvl].shape[-1]
            base = self.coef.offsets[lvl] + s0 * 3 * n * n
            if rects is not None:
                rect_step(self.coef, base, 3, n, rects[lvl], l1, self.abs_sum)
            else:
                self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, base, base + ns * 3 * n * n)
        n0 = self.ll.params[0].shape[-1]
        if self.freeze_ll:
            return
        if rects is not None:
            rect_step(self.ll, s0 * n0 * n0, 1, n0, rects[0], 0.0, None)
        else:
            self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale, None, s0 * n0 * n0, s1 * n0 * n0)

    def _scaler_probe(self, g0, g1, flag):
        """GradScaler.unscale_'s found_inf over g0 (+ g1) and an optional device flag; [1] float tensor."""
        probe = torch.empty(1, dtype=torch.float32, device=self.dev)
        found = torch.empty(1, dtype=torch.float32, device=self.dev)
        L.check(L.lib().tnl_scaler_probe(L.ptr(g0), L.u32(g0.numel()), L.ptr(g1), L.u32(0 if g1 is None else g1.numel()),
                                         L.ptr(flag), L.ptr(probe), L.ptr(found), L.stream()), "scaler_probe")
        if self.multi:
            dist.all_reduce(probe, group=self.pg)
            return (~torch.isfinite(probe)).to(torch.float32)
        return found

    def _adam_live_begin(self, lr_t, l1, found_inf, s0, s1, rects):
        """Opens the step of the live / deferred split: the deferred part catches up first if the regulariser's weight or
        the slice range changed, the live pieces are fixed at the first pending step, and the step's scalars go into their
        ring slot (returned) -- for the live pass, the replay, and the adjoint levels that carry the optimiser (fuse_live)."""
        lib = L.lib()
        if self._pending and self._defer_ctx != (s0, s1, l1):
            self.flush_deferred()                  # the regulariser's weight (or the slice range) changed: new period
        if self._pending == 0:
            self._live = self._live_rects(rects)
            self._live_bands = self._band_tables(self._live) if self.live_bands else [None] * self.J
            self.last_live = self._live            # kept after the flush, for reports
            self.last_live_bands = self._live_bands
            self._defer_ctx = (s0, s1, l1)
        slot = self._pending
        if self._capturing:      # the learning rate from device memory: no launch argument changes from step to step
            L.check(lib.tnl_adam_record_step_dev(L.ptr(self._ring), L.i32(slot), L.ptr(self._lr_dev), L.ptr(self.opt_steps),
                                                 L.f32(self.b1), L.f32(self.b2), L.ptr(found_inf), L.stream()),
                    "adam_record_step_dev")
        else:
            L.check(lib.tnl_adam_record_step(L.ptr(self._ring), L.i32(slot), L.f32(lr_t), L.ptr(self.opt_steps),
                                             L.f32(self.b1), L.f32(self.b2), L.ptr(found_inf), L.stream()),
                    "adam_record_step")
        return slot

    def _adam_levels_live(self, lr_t, l1, found_inf, inv_scale, s0, s1, rects, begun=None):
        """_adam_levels over the live rectangles only; the step's scalars are recorded for the replay (begun: already, in
        that ring slot, and the levels in self._fused_levels were updated by the adjoint)."""
        lib = L.lib()
        ns = s1 - s0
        slot = self._adam_live_begin(lr_t, l1, found_inf, s0, s1, rects) if begun is None else begun
        keep = [lvl for lvl in range(self.frozen_levels, self.J) if begun is None or lvl not in self._fused_levels]
        # every level (that the adjoint has not updated already) in ONE launch: its live rectangle, or the whole level where
        # nothing is deferred
        cf, K = self.coef, len(keep)
        sizes = [cf.params[lvl].shape[-1] for lvl in keep]
        offs = [cf.offsets[lvl] + s0 * 3 * n ** 2 for lvl, n in zip(keep, sizes)]
        live = [self._live[lvl] if self._live[lvl] is not None else [0, 0, 0, 0, 0, 0, n, n] for lvl, n in zip(keep, sizes)]
        flat = lambda rs: (C_.c_int32 * (8 * K))(*[x for r in rs for x in r[:8]])
        bt = [self._live_bands[lvl] for lvl in keep]
        if K:
            L.check(lib.tnl_adam_l1_step_live_bands(
                L.ptr(cf.data), L.ptr(cf.grad), L.ptr(cf.m), L.ptr(cf.v), L.u32(ns), L.u32(self.C), L.u32(s0), L.u32(K),
                (C_.c_uint64 * K)(*offs), (C_.c_uint32 * K)(*sizes), (C_.c_uint32 * K)(*([3] * K)), flat(live),
                flat([rects[lvl] for lvl in keep]),
                (C_.c_void_p * K)(*[None if b_ is None else b_[0].data_ptr() for b_ in bt]),
                (C_.c_uint32 * K)(*[0 if b_ is None else b_[1] for b_ in bt]),
                (C_.c_float * K)(*([l1] * K)), L.f32(lr_t), L.ptr(self.opt_steps), L.ptr(self._ring[4 * slot:]),
                L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0), L.ptr(inv_scale), L.ptr(found_inf),
                L.ptr(self.abs_sum), L.stream()), "adam_l1_step_live_bands")
        n0 = self.ll.params[0].shape[-1]
        ll = self.ll
        off = s0 * n0 * n0
        if self.freeze_ll:
            pass
        elif self._capturing:      # the step's scalars from the ring slot just written (the same bits)
            L.check(lib.tnl_adam_l1_step_rect_rec(
                L.ptr(ll.data[off:]), L.ptr(ll.grad[off:]), L.ptr(ll.m[off:]), L.ptr(ll.v[off:]), L.u32(ns), L.u32(1),
                L.u32(n0), L.u32(self.C), L.u32(s0), (C_.c_int32 * 8)(*rects[0]), L.ptr(self._ring[4 * slot:]),
                L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.ptr(inv_scale), L.f32(0.0),
                L.ptr(found_inf), L.ptr(None), L.stream()), "adam_l1_step_rect_rec")
        else:
            L.check(lib.tnl_adam_l1_step_rect(
                L.ptr(ll.data[off:]), L.ptr(ll.grad[off:]), L.ptr(ll.m[off:]), L.ptr(ll.v[off:]), L.u32(ns), L.u32(1),
                L.u32(n0), L.u32(self.C), L.u32(s0), (C_.c_int32 * 8)(*rects[0]), L.f32(lr_t), L.ptr(self.opt_steps),
                L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(1.0), L.ptr(inv_scale), L.f32(0.0),
                L.ptr(found_inf), L.ptr(None), L.stream()), "adam_l1_step_rect")
        self._last_slot = slot
        if any(lv is not None for lv in self._live):
            self._pending += 1
            self.deferred_steps += 1

    def flush_deferred(self):
        """Replays the pending steps for the coefficients outside the live rectangles (no-op when none are pending).
        Called by step() before a refresh / a window change, by rebuild_planes() of whole planes, and by anything that
        reads the coefficient or moment arrays (checkpoints, evaluation, sync_sharded_parameters)."""
        if self._pending == 0:
            return
        lib = L.lib()
        s0, s1, l1 = self._defer_ctx
        ns = s1 - s0
        self._ring_sums.zero_()
        for lvl in range(self.J):
            if self._live[lvl] is None:
                continue
            n = self.coef.params[lvl].shape[-1]
            base = self.coef.offsets[lvl] + s0 * 3 * n * n
            cf = self.coef
            bt = self._live_bands[lvl]
            L.check(lib.tnl_adam_l1_catchup_bands(
                L.ptr(cf.data[base:]), L.ptr(cf.m[base:]), L.ptr(cf.v[base:]), L.u32(ns), L.u32(3), L.u32(n),
                L.u32(self.C), L.u32(s0), (C_.c_int32 * 8)(*self._live[lvl]), L.ptr(None if bt is None else bt[0]),
                L.ptr(self._ring), L.i32(self._pending),
                L.f32(self.b1), L.f32(self.b2), L.f32(self.eps), L.f32(l1), L.ptr(self._ring_sums if l1 > 0 else None),
                L.stream()), "adam_l1_catchup_bands")
        if l1 > 0:
            self.deferred_reg += l1 * self._ring_sums[:self._pending].sum()
        self.last_flush_records = self._pending
        self._pending = 0
        self._live = None
        self.deferred_flushes += 1

    def pop_deferred_reg(self):
        """The L1 value (wavelet regulariser) of the replayed steps' deferred coefficients, summed over those steps
        and, in the sharded mode, over the ranks; the accumulator restarts from zero.  Add it to a sum of step losses."""
        self.flush_deferred()
        out = self.deferred_reg.clone()
        self.deferred_reg.zero_()
        if self.multi and self.dist_mode == "sharded":
            dist.all_reduce(out, group=self.pg)
        return out

    def _adam_sharded(self, lr_t, l1, found_inf, inv_scale, s0, s1):
        """Each rank updates only its (plane, channel) slices; afterwards parameters are all-gathered so the
        replicas stay identical (needed for checkpoints; the next rebuild_planes only reads the own slices)."""
        S = 3 * self.C
        for lvl in range(self.frozen_levels, self.J):
            n = self.coef.params[lvl].shape[-1]
            per = 3 * n * n
            base = self.coef.offsets[lvl]
            self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, base + s0 * per, base + s1 * per)
        n0 = self.ll.params[0].shape[-1]
        if not self.freeze_ll:
            self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale, None, s0 * n0 * n0, s1 * n0 * n0)
