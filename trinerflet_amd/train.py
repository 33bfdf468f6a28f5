"""TrainStep -- own counterpart of one iteration of Trainer.train_one_epoch2 (reference
reconstruction/nerf/utils.py:1134-1175) with train_step (:532-679), the optimiser / scheduler / GradScaler
wiring of reconstruction/main_nerf.py:115-129 and decay_function (utils.py:55-62).

One step = rebuild planes (IDWT) -> [every 16 steps: density-grid refresh] -> near/far -> march -> fused
field -> composite -> MSE + wavelet-L1 -> composite backward -> fused field backward -> IDWT adjoint ->
fused Adam(+L1).  Every arithmetic stage is a kernel of libtrinerflet_hip.so; torch supplies device
memory, the stream, a handful of [N,3] elementwise ops for the loss, and the collectives.

It is numerically the same step as driving the drop-in modules through autograd with
torch.optim.Adam + GradScaler (tests/test_train_gpu.py checks both against the CPU oracle); it differs
in what is NOT materialised: no autograd graph, the L1 regulariser never forms |coef| or sign(coef)
tensors, coefficient gradients are unscaled inside the Adam pass, the parameters / Adam moments /
gradients of all wavelet levels live in three flat buffers so the whole coefficient update is one launch.

Deferred coefficient pass (defer_adam, default for >= 32 M coefficients): between two density-grid
refreshes the coefficients outside the occupancy window's footprint are updated lazily -- their steps are recorded and
replayed, bit-identically, before anything inside this package reads them (refresh, window change, rebuild_planes of whole
planes, checkpoints / evaluation through Trainer, sync_sharded_parameters).  Code that reads or replaces
model.encoder's coefficient tensors or TrainStep's moment buffers directly must call TrainStep.flush_deferred() first;
rendering through the occupancy grid is unaffected (it never samples outside the window).  A step's returned loss then
carries the L1 value of the live coefficients only; pop_deferred_reg() hands out the rest.

Multi-GPU: rays are sharded across ranks, planes and MLP weights replicated.
  mode "allreduce": plane gradients all-reduced (RCCL) before the adjoint; every rank repeats the dense work.
  mode "sharded"  : the 3*C (plane, channel) slices are the shard unit (the IDWT is depthwise): plane gradients
                    are reduce-scattered by slice, each rank runs adjoint + Adam + IDWT on 3C/G slices, and the
                    rebuilt planes are all-gathered -- same bytes on the wire, dense HBM work divided by G.
"""
import ctypes as C_
import math
import types

import numpy as np

import torch
import torch.distributed as dist

from . import _lib as L
from . import distributed as D
from . import occupancy
from . import raymarching
from .nerf import field as F_
from .triplaneencoder.triplane_encoder import (_IDWTLevel, _ToTexelMajor, half_roi_into_texel_major, half_to_texel_major,
                                                idwt_level_half, idwt_level_half_roi)


def lr_factor(it, iters, warmup_steps, sched_base=0.1, warmup_factor=1e-3, sched_exp=2.5):
    """decay_function (utils.py:55-62) with accumelate_steps = 1."""
    w = max(warmup_steps, 0)
    if it < w:
        return sched_base * warmup_factor + it * (1 - warmup_factor) / (w - 1)
    return sched_base ** (min((it - w) / iters, 1) ** sched_exp)


class _Flat:
    """Parameters re-homed as views of one flat fp32 buffer, with matching grad / exp_avg / exp_avg_sq buffers."""

    def __init__(self, params):
        self.params = list(params)
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned segment starts (the Adam kernel uses float4)
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o, n in zip(self.params, self.offsets, sizes):
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
        self.sizes = sizes

    def grad_view(self, k):
        o, n = self.offsets[k], self.sizes[k]
        return self.grad[o:o + n].view(self.params[k].shape)

    def tune_placement(self, time_pass, candidates=12, spacing=3, good_gbs=5900.0):
        """The HBM-bound Adam pass over these four arrays runs 15-20 % slower for some PLACEMENTS of them than for
        others (tools/adam_regimes.py: same kernel, same data, same virtual spacing; the time follows which physical
        allocation holds the PARAMETER array relative to the other three -- any array may play g, m or v -- comes in
        three levels (6.1 / 5.8 / 5.1 TB/s of algorithmic bytes), is the same for neighbouring allocations over runs
        of 6-14 GB of address space, and stays with an allocation for its lifetime).  So the parameter array's
        placement is chosen by measurement, once: up to `candidates` buffers, `spacing` array sizes of address space
        apart, are timed in its role with the real kernel until one reaches `good_gbs`; the fastest is kept, the
        rest goes back to the allocator.  time_pass(data, grad, m, v) -> milliseconds must not change the arrays
        (lr = 0 and g = m = v = 0 here).  Returns a report dict."""
        nbytes = 28.0 * self.total
        gbs = lambda ms: nbytes / (ms * 1e-3) / 1e9
        t0 = time_pass(self.data, self.grad, self.m, self.v)
        report = {"before_ms": round(t0, 4), "before_GBs": round(gbs(t0), 1), "tried_ms": []}
        best_t, best = t0, None
        hold = []
        if gbs(t0) < good_gbs:
            need = (spacing + 2) * self.data.numel() * 4
            for _ in range(candidates):
                if torch.cuda.mem_get_info(self.data.device)[0] < need:      # never search a device into OOM
                    report["stopped"] = "free memory"
                    break
                cand = torch.empty_like(self.data)
                hold.append(cand)
                hold.extend(torch.empty_like(self.data) for _ in range(spacing))     # spacers: move on in address space
                t = time_pass(cand, self.grad, self.m, self.v)
                report["tried_ms"].append(round(t, 4))
                if t < best_t:
                    best_t, best = t, cand
                if gbs(best_t) >= good_gbs:
                    break
        if best is not None and best_t < 0.98 * t0:
            best.copy_(self.data)
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                p.data = best[o:o + n].view(p.shape)
            self.data = best
        else:
            best_t = t0
        # Still slow with every candidate in the parameter role (seen: twelve candidates, all 2.21-2.22 ms, in the first
        # process on a box): then one of the OTHER three arrays sits badly.  The buffers already held are timed in the
        # roles of exp_avg, exp_avg_sq and the gradient in turn, the fastest adopted each time.
        if gbs(best_t) < good_gbs and hold:
            report["other_roles"] = {}
            for role in ("m", "v", "grad"):
                if gbs(best_t) >= good_gbs:
                    break
                cur = {"m": self.m, "v": self.v, "grad": self.grad}
                pick_t, pick = best_t, None
                tried = []
                for cand in hold:
                    if cand is self.data or any(cand is t_ for t_ in cur.values()):
                        continue
                    cand.zero_()          # the timing pass leaves p alone only while g = m = v = 0
                    args = dict(cur)
                    args[role] = cand
                    t = time_pass(self.data, args["grad"], args["m"], args["v"])
                    tried.append(round(t, 4))
                    if t < pick_t:
                        pick_t, pick = t, cand
                    if len(tried) >= candidates or gbs(pick_t) >= good_gbs:
                        break
                report["other_roles"][role] = tried
                if pick is not None and pick_t < 0.98 * best_t:
                    pick.copy_(cur[role])
                    setattr(self, role, pick)
                    best_t = pick_t
        report["after_ms"], report["after_GBs"] = round(best_t, 4), round(gbs(best_t), 1)
        del hold
        return report


class _StepState(types.SimpleNamespace):
    """What the stages of one TrainStep.step() hand to each other."""


class TrainStep:
    def __init__(self, model, lr=1e-2, wavelet_regularization=0.4, iters=30000, warmup_steps=0,
                 betas=(0.9, 0.99), eps=1e-15, fp16=True, update_extra_interval=16, background_color=0.0,
                 max_steps=1024, dt_gamma=0.0, T_thresh=1e-4, init_scale=65536.0, growth_interval=2000,
                 dist_mode=None, process_group=None, single_rank_collectives=False, binned=True, fuse_adam=False, use_roi=True, tune_placement=None,
                 defer_adam=None, deterministic=False, live_bands=True, overlap_exchange=0, graph=False):
        enc = model.encoder
        assert model.cuda_ray, "TrainStep drives the cuda_ray renderer (every README configuration)"
        if not model._fused_ok():
            raise NotImplementedError("TrainStep needs a configuration the fused field kernel is built for")
        # wavelet_base_resolution > 0 (triplane_encoder.py:391-393: the levels below it keep their uncropped analysis size
        # and are synthesised without the zero halo = a crop of the padded kernel's output by 2 * pad per side): the
        # level sizes are then not base * 2^i, so the occupancy window / rectangle machinery stays off and the dense
        # rebuild crops (its adjoint zero-pads) those levels.  No README configuration uses the option.
        self.base_res = int(getattr(enc, "wavelet_base_resolution", 0) or 0)
        self.crop_k = 2 * int(getattr(enc, "planes_features_wavelet_pad", 0) or 0)
        self.model, self.enc = model, enc
        self.C, self.R, self.H = enc.number_of_features, enc.plane_resolution, model.hidden_dim
        self.J = enc.planes_features_wavelet_all_level
        self.lr, self.lam, self.iters, self.warmup = lr, wavelet_regularization, iters, warmup_steps
        self.b1, self.b2, self.eps = betas[0], betas[1], eps
        self.fp16 = fp16
        self.update_extra_interval = update_extra_interval
        self.bg = background_color
        self.max_steps, self.dt_gamma, self.T_thresh = max_steps, dt_gamma, T_thresh
        self.global_step = 0
        self.binned = binned
        # deterministic: every tile's list of the plane-gradient reduction ordered by sample id before it is consumed
        # (field.order_tile_lists), so that two runs on the same inputs produce the same bits.  A test / debugging knob.
        self.deterministic = deterministic
        # fuse_adam: Adam(+L1) applied inside the adjoint IDWT kernels (no coefficient-gradient buffer).  Measured
        # SLOWER at base (3.8 ms vs 0.95 + 2.03 ms): the epilogue's 4-byte p/m/v accesses are issued late and in
        # 128-B pieces, while the stand-alone pass streams 16 B/lane at the HBM ceiling.  Kept as an option.
        self.fuse_adam = fuse_adam
        # fuse_live: in the steady state of the live / deferred split (occupancy window, its rectangles and band pieces known)
        # the column-walk levels of the adjoint apply the optimiser to their live pieces in the kernel's epilogue
        # (tnl_idwt_level_backward_live_adam): the band gradients of the two finest levels -- 81 % of the live coefficients at
        # base -- are neither written nor read back.  Bit-identical to the separate passes.
        self.fuse_live = True
        self.fuse_live_levels = 2     # how many of the finest column-walk levels (all of them: 2 at R = 2048)
        self._fused_levels = ()
        self.last_fused_levels = ()
        # use_roi: between two density-grid refreshes no sample can leave the bounding window of the occupied cells,
        # so the finest IDWT level, the fp16 layout change, the plane gradient and the finest adjoint only touch
        # that window (compact arrays).  Refresh steps rebuild whole planes (the grid update queries density
        # everywhere).  Results are bit-identical to the whole-plane step.
        self.use_roi = (use_roi and binned and not fuse_adam and self.J > 0 and self.R % 64 == 0 and self.C % 8 == 0
                        and enc.plane_dtype == torch.float16 and self.base_res == 0)
        self._roi = None          # 8 ints {ox[3], oy[3], rw, rh} or None (whole planes)
        self._roi_valid = False   # False: recompute from the bitfield before it is used
        self._roi_request = None  # (pinned host buffer, device buffer, event, ...) of a window read-back in flight
        self._roi_host = None
        self._tm_full = None      # persistent fp16 [3,R,R,C]; the ROI steps refresh its window in place
        n0 = enc.planes_features.shape[-1]
        # gradient-support chain (windowed adjoint + rectangle-aware Adam): level sizes must be powers of two
        self._rect_ok = self.use_roi and n0 >= 32 and (n0 & (n0 - 1)) == 0 and self.R == n0 << self.J
        self._rects = [None] * max(self.J, 1)
        dev = enc.planes_features.device
        self.dev = dev
        self.coef = _Flat(list(enc.planes_features_wavelet_coefs))
        self.ll = _Flat([enc.planes_features])
        self.Ws = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight,
                   model.color_net[1].weight, model.color_net[2].weight]
        self.mlp = _Flat(self.Ws)
        assert self.mlp.total == sum(w.numel() for w in self.Ws) or True
        self.coef_numel = sum(self.coef.sizes)
        # placement of the coefficient arrays chosen by measurement (see _Flat.tune_placement): default for sets large
        # enough for the Adam pass to be the step's dominant kernel
        self.placement = None
        if (tune_placement if tune_placement is not None else self.coef.total >= 64_000_000) and self.coef.total > 0:
            self.placement = self.coef.tune_placement(self._time_adam_pass)
        # defer_adam: between two density-grid refreshes the coefficients outside the occupancy window's footprint are
        # neither read (windowed plane rebuild) nor reached by a data gradient, and their Adam(+L1) update is a closed
        # recurrence in their own p, m, v and the step's scalars.  The per-step pass then covers only the live
        # rectangle of each level; the rest is replayed in registers, all pending steps in one pass, before anything
        # reads it (next refresh, checkpoint, evaluation, flush_deferred()): bit-identical p, m, v, 1/16 of the bytes.
        # The L1 value of the deferred coefficients arrives with the replay (deferred_reg / pop_deferred_reg()), so a
        # step's returned loss carries the live rectangle's share only.  Default: on for large coefficient sets.
        self.defer_adam = (defer_adam if defer_adam is not None else self.coef.total >= 32_000_000) and self._rect_ok
        self._ring = torch.zeros(16 * 4, dtype=torch.float32, device=dev)      # csrc/adam.hip AdamStepRec[16]
        self._ring_sums = torch.zeros(16, dtype=torch.float32, device=dev)
        self._pending = 0          # recorded steps not yet applied outside the live rectangles
        self._live = None          # per level the live rectangle (8 ints) or None; fixed while steps are pending
        self._live_bands = None    # per level None or the rectangle's band table (see _band_tables)
        self.last_live_bands = None
        self._band_cache = {}
        self._row_ext = None
        self._rects_roi = None     # the occupancy window self._rects were computed under
        self.last_live = None
        self._defer_ctx = None     # (s0, s1, l1) of the pending steps
        self.deferred_reg = torch.zeros((), dtype=torch.float32, device=dev)   # replayed steps' L1 value, summed
        self.deferred_steps = 0    # counters for reports
        # model.state_dict() / torch.save(model.state_dict()) read the coefficient tensors directly: the deferred part
        # catches up first (flush_deferred is not a collective).  A weak reference: the hook must not keep this object alive.
        import weakref
        me = weakref.ref(self)

        def _flush_before_state_dict(module, prefix, keep_vars):
            ts = me()
            if ts is not None and ts.model is module:
                ts.flush_deferred()
        self._sd_hook = model.register_state_dict_pre_hook(_flush_before_state_dict)
        self.deferred_flushes = 0
        self.last_flush_records = 0
        # clip_far_in_order: march each ray of a refresh step's in-order march only to its exit from the occupied cells' box
        # (raymarching.clip_fars: the same samples to the bit).  Measured at base over whole periods: 3.98 ms / step with
        # it, 3.91 without (the box + clip launches cost what the shorter count pass saves): off.  The module path's
        # in-line march (renderer.run_cuda) does use the clip.
        self.clip_far_in_order = False
        self._occ_box = None       # device [6], valid for the current density_bitfield
        # GradScaler state (torch.cuda.amp.GradScaler defaults: 2^16, x2 every 2000 clean steps, x0.5 on inf)
        self.scale = torch.full((1,), init_scale if fp16 else 1.0, dtype=torch.float32, device=dev)
        self.growth_tracker = torch.zeros(1, dtype=torch.int32, device=dev)
        self.growth_interval = growth_interval
        self.opt_steps = torch.zeros(1, dtype=torch.float32, device=dev)   # optimiser steps taken (skips excluded)
        self.abs_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
        self.inv_scale = torch.ones(1, dtype=torch.float32, device=dev)   # 1 / loss scale, written by csrc/stepstate.hip
        self.last = {}
        self.section_names = None   # see _mark
        self._mark_seq = 0
        self.overlap_march = True   # the next batch's march + tile sort on a side stream (False: in order, kernels alone)
        # overlap_exchange = K > 1 ("sharded" mode with an occupancy window): the plane-gradient window is reduced and
        # reduce-scattered in K bands of rows -- band b's collective runs on the communication stream while the tile
        # reduction of band b + 1 runs on the launch stream; slice ownership and everything downstream are unchanged
        # (the bands' results are concatenated into the [S/G, rh, rw] array the adjoint reads).  See DESIGN.md section 5.
        self.overlap_exchange = int(overlap_exchange)
        self._comm = None
        self.live_col_align = 32    # column granule of the live rectangles (see _live_rects)
        self.live_bands = live_bands
        self._side = None
        self._prefetched = None     # (key, marched tensors) of a march started for the following call
        # graph: the steady-state steps of a density-grid period (positions 1 .. 14: not the refresh step, not the step whose
        # optimiser pass fills the ring and replays it) are captured once as HIP graphs -- one per position, since the ring
        # slots are launch arguments -- and replayed while the occupancy window, its pieces and the sample budget stay what
        # they were (see _graph_step).  ~40 launches become one; measured 3.79 -> 3.52 ms for such a step at the base
        # configuration (tools/exp_graph_step.py).  Off by default; single-process only.
        self.graph = bool(graph)
        self._graphs = {}           # period position -> _StepGraph
        self._graph_key = None      # what the captured launches depend on besides the ring position
        self._graph_pool = None
        self._graph_in = None       # static input tensors the captured launches read
        self._cap_stream = None
        self._graph_done = None
        self._capturing = False
        self._lr_dev = None
        self.graph_replays = 0      # counters for reports / tests
        self.graph_captures = 0
        self._stale_params = self._stale_moments = False
        # where the following batch's march + tile sort start on the side stream (binned mode): behind the field backward
        # ("bwd"), behind the tile reduction ("reduce") or behind the adjoint IDWT ("adjoint")
        self.prefetch_at = "bwd"
        self.side_count_form = 1     # raymarching.count_form of the prefetched march (0: the wavefront-per-ray count pass)
        # workgroups of the prefetched march's emit pass and of its tile sort's fill pass (raymarching.side_caps): 2 and 1 per
        # CU.  At full width the two flood the wave slots just as the adjoint's second column-walk level (two 192-register
        # workgroups per CU) is launched: 550 us for 190 alone; capped, the step is 0.14 ms shorter (3.86-3.89 vs 4.01-4.03
        # in alternation; (0, 0) = uncapped)
        # ... where the step's tail has the slack: base -0.091 ms [-0.106, -0.077], large -0.136 [-0.22, -0.05]; at small (50 M
        # coefficients: tail 0.8 ms against a side chain of 1.0) the capped chain is the critical path: +0.097 -- uncapped below
        # 2^27 coefficients
        n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count if torch.cuda.is_available() else 256
        self.side_caps = (2 * n_cu, n_cu) if 3 * self.C * self.R * self.R >= (1 << 27) else (0, 0)
        self.post_refresh = None    # optional callable run right after every density-grid refresh
        self.section_events = None  # set to [] to record HIP events (on the launch stream) around every stage
        # distributed
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist_mode and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # multi: the distributed code path is the one that runs.  single_rank_collectives=True takes it in a process group
        # of ONE rank as well (every collective issued, on RCCL each a real reduce_scatter_tensor / all_gather_into_tensor /
        # all_reduce over the group): the way to execute the 8-GPU call sequence on a one-GPU box (tests/test_dist_gpu.py)
        self.multi = self.world > 1 or bool(single_rank_collectives and dist_mode and dist.is_initialized())
        if self.multi and self.world == 1:
            D.FORCE_COLLECTIVES = True
        self.dist_mode = dist_mode if self.multi else None
        if self.dist_mode == "sharded":
            assert (3 * self.C) % self.world == 0, "3*channels must be divisible by the world size"
        if self.multi:
            # the density-grid refresh splits the 128^3 cells (and the H^3/4 picks of a partial refresh) evenly over
            # the ranks before an all-gather of equal shards (renderer.update_extra_state)
            g3 = model.grid_size ** 3
            assert g3 % self.world == 0 and (g3 // 4) % self.world == 0, \
                f"grid_size^3 / 4 = {g3 // 4} candidate cells cannot be split evenly over {self.world} ranks"

    # ------------------------------------------------------------------------------------------
    def _mark(self, name):
        """Section boundary.  With section_events set, a HIP event is recorded here (each costs the stream 6-8 us);
        section_names limits the recording to those boundaries (None = all)."""
        self._mark_seq += 1
        if self.section_events is not None and (self.section_names is None or name in self.section_names):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()  # torch's current stream = the stream every kernel of the step is launched on
            self.section_events.append((name, ev, self._mark_seq))

    def section_times(self, stat="mean"):
        """Milliseconds per stage over the recorded steps: the mean, or with stat="median" the median over the steps
        (one step that allocates or waits does not move it).  Requires a prior torch.cuda.synchronize()."""
        if not self.section_events:
            return {}
        vals = {}
        prev = None
        for name, ev, seq in self.section_events:
            # a section is timed only when the boundary before it was recorded too (consecutive _mark calls)
            if name != "begin" and prev is not None and prev[1] == seq - 1:
                vals.setdefault(name, []).append(prev[0].elapsed_time(ev))
            prev = (ev, seq)
        if stat == "median":
            return {k: float(np.median(v)) for k, v in vals.items()}
        return {k: sum(v) / len(v) for k, v in vals.items()}

    def invalidate_roi(self):
        """Call after changing model.density_bitfield by hand (update_extra_state inside step() is tracked): the
        occupancy window is recomputed and a march already started for the following batch is dropped."""
        self.flush_deferred()
        self._roi_valid = False
        self._roi_request = None
        self._occ_box = None
        self._drop_prefetch()

    def _roi10(self, s0=0):
        return None if self._roi is None else list(self._roi) + [self.C, s0]

    def _compute_roi(self):
        """Window of the plane grid (per plane origin, common size, multiples of 64) that contains the bilinear
        footprint of every position inside an occupied cell of any cascade.  One small host read-back."""
        self._request_roi()
        return self._finish_roi()

    def _request_roi(self):
        """The device half of _compute_roi: two small kernels over the bitfield and an asynchronous copy of their 2 KB of
        results into pinned host memory, behind an event.  A refresh step issues it right after the grid update and reads
        the result (_finish_roi) only where the window is first needed -- before the plane gradient -- so the host does not
        stall the launch stream in the middle of the step (0.3 ms per refresh: profiles/r03e_refresh_step_timeline.txt)."""
        model = self.model
        self._band_cache = {}
        self._row_ext = None
        self._roi_request = occupancy.request(model.density_bitfield, model.cascade, model.grid_size, model.bound, self.R,
                                              rows=self.live_bands, host=self._roi_host)
        self._roi_host = self._roi_request[0]            # pinned, allocated once

    def _finish_roi(self):
        model = self.model
        req, self._roi_request = self._roi_request, None
        roi, self._row_ext = occupancy.finish(req, model.cascade, model.grid_size, model.bound, self.R)
        return roi

    def _forward_windows(self):
        """Per level the window of its OUTPUT that the next level needs (occupancy.level_windows)."""
        return occupancy.level_windows(self._roi, self.J, self.R)

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

    def _forward_spans(self):
        """(per level the device table of the coarse pieces whose results something reads, the plane grid's own table):
        level lvl produces the grid the next level's needed coefficients (_level_needs) live on -- the finest one the
        texels tnl_occupancy_row_extents reports -- so a coarse row group needs the union of the two output row groups
        it produces, halved.  (None, ...) where there is nothing to gain or the geometry is not the plain one."""
        if "fwd" in self._band_cache:
            return self._band_cache["fwd"]
        out = ([None] * self.J, None)
        needs = self._level_needs() if self.live_bands else None
        if needs is not None:
            big = np.int64(0x7fffffff)
            tabs = []
            for lvl in range(self.J):
                src = self._row_ext if lvl == self.J - 1 else needs[lvl + 1]
                G = self.coef.params[lvl].shape[-1] // 8
                pair = src.reshape(3, G, 2, 2)
                lo, hi = pair[..., 0].min(2), pair[..., 1].max(2)
                has = hi > lo
                tabs.append(np.stack([np.where(has, lo // 2, big), np.where(has, (hi + 1) // 2, -1)], axis=-1))
            tabs.append(self._row_ext)
            flat = np.concatenate([t.reshape(-1) for t in tabs]).astype(np.int32)
            dev = torch.from_numpy(flat).to(self.dev)
            offs = np.cumsum([0] + [t.size for t in tabs])
            parts = [dev[offs[k]:offs[k + 1]] for k in range(len(tabs))]
            out = (parts[:-1], parts[-1])
        self._band_cache["fwd"] = out
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
                return half_roi_into_texel_major(planes, self._tm_full, self._roi10(), plane_spans)
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
        if roi:
            return D.all_gather_slices(x.reshape(s1 - s0, self._roi[7], self._roi[6]), self.pg)
        mine = x.reshape(s1 - s0, self.R, self.R)
        return D.all_gather_slices(mine, self.pg).view(3, self.C, self.R, self.R)

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
            g = D.reduce_scatter_slices(g, self.pg)
        ns = s1 - s0
        if fuse is not None:
            lr_t, l1, found_inf, inv_scale = fuse
            step_size, bias2_sqrt = self._adam_scalars(lr_t)
        adj_spans = self._adjoint_spans() if (roi is not None and fuse is None) else [None] * self.J
        for lvl in reversed(range(self.J)):
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

    def _adjoint_spans(self):
        """Per level the device table of the pieces whose band gradients the optimiser pass will read (its band pieces),
        or None: known once an adjoint has run under the current occupancy window (the live rectangles derive from the
        rectangles it returns, which depend on the window alone)."""
        none = [None] * self.J
        if not (self.defer_adam and self.live_bands and self._rect_ok and self._roi is not None and
                self._rects_roi is self._roi):
            return none
        tables = self._live_bands if self._pending else self._band_tables(self._live_rects(self._rects))
        return [None if t is None else t[3] for t in tables]

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
            self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum)
            self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale)
            return
        for lvl in range(self.J):
            n = self.coef.params[lvl].shape[-1]
            base = self.coef.offsets[lvl] + s0 * 3 * n * n
            if rects is not None:
                rect_step(self.coef, base, 3, n, rects[lvl], l1, self.abs_sum)
            else:
                self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, base, base + ns * 3 * n * n)
        n0 = self.ll.params[0].shape[-1]
        if rects is not None:
            rect_step(self.ll, s0 * n0 * n0, 1, n0, rects[0], 0.0, None)
        else:
            self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale, None, s0 * n0 * n0, s1 * n0 * n0)

    # ------------------------------------------------------------------------------------------
    # One step = the stages below, in this order (each ends with a _mark: bench.py times the sections between them).
    def step(self, rays_o, rays_d, gt_rgb, noises=None, n_global_rays=None, bg_color=None, next_rays=None):
        """rays_o, rays_d: [N,3]; gt_rgb: [N,3] (already blended with the background, utils.py:574-577).
        bg_color: None (the constructor's background_color) or a per-ray [N,3] tensor (--train_rand_bg,
        utils.py:568-570).  next_rays: optional (rays_o, rays_d[, noises]) of the FOLLOWING call: its march is
        then started on the side stream underneath this step's gradient / optimiser kernels (the march
        reads only rays and the occupancy bitfield), and the next call picks it up if it is given the same
        tensors.  Returns the (unscaled) loss as a device scalar; details in self.last."""
        self.model.train()
        N = rays_o.shape[0]
        st = _StepState(rays_o=rays_o, rays_d=rays_d, gt_rgb=gt_rgb, noises=noises, bg_color=bg_color, next_rays=next_rays,
                        N=N, n_glob=n_global_rays if n_global_rays is not None else N * self.world,
                        refresh=self.update_extra_interval > 0 and self.global_step % self.update_extra_interval == 0)
        if self.graph and self._graph_eligible(st):
            return self._graph_step(st)
        self._mark("begin")
        self._stage_pickup(st)         # the march started during the previous call, or one started now on the side stream
        self._stage_planes(st)         # (replay of the deferred pass) -> plane rebuild -> [grid refresh + new window]
        self._stage_march(st)          # wait for / run the march; samples of this step
        self._stage_render(st)         # fused field, compositing, loss and its gradient w.r.t. the rendered colours
        self._stage_backward(st)       # compositing backward, fused field backward, [next batch's side work], plane gradient
        self._stage_optimise(st)       # GradScaler probe, adjoint IDWT, Adam(+L1) passes, step epilogue
        return st.loss

    def _march(self, o, d, nz, sort_stream=None, clip=False):
        """near/far -> march_rays_train (+ the tile sort of the plane gradient, which needs only the positions).  Returns
        ((counter, xyzs, dirs, deltas, rays, sort_ws), (event after the march, event after the sort)).  sort_stream: the
        sort's scan + fill passes go there (refresh steps: beside the field forward).  clip: march each ray only to its
        exit from the occupied cells' box (raymarching.clip_fars: the same samples to the bit)."""
        model, R = self.model, self.R
        nears, fars = raymarching.near_far_from_aabb(o, d, model.aabb_train, model.min_near)
        if clip:
            if self._occ_box is None:
                self._occ_box = raymarching.occupied_box(model.density_bitfield, model.cascade, model.grid_size,
                                                         float(model.bound))
            fars = raymarching.clip_fars(o, d, fars, self._occ_box)
        counter = model.step_counter[model.local_step % 16]
        counter.zero_()
        model.local_step += 1
        # with a fixed sample budget the march also counts the samples per plane tile (first pass of the tile sort)
        fused_sort = self.binned and R % 32 == 0 and model.mean_count > 0
        sort_ws = None
        if fused_sort:
            mc = model.mean_count + (128 - model.mean_count % 128)    # the wrapper's budget rule (align = 128)
            sort_ws = F_.plane_grad_sort_workspace(mc, R, self.dev)
        out = raymarching.march_rays_train(
            o, d, model.bound, model.density_bitfield, model.cascade, model.grid_size, nears, fars,
            counter, model.mean_count, True, 128, False, self.dt_gamma, self.max_steps, nz,
            model.mean_count <= 0,   # zero fill only when the buffers are sized by the worst case (first steps)
            (R, sort_ws) if fused_sort else None)
        # the field forward needs the march only; the tile sort (needed much later, by the tile reduction) gets its own event
        ev_march = torch.cuda.Event()
        ev_march.record()
        if fused_sort and sort_stream is not None:
            assert out[0].shape[0] == mc
            sort_stream.wait_event(ev_march)
            with torch.cuda.stream(sort_stream):
                F_.plane_grad_sort_counted(sort_ws, out[0], float(model.bound), R, counter)
                ev_sort = torch.cuda.Event()
                ev_sort.record()
            for t_ in (counter, *out, sort_ws):
                if torch.is_tensor(t_):
                    t_.record_stream(sort_stream)
            return (counter, *out, sort_ws), (ev_march, ev_sort)
        if fused_sort:
            assert out[0].shape[0] == mc
            F_.plane_grad_sort_counted(sort_ws, out[0], float(model.bound), R, counter)
        else:
            sort_ws = F_.plane_grad_sort(out[0], float(model.bound), R, counter) if (self.binned and R % 32 == 0) \
                else torch.empty(0, device=self.dev)
        ev_sort = torch.cuda.Event()
        ev_sort.record()
        return (counter, *out, sort_ws), (ev_march, ev_sort)

    def _march_on_side(self, o, d, nz):
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(main)
        # beside the step's kernels the count pass runs one ray per lane: a seventh of the wavefront form's instructions at
        # one wave per SIMD (3x longer alone, but it takes almost nothing from the kernels it runs next to: A/B at base,
        # wavefront form on the side stream 4.13-4.26 ms per step at every start position vs 3.9)
        with torch.cuda.stream(self._side), raymarching.count_form(self.side_count_form), raymarching.side_caps(*self.side_caps):
            out = self._march(o, d, nz)
        for t_ in out[0]:
            t_.record_stream(main)
        return out

    def _stage_pickup(self, st):
        # The march (one ray per lane, latency-bound, ~1/8 of the chip's wave slots) depends only on the rays and
        # the occupancy bitfield, not on the planes: it runs on a side stream.  On grid-refresh steps the bitfield
        # changes first, so there the march stays in order (_stage_march).
        model = self.model
        st.side, st.marched = None, None
        pre = self._prefetched
        if pre is not None and not st.refresh and self._prefetch_matches(pre[0], st.rays_o, st.rays_d, st.noises):
            self._prefetched = None
            st.marched, st.side = pre[1], self._side          # started during the previous call
        else:
            self._drop_prefetch()                       # other rays than announced (or a refresh): marched for nothing
            if self.overlap_march and not st.refresh and model.mean_count > 0:
                st.marched, st.side = self._march_on_side(st.rays_o, st.rays_d, st.noises), self._side

    def _stage_planes(self, st):
        model = self.model
        if self._roi_request is not None:       # see rebuild_planes: never run a step on the window of the previous grid
            self._roi = self._finish_roi()
        if self._pending and (st.refresh or not self._roi_valid):
            self.flush_deferred()
            self._mark("adam_catchup")
        if self.use_roi and not st.refresh and not self._roi_valid:
            self._roi, self._roi_valid = self._compute_roi(), True
        st.tm = self.rebuild_planes(roi=self.use_roi and not st.refresh)
        self._mark("idwt_fwd")
        if st.refresh:
            if self.multi:
                # every rank evaluates 1/world of the candidate cells; the all-gather keeps the replicas' grids (hence
                # bitfield, occupancy window and collective sizes) bit-identical
                model.update_extra_state(shard=(self.rank, self.world, lambda t: D.all_gather_slices(t, self.pg)))
                # one sample budget for all ranks: the mean of the ranks' mean counts
                mc = torch.tensor([float(model.mean_count)], dtype=torch.float64, device=self.dev)
                dist.all_reduce(mc, group=self.pg)
                model.mean_count = int(mc.item() / self.world)
            else:
                model.update_extra_state()
            if self.post_refresh is not None:
                self.post_refresh()
            self._occ_box = None            # the bitfield changed: the march's far clip is rebuilt on first use
            if self.use_roi:
                self._request_roi()         # read where the window is first needed (_stage_backward): no host stall here
                self._roi_valid = True
            self._mark("grid_refresh")

    def _stage_march(self, st):
        model, R = self.model, self.R
        st.packed = F_.pack_weights(*self.Ws, self.C, self.H)
        st.sort_beside = False
        if st.side is None:
            if self.overlap_march and self.binned and R % 32 == 0 and model.mean_count > 0:
                if self._side is None:
                    self._side = torch.cuda.Stream()
                # samples in order (refresh steps: the bitfield has just changed), the sort passes beside the field forward
                st.marched = self._march(st.rays_o, st.rays_d, st.noises, sort_stream=self._side, clip=self.clip_far_in_order)
                st.sort_beside = True
            else:
                st.marched = self._march(st.rays_o, st.rays_d, st.noises, clip=self.clip_far_in_order)
        marched, (ev_march, st.ev_sort) = st.marched
        if st.side is not None:
            torch.cuda.current_stream().wait_event(ev_march)
        st.counter, st.xyzs, st.dirs, st.deltas, st.rays, st.sort_ws = marched
        st.M = st.xyzs.shape[0]
        self._mark("march")

    def _stage_render(self, st):
        model, lib = self.model, L.lib()
        C, R, H, N, M = self.C, self.R, self.H, st.N, st.M
        # rows past counter[0] are the zero padding of the sample budget: skipped on the device
        sigma, st.rgb, st.feats = F_.field_forward(st.tm, st.xyzs, st.dirs, st.packed, float(model.bound), C, R, H,
                                                   save_feats=True, m_actual=st.counter)
        st.sigma_field = sigma   # exp(logit) as the field produced it: the hidden-128 backward reads it
        if model.density_scale != 1:
            sigma = sigma * model.density_scale
        st.sigma = sigma
        self._mark("field_fwd")
        st.ws = torch.empty(N, dtype=torch.float32, device=self.dev)
        st.depth = torch.empty(N, dtype=torch.float32, device=self.dev)
        st.image = torch.empty(N, 3, dtype=torch.float32, device=self.dev)
        L.check(lib.tnl_composite_rays_train_forward(L.ptr(sigma), L.ptr(st.rgb), L.ptr(st.deltas), L.ptr(st.rays), L.u32(M),
                                                     L.u32(N), L.f32(self.T_thresh), L.ptr(st.ws), L.ptr(st.depth),
                                                     L.ptr(st.image), L.stream()), "composite_rays_train_forward")
        # image + (1 - ws) * bg (renderer.py:317), MSE mean over rays and channels (utils.py:595) and d(scaled loss):
        # one launch (csrc/loss.hip)
        bg = self.bg if st.bg_color is None else st.bg_color
        st.pred = torch.empty(N, 3, dtype=torch.float32, device=self.dev)
        st.g_pred = torch.empty(N, 3, dtype=torch.float32, device=self.dev)
        st.g_ws = torch.empty(N, dtype=torch.float32, device=self.dev)
        # one launch: MSE / L1 / non-finite accumulators and the MLP gradient zeroed, 1 / loss scale (csrc/stepstate.hip)
        st.mse_local = torch.empty((), dtype=torch.float32, device=self.dev)
        L.check(lib.tnl_step_prologue(L.ptr(self.scale), L.ptr(self.inv_scale), L.ptr(self.abs_sum),
                                      L.ptr(self.nonfinite), L.ptr(st.mse_local), L.ptr(self.mlp.grad),
                                      L.u32(self.mlp.grad.numel()), L.stream()), "step_prologue")
        bg_rays = bg.to(torch.float32).contiguous() if torch.is_tensor(bg) else None
        L.check(lib.tnl_mse_loss(L.ptr(st.image), L.ptr(st.ws), L.ptr(st.gt_rgb.contiguous()),
                                 L.f32(0.0 if bg_rays is not None else bg), L.ptr(bg_rays), L.u32(N),
                                 L.f32(1.0 / (3.0 * st.n_glob)), L.ptr(self.scale), L.ptr(st.pred), L.ptr(st.g_pred),
                                 L.ptr(st.g_ws), L.ptr(st.mse_local), L.stream()), "mse_loss")
        self._mark("composite_fwd_loss")

    def _stage_backward(self, st):
        model, lib = self.model, L.lib()
        C, R, H, N, M = self.C, self.R, self.H, st.N, st.M
        # rows behind the sample count are never read (m_actual) and the composite backward zeroes the in-buffer tail
        # of a ray the budget dropped: no zero fill of the two gradient buffers (raymarching.py:283-284)
        g_sigma = torch.empty(M, dtype=torch.float32, device=self.dev)
        g_rgb = torch.empty(M, 3, dtype=torch.float32, device=self.dev)
        L.check(lib.tnl_composite_rays_train_backward(L.ptr(st.g_ws), L.ptr(st.g_pred), L.ptr(st.sigma), L.ptr(st.rgb),
                                                      L.ptr(st.deltas), L.ptr(st.rays), L.ptr(st.ws), L.ptr(st.image), L.u32(M),
                                                      L.u32(N), L.f32(self.T_thresh), L.ptr(g_sigma), L.ptr(g_rgb),
                                                      L.stream()), "composite_rays_train_backward")
        if model.density_scale != 1:
            g_sigma = g_sigma * model.density_scale
        self._mark("composite_bwd")
        if self._roi_request is not None:           # a refresh step: the new window, requested right after the grid update
            self._roi = self._finish_roi()
        st.roi = self._roi if (self.use_roi and self.binned and R % 32 == 0) else None
        if self.binned and R % 32 == 0:
            # no global float atomics: dF -> fp16 -> tile-sorted matrix-core accumulation (csrc/scatter.hip), written
            # straight in the (3,C,R,R) layout the adjoint IDWT reads
            if st.roi is None:
                st.g_cm = torch.empty(3, C, R, R, dtype=torch.float32, device=self.dev)
            else:
                st.g_cm = torch.empty(3 * C, st.roi[7], st.roi[6], dtype=torch.float32, device=self.dev)
            dfeat = torch.empty(3, M, C, dtype=torch.float16, device=self.dev)   # plane-major, see field_bwd.hip
            F_.field_backward(g_sigma, g_rgb, st.sigma_field, None, st.feats, st.xyzs, st.dirs, st.packed, float(model.bound),
                              C, R, H, st.g_cm, self.mlp.grad, m_actual=st.counter, dfeat=dfeat)
            self._mark("field_bwd")
            # The following batch's march + tile sort (ALU/latency-bound, few waves) start here, underneath the
            # HBM-bound tail of the step (tile reduction, adjoint IDWT, Adam; in the multi-GPU modes the collectives):
            # the two MFMA field kernels own their SIMDs' whole register files, so side work beside them is time-sliced
            # in at their cost (docs/EXPERIMENTS.md: the start positions measured in rounds 2 and 3).
            if self.prefetch_at == "bwd":
                self._prefetch_next(st.next_rays)
            if st.side is not None or st.sort_beside:
                torch.cuda.current_stream().wait_event(st.ev_sort)
            if self.deterministic:
                F_.order_tile_lists(st.sort_ws, R, st.xyzs.shape[0])
            st.scattered = False
            bands = self._exchange_bands(st.roi)
            if bands is None:
                F_.plane_grad_reduce(st.sort_ws, dfeat, st.xyzs, float(model.bound), C, R, st.g_cm, channel_major=True,
                                     nonfinite_flag=self.nonfinite, roi=self._roi10() if st.roi is not None else None)
            else:
                # band b: tile reduction of its rows on the launch stream, then its reduce-scatter on the communication
                # stream (RCCL: behind an event) while band b + 1 is being reduced
                if self._comm is None:
                    self._comm = torch.cuda.Stream()
                main = torch.cuda.current_stream()
                parts, waits = [], []
                for y0, hb in bands:
                    buf = torch.empty(3 * C, hb, st.roi[6], dtype=torch.float32, device=self.dev)
                    sub = list(st.roi)
                    sub[3:6] = [oy + y0 for oy in st.roi[3:6]]
                    sub[7] = hb
                    F_.plane_grad_reduce(st.sort_ws, dfeat, st.xyzs, float(model.bound), C, R, buf, channel_major=True,
                                         nonfinite_flag=self.nonfinite, roi=sub + [C, 0])
                    self._comm.wait_stream(main)
                    with torch.cuda.stream(self._comm):
                        part, wait = D.reduce_scatter_slices_async(buf, self.pg if self.multi else None)
                    buf.record_stream(self._comm)
                    parts.append(part)
                    waits.append(wait)
                for w in waits:
                    w()
                main.wait_stream(self._comm)
                s0, s1 = self._slice_range() if self.multi else (0, 3 * C)
                st.g_cm = torch.cat([p_[: s1 - s0] if not self.multi else p_ for p_ in parts], dim=1)   # [S/G, rh, rw]
                st.scattered = self.multi
            if self.prefetch_at == "reduce":
                self._prefetch_next(st.next_rays)
            self._mark("plane_grad_binned")
            st.grad_tm = None
        else:
            st.g_cm = None
            st.grad_tm = torch.zeros(3, R, R, C, dtype=torch.float32, device=self.dev)
            F_.field_backward(g_sigma, g_rgb, st.sigma_field, None, st.feats, st.xyzs, st.dirs, st.packed, float(model.bound),
                              C, R, H, st.grad_tm, self.mlp.grad, m_actual=st.counter)
            self._mark("field_bwd")
            self._prefetch_next(st.next_rays)

    def _stage_optimise(self, st):
        lib = L.lib()
        lr_t = self.lr * lr_factor(self.global_step, self.iters, self.warmup)
        l1 = self.lam / (self.J * self.coef_numel) if (self.J > 0 and self.lam > 0) else 0.0
        inv_scale = self.inv_scale
        if self.multi:
            dist.all_reduce(self.mlp.grad, group=self.pg)
        if st.g_cm is not None:
            # GradScaler probe BEFORE the dense backward, so that the optimiser can be fused into it: the plane
            # gradient reports non-finite values through the tile kernel's flag, the MLP gradient is 13.5k floats
            found_inf = self._scaler_probe(self.mlp.grad, None, self.nonfinite)
            self._mark("scaler_probe")
            if self.fuse_adam:
                s0, s1 = self._adjoint(None, st.g_cm, fuse=(lr_t, l1, found_inf, inv_scale))
                if self.prefetch_at == "adjoint":
                    self._prefetch_next(st.next_rays)
                self._mark("idwt_adjoint_adam")
            else:
                scattered = getattr(st, "scattered", False)
                begun = None
                self._fused_levels = ()
                if (self.fuse_live and self.defer_adam and st.roi is not None and self._rect_ok and self._rects_roi is self._roi
                        and all(r is not None for r in self._rects)):
                    # the window's rectangles are known from the adjoint of an earlier step under it: the step is recorded
                    # (and the live pieces fixed) before the adjoint, whose column-walk levels then carry the optimiser
                    sl = self._slice_range() if (scattered or (self.multi and self.dist_mode == "sharded")) else (0, 3 * self.C)
                    begun = self._adam_live_begin(lr_t, l1, found_inf, sl[0], sl[1], self._rects)
                    wmin = int(lib.tnl_idwt_get_walk_min_n())
                    self._fused_levels = tuple(
                        lvl for lvl in range(self.J)
                        if self._live[lvl] is not None and self.coef.params[lvl].shape[-1] >= wmin
                        and self.coef.params[lvl].shape[-1] % 8 == 0)[-self.fuse_live_levels:]
                    self.last_fused_levels = self._fused_levels      # (for reports: kept over refresh steps)
                s0, s1 = self._adjoint(None, st.g_cm, roi=st.roi, scattered=scattered,
                                       live_adam=None if begun is None else (begun, l1, found_inf, inv_scale))
                if self.prefetch_at == "adjoint":
                    self._prefetch_next(st.next_rays)
                self._mark("idwt_adjoint")
                rects = self._rects if (st.roi is not None and self._rect_ok) else None
                if self.defer_adam and rects is not None:
                    self._adam_levels_live(lr_t, l1, found_inf, inv_scale, s0, s1, rects, begun=begun)
                else:
                    self._adam_levels(lr_t, l1, found_inf, inv_scale, s0, s1, rects)
                self._mark("adam_coef")
                if self._pending == 16:     # the ring is full (a whole density-grid period at the default interval)
                    self.flush_deferred()
                    self._mark("adam_catchup")
        else:
            s0, s1 = self._adjoint(st.grad_tm, None)
            self._mark("idwt_adjoint")
            # GradScaler: skip the step when any gradient is non-finite.  A non-finite plane gradient always
            # reaches the coarse LL gradient through the low-pass adjoint, so checking LL + MLP grads suffices.
            found_inf = self._scaler_probe(self.mlp.grad, self.ll.grad, None)
            self._mark("scaler_probe")
            if self.dist_mode == "sharded":
                self._adam_sharded(lr_t, l1, found_inf, inv_scale, s0, s1)
            else:
                self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum)
                self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale)
            self._mark("adam_coef")
        if self._capturing:
            mlp = self.mlp
            L.check(lib.tnl_adam_l1_step_rec(L.ptr(mlp.data), L.ptr(mlp.grad), L.ptr(mlp.m), L.ptr(mlp.v), L.u64(mlp.total),
                                             L.ptr(self._ring[4 * self._last_slot:]), L.f32(self.b1), L.f32(self.b2),
                                             L.f32(self.eps), L.ptr(inv_scale), L.f32(0.0), L.ptr(found_inf), L.ptr(None),
                                             L.stream()), "adam_l1_step_rec")
        else:
            self._adam(self.mlp, lr_t, 0.0, found_inf, inv_scale)
        # optimiser-step count, GradScaler.update(), L1 value: one launch
        reg = torch.empty((), dtype=torch.float32, device=self.dev)
        L.check(lib.tnl_step_epilogue(L.ptr(found_inf), L.ptr(self.opt_steps), L.ptr(self.scale),
                                      L.ptr(self.growth_tracker), L.f32(2.0), L.f32(0.5), L.i32(self.growth_interval),
                                      L.i32(int(self.fp16)), L.ptr(self.abs_sum if l1 > 0 else None), L.f32(l1),
                                      L.ptr(reg), L.stream()), "step_epilogue")
        self.global_step += 1
        self._stale_params = self._stale_moments = True    # "sharded" mode: see sync_sharded_parameters
        if self.multi:
            mse = st.mse_local.clone()
            dist.all_reduce(mse, group=self.pg)
            if self.dist_mode == "sharded":
                dist.all_reduce(reg, group=self.pg)
        else:
            mse = st.mse_local
        st.loss = mse + reg
        self._mark("tail")
        self.last = {'mse': mse, 'wavelet_reg': reg, 'M': st.M, 'found_inf': found_inf, 'image': st.pred, 'ws': st.ws,
                     'depth': st.depth, 'counter': st.counter, 'lr': lr_t}

    # ------------------------------------------------------------------------------------------
    # captured steps (graph=True)
    def _graph_position(self):
        return self.global_step % self.update_extra_interval if self.update_extra_interval > 0 else -1

    def _graph_eligible(self, st):
        """A step whose every launch argument is fixed by (period position, occupancy window and its pieces, sample
        budget, batch size): a steady-state step of the windowed, deferred path with the prefetched march of this batch
        at hand and the following batch announced."""
        j = self._graph_position()
        model = self.model
        if (self.multi or not (1 <= j <= min(self.update_extra_interval, 16) - 2) or st.refresh or self.section_events is not None
                or not (self.binned and self.use_roi and self.defer_adam and self._rect_ok and self.overlap_march)
                or self.fuse_adam or self.overlap_exchange > 1 or self._roi is None or not self._roi_valid
                or self._roi_request is not None or self._rects_roi is not self._roi or self._live is None
                or self._pending != j or self._pending >= 15 or st.noises is None or st.bg_color is not None
                or torch.is_tensor(self.bg) or st.next_rays is None or len(st.next_rays) < 3 or st.next_rays[2] is None
                or model.mean_count <= 0 or self._prefetched is None or self.R % 32 != 0):
            return False
        pre = self._prefetched
        if not self._prefetch_matches(pre[0], st.rays_o, st.rays_d, st.noises):
            return False
        g = self._graphs.get(j)
        if g is not None and (g.pending != self._pending or g.local_step_mod != model.local_step % 16):
            return False
        return all(t_.dtype == torch.float32 and t_.is_contiguous() for t_ in
                   (st.rays_o, st.rays_d, st.gt_rgb, st.noises, *st.next_rays[:3]))

    def _graph_signature(self, st):
        roi = tuple(self._roi)
        ext = None if self._row_ext is None else hash(self._row_ext.tobytes())
        tup = lambda rs: tuple(None if r is None else tuple(int(x) for x in r) for r in rs)
        return (roi, ext, tup(self._rects), tup(self._live), int(self.model.mean_count), st.N, st.n_glob, float(self.bg),
                bool(self.deterministic), bool(self.live_bands), self.update_extra_interval, self.max_steps, self.dt_gamma,
                bool(self.fuse_live))

    def drop_graphs(self):
        self._graphs = {}
        self._graph_key = None

    def _graph_step(self, st):
        """One captured step: inputs copied into the static buffers the launches read, the learning rate into its device
        word, then one graph launch (captured on first use).  Host-side state moves as an eager step moves it."""
        model, dev = self.model, self.dev
        j = self._graph_position()
        sig = self._graph_signature(st)
        if sig != self._graph_key:
            self._graphs, self._graph_key = {}, sig
        N = st.N
        if self._graph_in is None or self._graph_in["o"].shape[0] != N:
            mk = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            self._graph_in = {"o": mk(N, 3), "d": mk(N, 3), "gt": mk(N, 3), "nz": mk(N), "o2": mk(N, 3), "d2": mk(N, 3),
                              "nz2": mk(N)}
            self._graphs = {}
        if self._lr_dev is None:
            self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
            self._cap_stream = torch.cuda.Stream()
            self._graph_done = torch.cuda.Event()
            self._graph_done.record()
        gi, nxt = self._graph_in, st.next_rays
        if nxt[0].shape[0] != N:
            return self._eager_step(st)
        lr_t = self.lr * lr_factor(self.global_step, self.iters, self.warmup)
        pre = self._prefetched
        g = self._graphs.get(j)
        main = torch.cuda.current_stream()
        from_graph = pre[1][1][0] is self._graph_done
        if self._side is not None and not from_graph:
            main.wait_stream(self._side)           # the prefetch of an EAGER step may still be running (a captured step
            #                                        joined its side work before it ended)
        torch._foreach_copy_([gi["o"], gi["d"], gi["gt"], gi["nz"], gi["o2"], gi["d2"], gi["nz2"]],
                             [st.rays_o, st.rays_d, st.gt_rgb, st.noises, nxt[0], nxt[1], nxt[2]])
        self._lr_dev.fill_(lr_t)
        if g is None:
            # device tables the stage methods build lazily on a period's first windowed step (host-to-device copies are
            # not allowed inside a capture): now
            self._forward_spans()
            self._adjoint_spans()
            g = self._capture_step(st, j, pre)
            self._graphs[j] = g
            self.graph_captures += 1
        else:
            # the march this step consumes: where the captured launches expect it
            src = [t_ for t_ in pre[1][0] if torch.is_tensor(t_)]
            dst = [t_ for t_ in g.in_marched if torch.is_tensor(t_)]
            if any(a is not b for a, b in zip(src, dst)):
                torch._foreach_copy_(dst, src)
            # host-side state, as the stage methods leave it
            self._prefetched = None
            model.local_step += 1                  # the ring slot the prefetched march (of the NEXT batch) takes
            self._pending += 1
            self.deferred_steps += 1
            self.global_step += 1
            self._stale_params = self._stale_moments = True
        g.graph.replay()
        self.graph_replays += 1
        # the graph joined its side work before it ended: whoever consumes the prefetch -- the next captured step (which
        # does not look at events) or an eager one -- is ordered behind it by the launch stream alone; the events the
        # capture recorded are not real ones, an eager consumer gets one recorded here
        ev = self._graph_done                      # (one event recorded once, long complete: waiting for it is free)
        self._prefetched = (self._prefetch_key(nxt), (g.out_prefetch[0], (ev, ev)), g.slot_step_of(model))
        self.last = dict(g.last)
        self.last["lr"] = lr_t
        return g.loss

    def _eager_step(self, st):
        self._mark("begin")
        self._stage_pickup(st)
        self._stage_planes(st)
        self._stage_march(st)
        self._stage_render(st)
        self._stage_backward(st)
        self._stage_optimise(st)
        return st.loss

    def _capture_step(self, st, j, pre):
        """Runs the stage methods of an eager step under stream capture (nothing executes: the caller replays the graph
        once); the host-side state changes they make are the step's."""
        model = self.model
        gi = self._graph_in
        g = types.SimpleNamespace()
        g.graph = torch.cuda.CUDAGraph()
        g.in_marched = pre[1][0]
        g.pending = self._pending
        g.local_step_mod = model.local_step % 16
        g.slot_step_of = lambda m: m.local_step - 1
        # what the captured launches point at and this object would otherwise let go at the next refresh
        g.keep = (self._band_cache, self._live_bands, self._live, self._rects, gi, self._tm_full, pre)
        st.rays_o, st.rays_d, st.gt_rgb, st.noises = gi["o"], gi["d"], gi["gt"], gi["nz"]
        st.next_rays = (gi["o2"], gi["d2"], gi["nz2"])
        main = torch.cuda.current_stream()
        self._cap_stream.wait_stream(main)
        self._capturing = True
        try:
            with torch.cuda.graph(g.graph, pool=self._graph_pool, stream=self._cap_stream):
                # the prefetched march is complete (the launch stream waited for the side stream); its events belong to
                # uncaptured work and cannot be waited for in here: stand-ins recorded inside the capture
                e = torch.cuda.Event()
                e.record()
                st.marched, st.side = (pre[1][0], (e, e)), self._side
                self._prefetched = None
                self._stage_planes(st)
                self._stage_march(st)
                self._stage_render(st)
                self._stage_backward(st)
                self._stage_optimise(st)
                torch.cuda.current_stream().wait_stream(self._side)     # a graph cannot leave a forked stream open
        finally:
            self._capturing = False
        main.wait_stream(self._cap_stream)
        g.loss, g.last = st.loss, dict(self.last)
        g.out_prefetch = self._prefetched[1]
        return g

    def _exchange_bands(self, roi):
        """[(first row, rows)] of the bands the plane-gradient window is exchanged in, or None (one piece): overlap_exchange
        K > 1, an occupancy window, the slice-sharded mode (or a single process, where only the banded reduction's own
        cost shows: the measurement of DESIGN.md section 5), not the Adam-fused adjoint."""
        K = self.overlap_exchange
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

    def _prefetch_next(self, next_rays):
        """Starts the following batch's march + tile sort on the side stream (see step(next_rays=...))."""
        model = self.model
        next_refresh = self.update_extra_interval > 0 and (self.global_step + 1) % self.update_extra_interval == 0
        if next_rays is None or not self.overlap_march or next_refresh or model.mean_count <= 0:
            return
        no, nd = next_rays[0], next_rays[1]
        nn = next_rays[2] if len(next_rays) > 2 else None
        key = self._prefetch_key(next_rays)
        # the ring slot the march takes (run_cuda's local_step rule), so that a dropped prefetch gives back exactly it
        slot_step = model.local_step
        self._prefetched = (key, self._march_on_side(no, nd, nn), slot_step)

    @staticmethod
    def _prefetch_key(next_rays):
        # the announced tensors are kept (their storage cannot be recycled for another batch meanwhile) together with
        # their version counters (an in-place refill of a persistent ray buffer is noticed)
        no, nd = next_rays[0], next_rays[1]
        nn = next_rays[2] if len(next_rays) > 2 else None
        return tuple((t_, t_.data_ptr(), tuple(t_.shape), t_._version) if t_ is not None else None for t_ in (no, nd, nn))

    @staticmethod
    def _prefetch_matches(key, rays_o, rays_d, noises):
        for k, t_ in zip(key, (rays_o, rays_d, noises)):
            if (k is None) != (t_ is None):
                return False
            if k is not None and (k[1] != t_.data_ptr() or k[2] != tuple(t_.shape) or k[3] != t_._version
                                  or k[0]._version != k[3] or k[0].dtype != t_.dtype):
                return False
        return True

    def _drop_prefetch(self):
        """Forget a march started for a batch that is not coming: its step_counter slot and local_step are given back
        (mean_count at the next refresh averages the slots), and the launch stream is ordered behind it."""
        pre, self._prefetched = self._prefetched, None
        if pre is None:
            return
        (_, (_, ev_sort)) = pre[1]
        torch.cuda.current_stream().wait_event(ev_sort)
        # only if nothing moved the ring meanwhile (a manual update_extra_state() resets local_step to 0: the slot then
        # belongs to a finished period and mean_count has already been taken)
        if self.model.local_step == pre[2] + 1:
            self.model.local_step -= 1
            self.model.step_counter[self.model.local_step % 16].zero_()

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

    # ------------------------------------------------------------------------------------------
    # live / deferred split of the coefficient pass (defer_adam)
    def _live_rects(self, rects):
        """Per level the rectangle (per plane origin, common size, columns in multiples of 32, rows of 8; the level's own
        coordinates) holding
        everything the windowed rebuild reads -- the level's output window halved and grown by 6 (the longest filter,
        bior6.8, reaches 4-5 coefficients to either side) -- and everything the windowed adjoint writes (rects).
        None: the whole level stays live.  tests/test_adam_deferred_gpu.py poisons everything outside with NaN."""
        return occupancy.live_rects(self._forward_windows(), rects, [p.shape[-1] for p in self.coef.params],
                                    getattr(self, "live_col_align", 32))

    def _level_needs(self):
        """Per level [3, n/8, 2] int arrays: for plane p and rows 8b .. 8b+7 of the level's n x n grid the column piece
        [lo, end) that can reach a sampled texel (and that a gradient can reach) -- the planes' own piece
        (tnl_occupancy_row_extents) halved and grown by 6 per level, exactly like the rectangles of _live_rects; rows
        8b .. 8b+7 are within 6 of the halves of rows 16b-12 .. 16b+27 of the next finer grid = its row groups
        2b-2 .. 2b+3.  Empty pieces are (big, -1).  None when the geometry is not the plain dyadic one."""
        if "needs" in self._band_cache:
            return self._band_cache["needs"]
        needs = None
        if self._row_ext is not None and not self.base_res:
            big = np.int64(0x7fffffff)
            cur, nf = self._row_ext, self.R
            needs = [None] * self.J
            for lvl in reversed(range(self.J)):
                n = self.coef.params[lvl].shape[-1]
                if 2 * n != nf or n % 8 != 0:
                    needs = None
                    break
                G, Gf = n // 8, cur.shape[1]
                lo = np.full((3, G), big)
                hi = np.full((3, G), -1, dtype=np.int64)
                for b in range(G):
                    seg = cur[:, max(2 * b - 2, 0):min(2 * b + 3, Gf - 1) + 1]
                    lo[:, b], hi[:, b] = seg[:, :, 0].min(1), seg[:, :, 1].max(1)
                has = hi > lo
                lo = np.where(has, np.maximum(lo // 2 - 6, 0), big)
                hi = np.where(has, np.minimum((hi + 1) // 2 + 6, n), -1)
                needs[lvl] = cur = np.stack([lo, hi], axis=-1)
                nf = n
        self._band_cache["needs"] = needs
        return needs

    def _band_tables(self, live):
        """Per level None or (device int32 band table, float4s per slice, host table, device span table of the same
        pieces for the adjoint) for tnl_adam_l1_step_live_bands:
        the live rectangle's 8-row bands cut down to the columns _level_needs allows (aligned outward to the column
        granule, one width per band over the three planes).  None where that saves less than 8 % of the rectangle."""
        key = tuple(None if lv is None else tuple(lv) for lv in live)
        if key in self._band_cache:
            return self._band_cache[key]
        needs = self._level_needs()
        out = [None] * self.J
        host = []
        al = getattr(self, "live_col_align", 32)
        for lvl, lv in enumerate(live):
            if lv is None or needs is None or lv[7] % 8 != 0 or lv[7] // 8 > 128 or any(o % 8 for o in lv[3:6]):
                continue
            rw, rh = lv[6], lv[7]
            nb = rh // 8
            E = needs[lvl]
            w = np.zeros(nb, dtype=np.int64)
            x0 = np.zeros((3, nb), dtype=np.int64)
            los, his = [], []
            for p in range(3):
                e = E[p, lv[3 + p] // 8: lv[3 + p] // 8 + nb]
                lo = np.clip(e[:, 0], lv[p], lv[p] + rw) // al * al
                hi = np.clip((np.clip(e[:, 1], lv[p], lv[p] + rw) + al - 1) // al * al, lv[p], lv[p] + rw)
                empty = e[:, 1] <= e[:, 0]
                lo = np.where(empty, lv[p], lo)
                hi = np.where(empty, lv[p], np.maximum(hi, lo))
                los.append(lo)
                his.append(hi)
                w = np.maximum(w, hi - lo)
            for p in range(3):
                x0[p] = np.clip(np.minimum(los[p], lv[p] + rw - w), lv[p], None)
            quads = int(2 * w.sum())
            if quads == 0 or quads > 0.92 * rh * (rw // 4) or (al % 4) != 0:
                continue
            pref = np.concatenate([[0], np.cumsum(2 * w)])
            tbl = np.concatenate([pref, w // 4, x0.reshape(-1)]).astype(np.int32)
            # the same pieces as a span table of the level's grid (tnl_idwt_level_backward_spans)
            n = self.coef.params[lvl].shape[-1]
            sp = np.empty((3, n // 8, 2), dtype=np.int32)
            sp[..., 0], sp[..., 1] = 0x7fffffff, -1
            for p in range(3):
                g0 = lv[3 + p] // 8
                sp[p, g0:g0 + nb, 0] = np.where(w > 0, x0[p], 0x7fffffff)
                sp[p, g0:g0 + nb, 1] = np.where(w > 0, x0[p] + w, -1)
            host.append((lvl, np.concatenate([tbl, np.zeros(-tbl.size % 4, np.int32), sp.reshape(-1)]), quads, tbl.size,
                         (tbl.size + 3) // 4 * 4))
        if host:
            # one upload for all levels (each table 16-byte aligned inside it)
            offs, tot = [], 0
            for h in host:
                offs.append(tot)
                tot += (h[1].size + 3) // 4 * 4
            flat = np.zeros(tot, dtype=np.int32)
            for o, h in zip(offs, host):
                flat[o:o + h[1].size] = h[1]
            dev = torch.from_numpy(flat).to(self.dev)
            for o, (lvl, both, quads, nt, so) in zip(offs, host):
                out[lvl] = (dev[o:o + nt], quads, both[:nt], dev[o + so:o + both.size])
        self._band_cache[key] = out
        return out

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
        keep = [lvl for lvl in range(self.J) if begun is None or lvl not in self._fused_levels]
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
        if self._capturing:      # the step's scalars from the ring slot just written (the same bits)
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
        for lvl in range(self.J):
            n = self.coef.params[lvl].shape[-1]
            per = 3 * n * n
            base = self.coef.offsets[lvl]
            self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, base + s0 * per, base + s1 * per)
        n0 = self.ll.params[0].shape[-1]
        self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale, None, s0 * n0 * n0, s1 * n0 * n0)

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
