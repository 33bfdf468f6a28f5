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

Layout: this file holds the constructor, step() and the six stages of a step; the rest of the class lives in mixins under
_step/ -- window.py (occupancy window, span tables, live rectangles and band pieces), planes.py (rebuild, adjoint, banded
exchange), optimiser.py (Adam passes, the live / deferred split and its replay), prefetch.py (march + tile sort, in order
or for the following batch on the side stream), graph.py (captured steps), common.py (imports, schedule, flat buffers).

Multi-GPU: rays are sharded across ranks, planes and MLP weights replicated.
  mode "allreduce": plane gradients all-reduced (RCCL) before the adjoint; every rank repeats the dense work.
  mode "sharded"  : the 3*C (plane, channel) slices are the shard unit (the IDWT is depthwise): plane gradients
                    are reduce-scattered by slice, each rank runs adjoint + Adam + IDWT on 3C/G slices, and the
                    rebuilt planes are all-gathered -- same bytes on the wire, dense HBM work divided by G.
"""
from ._step.common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                           half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                           torch, types)
from ._step.graph import GraphMixin
from ._step.optimiser import OptimiserMixin
from ._step.planes import PlanesMixin
from ._step.prefetch import PrefetchMixin
from ._step.window import WindowMixin


class TrainStep(WindowMixin, PlanesMixin, OptimiserMixin, PrefetchMixin, GraphMixin):
    def __init__(self, model, lr=1e-2, wavelet_regularization=0.4, iters=30000, warmup_steps=0,
                 betas=(0.9, 0.99), eps=1e-15, fp16=True, update_extra_interval=16, background_color=0.0,
                 max_steps=1024, dt_gamma=0.0, T_thresh=1e-4, init_scale=65536.0, growth_interval=2000,
                 dist_mode=None, process_group=None, single_rank_collectives=False, binned=True, fuse_adam=False, use_roi=True, tune_placement=None,
                 defer_adam=None, deterministic=False, live_bands=True, overlap_exchange=0, graph=False,
                 min_wavelet_resolution_to_learn=-1, grad_transport="fp32"):
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
        # everywhere).  Results are bit-identical to the whole-plane step.  Both plane precisions (round 6): with
        # plane_dtype = float32 -- the reference's training precision, utils.py:1138-1140 -- the finest level writes its
        # window of a full-size fp32 array and the layout pass converts that window (rebuild_planes).
        self.use_roi = (use_roi and binned and not fuse_adam and self.J > 0 and self.R % 64 == 0 and self.C % 8 == 0
                        and enc.plane_dtype in (torch.float16, torch.float32) and self.base_res == 0)
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
        # min_wavelet_resolution_to_learn (run_utils.py:88; Trainer.clear_grad, utils.py:1105-1114, called between backward
        # and the optimiser step :1168): when > 0 every gradient of the model is dropped except those of the ENCODER
        # parameters whose last dimension exceeds it -- the MLP weights, the LL plane and the coarse wavelet levels (a
        # prefix: sizes grow with the level) then take no optimiser step at all (torch.optim.Adam skips a parameter
        # without a gradient: no moment decay, no step count, no L1 pull), while the loss still carries their L1 value.
        # Here: the adjoint stops above the frozen levels, their Adam passes (and the MLP's) are not launched, their
        # |coef| sum is a cached constant.
        thr = int(min_wavelet_resolution_to_learn or -1)
        self.min_res_learn = thr
        self.frozen_levels = sum(1 for q in self.coef.params if q.shape[-1] <= thr) if thr > 0 else 0
        self.freeze_ll = thr > 0 and enc.planes_features.shape[-1] <= thr
        self.freeze_mlp = thr > 0
        assert self.frozen_levels == 0 or self.freeze_ll      # level 0 has the LL plane's size
        if thr > 0 and fuse_adam:
            raise NotImplementedError("fuse_adam with min_wavelet_resolution_to_learn")
        self._frozen_abs = None
        if thr > 0:
            me_ = __import__("weakref").ref(self)

            def _reloaded(module, incompatible):
                ts_ = me_()
                if ts_ is not None:
                    ts_._frozen_abs = None
            self._ld_hook = model.register_load_state_dict_post_hook(_reloaded)
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
        # ("auto": K from the cost model, distributed.plan_exchange, once the window and the sample budget are known)
        self.overlap_exchange = overlap_exchange if overlap_exchange == "auto" else int(overlap_exchange)
        self._auto_plan = None
        # grad_transport "bf16": the plane-gradient window travels as bfloat16 and is accumulated in fp32 on the slice's
        # owner (distributed._reduce_scatter_bf16: half the reduce-scatter's bytes); "fp32": as computed
        assert grad_transport in ("fp32", "bf16")
        self.grad_transport = grad_transport
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
        self._key_captures = self._key_replays = self._wasted_keys = 0     # see _graph_step: graphs that never pay are dropped
        self.graph_auto_disabled = False
        self._stale_params = self._stale_moments = False
        # where the following batch's march + tile sort start on the side stream (binned mode): behind the field backward
        # ("bwd"), behind the tile reduction ("reduce") or behind the adjoint IDWT ("adjoint")
        self.prefetch_at = "bwd"
        self.side_count_form = 1     # raymarching.count_form of the prefetched march (0: the wavefront-per-ray count pass)
        if __import__("os").environ.get("TNL_SIDE_COUNT_FORM") is not None:      # A/B knob (tools/ab_small.sh)
            self.side_count_form = int(__import__("os").environ["TNL_SIDE_COUNT_FORM"])
        if __import__("os").environ.get("TNL_PREFETCH_AT") is not None:
            self.prefetch_at = __import__("os").environ["TNL_PREFETCH_AT"]
        # workgroups of the prefetched march's emit pass and of its tile sort's fill pass (raymarching.side_caps): 2 and 1 per
        # CU.  At full width the two flood the wave slots just as the adjoint's second column-walk level (two 192-register
        # workgroups per CU) is launched: 550 us for 190 alone; capped, the step is 0.14 ms shorter (3.86-3.89 vs 4.01-4.03
        # in alternation; (0, 0) = uncapped)
        # ... where the step's tail has the slack: base -0.091 ms [-0.106, -0.077], large -0.136 [-0.22, -0.05]; at small (50 M
        # coefficients: tail 0.8 ms against a side chain of 1.0) the capped chain is the critical path: +0.097 -- uncapped below
        # 2^27 coefficients
        n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count if torch.cuda.is_available() else 256
        self.side_caps = (2 * n_cu, n_cu) if 3 * self.C * self.R * self.R >= (1 << 27) else (0, 0)
        if __import__("os").environ.get("TNL_SIDE_CAPS"):            # A/B knob: "emit,fill" workgroups (tools/ab_small.sh)
            self.side_caps = tuple(int(v) for v in __import__("os").environ["TNL_SIDE_CAPS"].split(","))
        self.post_refresh = None    # optional callable run right after every density-grid refresh
        self.section_events = None  # set to [] to record HIP events (on the launch stream) around every stage
        # distributed
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist_mode and dist.is_initialized()) else 1
        # dist_mode "auto": what the cost model of DESIGN.md section 5 picks for this world size (distributed.plan_exchange:
        # the slice-sharded step wherever 3 * channels divides by the world size -- it moves no more bytes than the
        # all-reduce and divides the dense work -- with the band count of the exchange from the link rate TNL_XGMI_GBS)
        if dist_mode == "auto":
            dist_mode = "sharded" if (3 * self.C) % max(self.world, 1) == 0 else "allreduce"
            if overlap_exchange == 0:
                self.overlap_exchange = "auto"
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # multi: the distributed code path is the one that runs.  single_rank_collectives=True takes it in a process group
        # of ONE rank as well (every collective issued, on RCCL each a real reduce_scatter_tensor / all_gather_into_tensor /
        # all_reduce over the group): the way to execute the 8-GPU call sequence on a one-GPU box (tests/test_dist_gpu.py)
        self.multi = self.world > 1 or bool(single_rank_collectives and dist_mode and dist.is_initialized())
        if self.multi and self.world == 1:
            D.force_collectives(process_group)       # scoped to this step's group (not the module-wide switch)
            import weakref
            weakref.finalize(self, D.force_collectives, process_group, False)
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
                        part, wait = D.reduce_scatter_slices_async(buf, self.pg if self.multi else None, self.grad_transport)
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
                        and all(r is not None for r in self._rects[self.frozen_levels:])):
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
                self._adam(self.coef, lr_t, l1, found_inf, inv_scale, self.abs_sum, lo=self._learn_from())
                if not self.freeze_ll:
                    self._adam(self.ll, lr_t, 0.0, found_inf, inv_scale)
            self._mark("adam_coef")
        if self._capturing:
            mlp = self.mlp
            L.check(lib.tnl_adam_l1_step_rec(L.ptr(mlp.data), L.ptr(mlp.grad), L.ptr(mlp.m), L.ptr(mlp.v), L.u64(mlp.total),
                                             L.ptr(self._ring[4 * self._last_slot:]), L.f32(self.b1), L.f32(self.b2),
                                             L.f32(self.eps), L.ptr(inv_scale), L.f32(0.0), L.ptr(found_inf), L.ptr(None),
                                             L.stream()), "adam_l1_step_rec")
        elif not self.freeze_mlp:
            self._adam(self.mlp, lr_t, 0.0, found_inf, inv_scale)
        if self.frozen_levels and l1 > 0:
            self.abs_sum.add_(self._frozen_abs_sum())      # the frozen levels' share of the regulariser's VALUE
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

    def _eager_step(self, st):
        self._mark("begin")
        self._stage_pickup(st)
        self._stage_planes(st)
        self._stage_march(st)
        self._stage_render(st)
        self._stage_backward(st)
        self._stage_optimise(st)
        return st.loss
