"""TrainStep: steady-state steps of a density-grid period captured and replayed as HIP graphs (graph=True)."""
from .common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                     half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                     torch, types)


class GraphMixin:
    """Methods of TrainStep (trinerflet_amd/train.py): steady-state steps of a density-grid period captured and
    replayed as HIP graphs (graph=True)."""

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
                or self.fuse_adam or self.overlap_exchange == "auto" or self.overlap_exchange > 1 or self._roi is None or not self._roi_valid
                or self._roi_request is not None or self._rects_roi is not self._roi or self._live is None
                or self._pending != j or self._pending >= 15 or st.noises is None or st.bg_color is not None
                or torch.is_tensor(self.bg) or st.next_rays is None or len(st.next_rays) < 3 or st.next_rays[2] is None
                or model.mean_count <= 0 or self._prefetched is None or self.R % 32 != 0 or self.min_res_learn > 0):
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
            # a signature under which no captured step was ever replayed a second time cost more than it saved: on a real
            # trajectory every grid refresh moves the sample budget (part of the signature), so every step is captured AND
            # replayed -- never faster than the eager step.  Three such signatures in a row: graphs
            # are switched off for this object instead of being re-captured for ever (they pay with a pinned budget).
            if self._graph_key is not None and self._key_captures > 0:
                self._wasted_keys = self._wasted_keys + 1 if self._key_replays <= self._key_captures else 0
            self._key_captures = self._key_replays = 0
            self._graphs, self._graph_key = {}, sig
            if self._wasted_keys >= 3:
                import warnings
                warnings.warn("TrainStep(graph=True): the captured steps were re-captured at every density-grid refresh "
                              f"({self.graph_captures} captures, {self.graph_replays} replays): graphs switched off")
                self.graph = False
                self.graph_auto_disabled = True
                self.drop_graphs()
                return self._eager_step(st)
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
            self._key_captures += 1
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
        self._key_replays += 1
        # the graph joined its side work before it ended: whoever consumes the prefetch -- the next captured step (which
        # does not look at events) or an eager one -- is ordered behind it by the launch stream alone; the events the
        # capture recorded are not real ones, an eager consumer gets one recorded here
        ev = self._graph_done                      # (one event recorded once, long complete: waiting for it is free)
        self._prefetched = (self._prefetch_key(nxt), (g.out_prefetch[0], (ev, ev)), g.slot_step_of(model))
        self.last = dict(g.last)
        self.last["lr"] = lr_t
        return g.loss

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
