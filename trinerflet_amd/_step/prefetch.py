"""TrainStep: the march + tile sort of a batch, in order or started for the following batch on the side stream."""
from .common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                     half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                     torch, types)


class PrefetchMixin:
    """Methods of TrainStep (trinerflet_amd/train.py): the march + tile sort of a batch, in order or started for the
    following batch on the side stream."""

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
