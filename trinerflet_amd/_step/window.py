"""TrainStep: the occupancy window of a density-grid period: its read-back, the per-level windows, span tables, live rectangles and band pieces."""
from .common import (C_, D, F_, L, _Flat, _IDWTLevel, _StepState, _ToTexelMajor, dist, half_roi_into_texel_major,  # noqa: F401
                     half_to_texel_major, idwt_level_half, idwt_level_half_roi, lr_factor, math, np, occupancy, raymarching,
                     torch, types)


class WindowMixin:
    """Methods of TrainStep (trinerflet_amd/train.py): the occupancy window of a density-grid period: its read-back,
    the per-level windows, span tables, live rectangles and band pieces."""

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
