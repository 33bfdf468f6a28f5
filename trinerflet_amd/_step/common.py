"""What every part of TrainStep shares: imports, the learning-rate schedule, the flat parameter buffers."""
import ctypes as C_
import math
import types

import numpy as np

import torch
import torch.distributed as dist

from .. import _lib as L
from .. import distributed as D
from .. import occupancy
from .. import raymarching
from ..nerf import field as F_
from ..triplaneencoder.triplane_encoder import (_IDWTLevel, _ToTexelMajor, half_roi_into_texel_major, half_to_texel_major,
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


