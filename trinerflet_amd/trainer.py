"""Trainer -- the loop around TrainStep.

Counterpart of the reference's `Trainer` (reconstruction/nerf/utils.py:375-530 ctor, :762-816 train,
:1116-1228 train_one_epoch2, :1229-1388 evaluate_one_epoch / test, :1390-1532 save/load_checkpoint) and of the
stage loop of reconstruction/main_nerf.py:168-205, restricted to what the README configurations run
(`--fp16 --cuda_ray --triplane_wavelet --ckpt latest_model --ema_decay -1 --fast_training`):

  * one epoch = every pixel of the pool once, in a fresh random order, `num_rays` at a time (RayPool: the
    permutation and the ray generation happen on the device);
  * one iteration = TrainStep.step (planes rebuild, grid refresh every `update_extra_interval`, render, loss, fused
    backward, Adam, GradScaler update, LambdaLR(decay_function));
  * checkpoints are the reference's `.pth` dictionaries: keys epoch / global_step / stats / mean_count /
    mean_density / model [/ optimizer / lr_scheduler / scaler when full], the model under the reference's
    state-dict names, the optimiser as a torch.optim.Adam state_dict over `model.get_params(lr)` -- a checkpoint
    written here loads in the reference's Trainer and vice versa;
  * `use_checkpoint="latest_model"` loads the newest checkpoint's model with strict=False: moving to the next
    resolution keeps LL and the existing wavelet levels and starts the new finest level at zero (main_nerf.py's
    stage hand-off);
  * evaluation renders whole images (renderer eval branch), accumulates PSNR per image as PSNRMeter does, on the
    device, images striped over the ranks when torch.distributed is initialised.

Not here (out of scope): tensorboard, EMA, LPIPS/SSIM, video/mesh export, the GUI, error maps,
patch sampling, CLIP loss.
"""
import gc
import glob
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import distributed as D
from .train import TrainStep, lr_factor


class PSNRMeter:
    """utils.py:245-282: per update -10*log10(mean((pred-truth)^2)); measure() = mean over updates.  The running
    sum stays on the device (one read-back in measure())."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.V = None
        self.N = 0

    def update(self, preds, truths):
        mse = ((preds.to(torch.float32) - truths.to(torch.float32)) ** 2).mean()
        psnr = -10.0 * torch.log10(mse)
        self.V = psnr if self.V is None else self.V + psnr
        self.N += 1

    def state(self, device):
        v = torch.zeros((), device=device, dtype=torch.float64) if self.V is None else self.V.to(torch.float64)
        return torch.stack([v, torch.tensor(float(self.N), device=device, dtype=torch.float64)])

    def measure(self):
        return float(self.V) / self.N

    def report(self):
        return f"PSNR = {self.measure():.6f}"


class Trainer:
    def __init__(self, name, model, workspace=None, lr=1e-2, iters=30000, warmup_steps=0, num_rays=4096,
                 wavelet_regularization=0.0, background_color=0.0, train_rand_bg=False, fp16=True,
                 update_extra_interval=16, max_steps=1024, dt_gamma=0.0, T_thresh=1e-4, use_checkpoint="latest",
                 max_keep_ckpt=2, eval_interval=1, fast_training=False, seed=0, dist_mode=None, process_group=None,
                 mute=True, train_step_kwargs=None, infer_min_step=1, min_wavelet_resolution_to_learn=-1):
        self.name, self.model, self.workspace = name, model, workspace
        self.lr, self.iters, self.warmup_steps, self.num_rays = lr, iters, warmup_steps, num_rays
        self.background_color, self.train_rand_bg = background_color, train_rand_bg
        self.fp16 = fp16
        self.max_steps, self.dt_gamma, self.T_thresh = max_steps, dt_gamma, T_thresh
        self.max_keep_ckpt, self.eval_interval, self.fast_training = max_keep_ckpt, eval_interval, fast_training
        self.seed, self.mute = seed, mute
        self.infer_min_step = infer_min_step   # != 1: the alive-ray loop with wider iterations instead of the one-kernel render
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.device = next(model.parameters()).device
        self.epoch = 0
        self.stats = {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}
        self.ckpt_path = None
        if workspace is not None:
            os.makedirs(workspace, exist_ok=True)
            self.ckpt_path = os.path.join(workspace, "checkpoints")
            os.makedirs(self.ckpt_path, exist_ok=True)
        # checkpoint to start from (utils.py:468-488) -- BEFORE TrainStep re-homes the parameters in flat buffers
        self._pending_full = None
        if self.ckpt_path is not None:
            if use_checkpoint == "scratch":
                pass
            elif use_checkpoint == "latest":
                self._pending_full = self.load_checkpoint(model_only=False, _defer_optimizer=True)
            elif use_checkpoint == "latest_model":
                self.load_checkpoint(model_only=True)
            elif use_checkpoint:
                self._pending_full = self.load_checkpoint(use_checkpoint, model_only=False, _defer_optimizer=True)
        kw = dict(train_step_kwargs or {})
        # --min_wavelet_resolution_to_learn (run_utils.py:88; Trainer.clear_grad, utils.py:1105-1114)
        kw.setdefault("min_wavelet_resolution_to_learn", min_wavelet_resolution_to_learn)
        self.ts = TrainStep(model, lr=lr, wavelet_regularization=wavelet_regularization, iters=iters,
                            warmup_steps=warmup_steps, fp16=fp16, update_extra_interval=update_extra_interval,
                            background_color=background_color, max_steps=max_steps, dt_gamma=dt_gamma,
                            T_thresh=T_thresh, dist_mode=dist_mode if self.world > 1 else None,
                            process_group=process_group, **kw)
        if self._pending_full is not None:
            self._restore_training_state(self._pending_full)
            self._pending_full = None

    # ----------------------------------------------------------------------------------------------------------
    @property
    def global_step(self):
        return self.ts.global_step

    def log(self, *a):
        if not self.mute and self.rank == 0:
            print(*a)

    # ----------------------------------------------------------------------------------------------------------
    # training (utils.py:762-816, :1116-1228)
    # ----------------------------------------------------------------------------------------------------------
    def max_epochs_for(self, pool):
        """main_nerf.py:147-148: ceil((iters + max(warmup, 0)) / steps_per_epoch)."""
        return int(math.ceil((self.iters + max(self.warmup_steps, 0)) / (pool.total / self.num_rays)))

    def train(self, train_pool, valid_pool=None, max_epochs=None, mark_untrained=True):
        if max_epochs is None:
            max_epochs = self.max_epochs_for(train_pool)
        if mark_untrained and self.model.cuda_ray:                                  # utils.py:768-770
            self.model.mark_untrained_grid(train_pool.poses, train_pool.intrinsics)
            self.ts.invalidate_roi()
        if not self.fast_training and valid_pool is not None:
            self.evaluate_one_epoch(valid_pool)
        # the module tree and the ray pool live for the whole run: out of the collector's way, so that a generation-2
        # pass (10-20 ms) does not stall the launch queue of a 7-ms step
        gc.collect()
        gc.freeze()
        t_train = 0.0
        for epoch in range(self.epoch + 1, max_epochs + 1):
            t0 = time.time()
            self.epoch = epoch
            self.train_one_epoch(train_pool)
            torch.cuda.synchronize()
            t_train += time.time() - t0
            self.log(f"epoch {epoch} time: {time.time() - t0:.2f}[s]")
            if not self.fast_training and self.workspace is not None and epoch % self.eval_interval == 0:
                if valid_pool is not None:
                    self.evaluate_one_epoch(valid_pool)
                self.save_checkpoint(full=True)
        self.log(f"training time: {t_train:.2f}[s]")
        if self.workspace is not None:
            self.save_checkpoint(full=True, remove_old=False)
        return t_train

    def train_one_epoch(self, pool):
        """train_one_epoch2: a fresh permutation of every pixel, ceil(total/num_rays) iterations (the last batch is
        short); each rank of a multi-GPU job takes its contiguous share of every batch."""
        model = self.model
        model.train()
        model.local_step = 0
        pool.shuffle(self.seed * 1000003 + self.epoch)
        steps = pool.steps_per_epoch(self.num_rays)
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self.seed * 7919 + self.epoch)

        def fetch(batch_idx):
            data = pool.batch(batch_idx, self.num_rays, bg_color=self.background_color)
            n = data["rays_o"].shape[0]
            bg = None
            if self.train_rand_bg and pool.channels == 4:                           # utils.py:568-570
                bg = torch.rand(n, 3, device=self.device, generator=gen)
                data = pool.batch(batch_idx, self.num_rays, bg_rand=bg)
            o, d, gt = data["rays_o"], data["rays_d"], data["gt_rgb"]
            if self.world > 1:                                                      # rays sharded across ranks
                lo, hi = n * self.rank // self.world, n * (self.rank + 1) // self.world
                o, d, gt = o[lo:hi].contiguous(), d[lo:hi].contiguous(), gt[lo:hi].contiguous()
                bg = bg[lo:hi] if bg is not None else None
            return o, d, gt, bg, n

        cur = fetch(0)
        for batch_idx in range(steps):
            # the next batch is one cheap launch: having it now lets TrainStep march it underneath this step
            nxt = fetch(batch_idx + 1) if batch_idx + 1 < steps else None
            o, d, gt, bg, n = cur
            loss = self.ts.step(o, d, gt, n_global_rays=n, bg_color=bg,
                                next_rays=None if nxt is None else (nxt[0], nxt[1]))
            total += loss.detach()
            cur = nxt
        total += self.ts.pop_deferred_reg()     # L1 value of the coefficients whose pass TrainStep deferred (defer_adam)
        avg = float(total) / steps                                                  # the epoch's only read-back
        self.stats["loss"].append(avg)
        self.log(f"==> Finished Epoch {self.epoch}, loss {avg:.6f}")
        return avg

    # ----------------------------------------------------------------------------------------------------------
    # evaluation (utils.py:681-735 eval_step / test_step, :1229-1388 evaluate_one_epoch / test)
    # ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_image(self, pool, index, max_steps=None, perturb=False, rows=None):
        """All rays of one pose through the renderer's inference branch -> (pred [H,W,3], depth [H,W], gt or None).
        rows (LongTensor of image rows on the device): only those rows are rendered -> ([len(rows),W,3], [len(rows),W],
        gt rows); every ray is marched, evaluated and composited on its own, so a row's pixels do not depend on which
        other rows share the launch (test(): rows striped over the ranks)."""
        model = self.model
        was_training = model.training
        model.eval()
        self.ts.sync_sharded_parameters()     # "sharded": planes are rebuilt from every slice below (a collective;
        #                                       evaluate_one_epoch / test call this on all ranks before striping)
        self.model.encoder.reset_cahce()      # the training loop only refreshes the occupancy window of the planes
        data = pool.image_rays(index, bg_color=self.background_color)
        H, W = pool.H, pool.W
        if rows is not None:
            sel = lambda t: None if t is None else t.reshape(H, W, -1).index_select(0, rows).reshape(-1, t.shape[-1])
            data = {"rays_o": sel(data["rays_o"]), "rays_d": sel(data["rays_d"]), "gt_rgb": sel(data["gt_rgb"])}
            H = int(rows.numel())
        out = model.render(data["rays_o"].unsqueeze(0), data["rays_d"].unsqueeze(0), staged=True,
                           bg_color=self.background_color, perturb=perturb, dt_gamma=self.dt_gamma,
                           max_steps=max_steps or self.max_steps,
                           **({"infer_min_step": self.infer_min_step} if self.infer_min_step != 1 else {}))
        pred = out["image"].reshape(H, W, 3)
        depth = out["depth"].reshape(H, W)
        gt = data["gt_rgb"].reshape(H, W, 3) if data["gt_rgb"] is not None else None
        if was_training:
            model.train()
        return pred, depth, gt

    def evaluate_one_epoch(self, pool, max_steps=None):
        """PSNR over the pool's images (PSNRMeter: mean of per-image PSNRs) and the mean MSE 'loss'; images are
        striped over the ranks and the two sums all-reduced."""
        meter = PSNRMeter()
        loss = torch.zeros((), dtype=torch.float64, device=self.device)
        self.ts.sync_sharded_parameters()     # before the ranks take different numbers of images
        for i in range(self.rank, pool.B, self.world):
            pred, _, gt = self.render_image(pool, i, max_steps=max_steps)
            meter.update(pred, gt)
            loss += ((pred - gt) ** 2).mean().to(torch.float64)
        acc = torch.cat([meter.state(self.device), loss.reshape(1)])
        if self.world > 1:
            dist.all_reduce(acc, group=self.pg)
        psnr = float(acc[0] / acc[1])
        avg_loss = float(acc[2] / acc[1])
        self.stats["valid_loss"].append(avg_loss)
        self.stats["results"].append(avg_loss)                                      # utils.py:1340-1350 keeps the loss
        self.log(f"++> Evaluate epoch {self.epoch}: PSNR = {psnr:.6f}")
        return {"PSNR": psnr, "loss": avg_loss}

    evaluate = evaluate_one_epoch

    def test(self, pool, save_path=None, max_steps=None):
        """Render every pose (test_step); returns [B,H,W,3] uint8 on the host (on every rank) and, if a path is given,
        rank 0 writes binary PPMs there (the reference writes PNG + MP4 through cv2 / imageio, which this build does not
        carry).  Multi-GPU (the live form of the all_gather at reconstruction/nerf/utils.py:1269-1289): the ROWS of each
        image are striped over the ranks -- rank r renders rows r, r + G, r + 2G, ...: balanced whatever the image shows,
        and it also splits a single pose -- and one all-gather per image of the uint8 rows puts the frame together; the
        pixels are those of the one-rank render bit for bit (tests/test_dist_gpu.py)."""
        frames = []
        self.ts.sync_sharded_parameters()
        H, W, G = pool.H, pool.W, self.world
        per = -(-H // G)                                    # rows per rank (the last stripes of a ragged H are padding)
        mine = (torch.arange(per, device=self.device) * G + self.rank).clamp_(max=H - 1) if G > 1 else None
        for i in range(pool.B):
            pred, _, _ = self.render_image(pool, i, max_steps=max_steps, rows=mine)
            u8 = (pred.clamp(0, 1) * 255).to(torch.uint8)
            if G > 1:
                allr = D.all_gather_slices(u8.unsqueeze(0), self.pg)            # [G, per, W, 3] in rank order
                u8 = allr.permute(1, 0, 2, 3).reshape(per * G, W, 3)[:H]       # row j*G + r <- (r, j)
            frames.append(u8.cpu().numpy())
        frames = np.stack(frames)
        if save_path is not None and self.rank == 0:
            os.makedirs(save_path, exist_ok=True)
            for i, f in enumerate(frames):
                with open(os.path.join(save_path, f"{self.name}_{i:04d}_rgb.ppm"), "wb") as fh:
                    fh.write(f"P6 {f.shape[1]} {f.shape[0]} 255\n".encode())
                    fh.write(f.tobytes())
        return frames

    # ----------------------------------------------------------------------------------------------------------
    # plane dumps (utils.py:1535-1661 save_tensor / get_wavelet_img / save_triplane)
    # ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _normalised(planes):
        """utils.py:1618-1623: per (plane, channel) min-max to [0, 1], then torchvision's adjust_contrast(., 2) of a
        one-channel image = clamp(2 x - mean(x), 0, 1)."""
        planes = planes.detach().float()
        flat = planes.reshape(planes.shape[0], planes.shape[1], -1)
        a = flat.min(dim=-1).values[..., None, None]
        b = flat.max(dim=-1).values[..., None, None]
        x = (planes - a) / (b - a)
        mean = x.mean(dim=(-2, -1), keepdim=True)
        return (2.0 * x - mean).clamp(0.0, 1.0)

    @staticmethod
    def wavelet_image(planes_features, coefs):
        """get_wavelet_img (utils.py:1570-1595): the classic pyramid picture -- normalised LL in the top-left corner, every
        level's |lh| to its right, |hl| below, |hh| diagonal, each band scaled by its own maximum."""
        ll = Trainer._normalised(planes_features)
        for w in coefs:
            w = w.detach().float().abs()
            w = w / w.reshape(w.shape[0], w.shape[1], w.shape[2], -1).max(dim=-1).values[..., None, None]
            top = torch.cat([ll, w[:, :, 0]], dim=3)
            bottom = torch.cat([w[:, :, 1], w[:, :, 2]], dim=3)
            ll = torch.cat([top, bottom], dim=2)
        return ll

    def save_triplane(self, all=False, save_wavelet=False):
        """Trainer.save_triplane (utils.py:1600-1661): grey images of the reconstructed planes (one random channel per
        plane, or all), of the nested zoom planes, and optionally of the wavelet pyramid and of every resolution level,
        under <workspace>/planes.  Written as binary PGM (this build carries no PNG encoder); names, normalisation and
        contrast as in the reference.  Returns the list of files."""
        if self.workspace is None:
            raise RuntimeError("Trainer was built without a workspace")
        # a collective in "sharded" mode (flush + all-gather): on EVERY rank, before the ranks diverge (as save_checkpoint)
        self.ts.sync_sharded_parameters()
        if self.rank != 0:
            return []
        enc = self.model.encoder
        root = os.path.join(self.workspace, "planes")
        os.makedirs(root, exist_ok=True)
        rng = np.random.default_rng(self.seed + self.epoch)
        written = []

        def save_tensor(t, path, prefix="plane"):          # utils.py:1535-1567
            os.makedirs(path, exist_ok=True)
            t = t.cpu()
            for axis in range(t.shape[0]):
                chans = range(t.shape[1]) if all else [int(rng.integers(t.shape[1]))]
                for ch in chans:
                    img = (t[axis, ch] * 255).round().numpy().astype(np.uint8)
                    f = os.path.join(path, f"{prefix}_{self.epoch}_{axis}_{ch}.pgm")
                    with open(f, "wb") as fh:
                        fh.write(f"P5 {img.shape[1]} {img.shape[0]} 255\n".encode())
                        fh.write(img.tobytes())
                    written.append(f)

        with torch.no_grad():
            enc.reset_cahce()
            planes = enc.get_planes()
            upscaled = []
            if enc.upscale_enabled:
                planes, upscaled = planes[0], planes[1:]
            save_tensor(self._normalised(planes), root)
            for idx, pu in enumerate(upscaled):
                save_tensor(self._normalised(pu), root, prefix=f"plane_upscaled_{idx}")
            if save_wavelet:
                save_tensor(self.wavelet_image(enc.planes_features, enc.get_wavelet_features()),
                            os.path.join(root, "wavelet_features"), prefix="wavelet_features")
                enc.reset_cahce()
                for idx, lv in enumerate(enc.get_planes(get_all_resolutions=True)):
                    save_tensor(self._normalised(lv), os.path.join(root, f"levels_{idx}"))
            enc.reset_cahce()
        return written

    # ----------------------------------------------------------------------------------------------------------
    # checkpoints (utils.py:1390-1532)
    # ----------------------------------------------------------------------------------------------------------
    def _torch_optimizer(self):
        """A torch.optim.Adam over model.get_params(lr) (main_nerf.py:119) carrying TrainStep's moments -- only to
        read / write the reference's optimiser state_dict layout."""
        opt = torch.optim.Adam(self.model.get_params(self.lr), betas=(self.ts.b1, self.ts.b2), eps=self.ts.eps)
        return opt

    def _flat_slots(self):
        """parameter -> (flat buffer, segment index) for every parameter TrainStep owns."""
        slots = {}
        for flat in (self.ts.ll, self.ts.coef, self.ts.mlp):
            for k, p in enumerate(flat.params):
                slots[id(p)] = (flat, k)
        return slots

    def optimizer_state_dict(self):
        """torch.optim.Adam state_dict over model.get_params(lr).  No collective in here: in "sharded" mode the caller
        (save_checkpoint, on EVERY rank) has gathered parameters and moments before the ranks diverge."""
        self.ts.flush_deferred()      # not a collective: the deferred coefficients / moments catch up before they are read
        opt = self._torch_optimizer()
        slots = self._flat_slots()
        step = self.ts.opt_steps.detach().to("cpu", torch.float32).reshape(())
        for group in opt.param_groups:
            group["lr"] = self.lr * lr_factor(self.global_step, self.iters, self.warmup_steps)
            group["initial_lr"] = self.lr
            for p in group["params"]:
                flat, k = slots[id(p)]
                o, n = flat.offsets[k], flat.sizes[k]
                opt.state[p] = {"step": step.clone(), "exp_avg": flat.m[o:o + n].view(p.shape).clone(),
                                "exp_avg_sq": flat.v[o:o + n].view(p.shape).clone()}
        return opt.state_dict()

    def load_optimizer_state_dict(self, sd):
        opt = self._torch_optimizer()
        opt.load_state_dict(sd)
        slots = self._flat_slots()
        step = None
        for group in opt.param_groups:
            for p in group["params"]:
                st = opt.state.get(p)
                if not st:
                    continue
                flat, k = slots[id(p)]
                o, n = flat.offsets[k], flat.sizes[k]
                flat.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                flat.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = float(st["step"])
        if step is not None:
            self.ts.opt_steps.fill_(step)

    def scheduler_state_dict(self):
        """state_dict() of the LambdaLR(decay_function) the reference steps once per iteration
        (main_nerf.py:129, utils.py:1173), positioned at the current iteration -- produced by a real torch scheduler
        so the layout is this torch version's."""
        it = self.global_step
        opt = self._torch_optimizer()
        sched = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda k: lr_factor(k, self.iters, self.warmup_steps))
        sched.last_epoch = it
        sched._step_count = it + 1
        sched._last_lr = [self.lr * lr_factor(it, self.iters, self.warmup_steps)] * len(opt.param_groups)
        return sched.state_dict()

    def scaler_state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict() layout."""
        if not self.fp16:
            return {}
        return {"scale": float(self.ts.scale), "growth_factor": 2.0, "backoff_factor": 0.5,
                "growth_interval": self.ts.growth_interval, "_growth_tracker": int(self.ts.growth_tracker)}

    def save_checkpoint(self, name=None, full=False, remove_old=True):
        if self.ckpt_path is None:
            raise RuntimeError("Trainer was built without a workspace")
        # collectives first, on every rank ("sharded": a rank only steps its own (plane, channel) slices and their
        # moments); only then do the ranks part ways
        self.ts.sync_sharded_parameters(moments=full)
        if self.rank != 0:
            return None
        if name is None:
            name = f"{self.name}_ep{self.epoch:04d}"
        state = {"epoch": self.epoch, "global_step": self.global_step, "stats": self.stats}
        if self.model.cuda_ray:
            state["mean_count"] = self.model.mean_count
            state["mean_density"] = self.model.mean_density
        if full:
            state["optimizer"] = self.optimizer_state_dict()
            state["lr_scheduler"] = self.scheduler_state_dict()
            state["scaler"] = self.scaler_state_dict()
        state["model"] = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        path = os.path.join(self.ckpt_path, f"{name}.pth")
        if remove_old:
            self.stats["checkpoints"].append(path)
            if len(self.stats["checkpoints"]) > self.max_keep_ckpt:
                old = self.stats["checkpoints"].pop(0)
                if os.path.exists(old):
                    os.remove(old)
        torch.save(state, path)
        return path

    def load_checkpoint(self, checkpoint=None, model_only=False, _defer_optimizer=False):
        if checkpoint is None:
            found = sorted(glob.glob(os.path.join(self.ckpt_path, f"{self.name}_ep*.pth")))
            if not found:
                self.log("[WARN] No checkpoint found, model randomly initialized.")
                return None
            checkpoint = found[-1]
        ckpt = torch.load(checkpoint, map_location=self.device, weights_only=False)
        if "model" not in ckpt:
            self._load_model_state(ckpt, strict=True)
            return None
        self._load_model_state(ckpt["model"], strict=False)
        if self.model.cuda_ray:
            self.model.mean_count = ckpt.get("mean_count", self.model.mean_count)
            self.model.mean_density = ckpt.get("mean_density", self.model.mean_density)
        if model_only:
            return None
        self.stats = ckpt["stats"]
        self.epoch = ckpt["epoch"]
        if _defer_optimizer:
            return ckpt
        self._restore_training_state(ckpt)
        return None

    def _load_model_state(self, sd, strict):
        """load_state_dict with the reference's tolerance (strict=False: missing new wavelet levels stay at their
        zero init, utils.py:1481) plus a shape filter: a tensor whose shape changed with the resolution is skipped
        with a warning instead of raising."""
        if hasattr(self, "ts"):
            self.ts.flush_deferred()    # pending deferred updates belong to the values about to be replaced
        own = self.model.state_dict()
        ok = {k: v for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
        skipped = [k for k in sd if k not in ok]
        missing, unexpected = self.model.load_state_dict(ok, strict=False)
        if strict and (missing or skipped):
            raise RuntimeError(f"checkpoint does not match the model: missing {missing}, skipped {skipped}")
        if missing:
            self.log(f"[WARN] missing keys: {missing}")
        if skipped:
            self.log(f"[WARN] unexpected / reshaped keys: {skipped}")
        if hasattr(self, "ts"):
            self.ts.invalidate_roi()
        self.model.encoder.reset_cahce()

    def _restore_training_state(self, ckpt):
        self.ts.global_step = int(ckpt.get("global_step", 0))
        if "optimizer" in ckpt:
            try:
                self.load_optimizer_state_dict(ckpt["optimizer"])
            except Exception as e:   # utils.py:1511-1517: a stage with new shapes keeps a fresh optimiser
                self.log("[WARN] Failed to load optimizer.", str(e))
        sc = ckpt.get("scaler") or {}
        if self.fp16 and "scale" in sc:
            self.ts.scale.fill_(float(sc["scale"]))
            self.ts.growth_tracker.fill_(int(sc.get("_growth_tracker", 0)))


def train_stages(make_model, make_pools, stages, workspace, name="trinerflet", **common):
    """The stage loop of main_nerf.py:168-205: every stage builds a fresh model at its own resolution / wavelet
    depth and a fresh Trainer with `--ckpt latest_model`, so it starts from the previous stage's planes (new
    finest level zero), a new optimiser, scaler and schedule.  `stages` is a list of dicts with the per-stage
    options (iters, num_rays, triplane_resolution, triplane_wavelet_levels, warmup_steps, lr,
    wavelet_regularization); `make_model(stage)` and `make_pools(stage)` supply the model and
    (train_pool, valid_pool).  Returns the last Trainer."""
    trainer = None
    for stage in stages:
        model = make_model(stage)
        train_pool, valid_pool = make_pools(stage)
        opts = dict(common)
        for k in ("iters", "num_rays", "warmup_steps", "lr", "wavelet_regularization"):
            if k in stage:
                opts[k] = stage[k]
        trainer = Trainer(name, model, workspace=workspace, use_checkpoint="latest_model", **opts)
        trainer.train(train_pool, valid_pool)
    return trainer
