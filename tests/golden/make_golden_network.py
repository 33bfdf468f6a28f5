"""Generates tests/golden/network_reference.npz by RUNNING THE REFERENCE's own Python in this container (never on
the GPU box; /root/reference does not travel):

    python tests/golden/make_golden_network.py

What runs, imported from /root/reference UNMODIFIED:
  reconstruction/nerf/network.py      NeRFNetwork.forward / density / color            (F-MLP,  SURVEY 8(a) A5/A6)
  reconstruction/nerf/renderer.py     NeRFRenderer.run (pure-torch renderer)           (F-RUN,  A14)
                                      NeRFRenderer.run_cuda, training + eval branch    (F-STEP / F-INFER, A10 / A11 glue)
  reconstruction/nerf/utils.py        Trainer.train_step + backward + Adam + LambdaLR  (F-STEP, A13)
  reconstruction/triplaneencoder/...  TriPlaneVolume (planes, lookup, wavelet features)
  reconstruction/encoding.py, activation.py, run_utils.py (the README's flag defaults)
  aux_libs/raymarching/raymarching.py the nine autograd Functions (budget / alignment / zero-fill rules)
  aux_libs/shencoder/sphere_harmonics.py SHEncoder

What cannot run here and is stood in for:
  * `_raymarching` / `_shencoder` (CUDA-only pybind modules, raymarching.cu / shencoder.cu): modules with the ten +
    two prototypes of raymarching.h:7-17 / shencoder.h backed by the C oracle (oracle/trinerflet_oracle.c).  So these
    fixtures pin the reference's PYTHON (MLP composition, run / run_cuda glue, loss, regulariser, optimiser wiring,
    wrapper budget rules) -- the CUDA kernels themselves stay pinned only by restatement;
  * `pytorch_wavelets` (un-vendored, 1.3.0): an ADAPTER that forwards DWTInverse to the real PyWavelets
    (`pywt.idwt2`, mode='zero', under /opt/conda/bin/python3.9); its backward is PyWavelets' `dwt2` with the
    transposed filter bank (dec = reversed rec), checked below against <Ax,y> = <x,A^T y>;
  * third-party packages absent from this image (cv2, tensorboardX, mcubes, ...): empty placeholder modules;
  * `torch.Tensor.cuda` returns the tensor itself (the wrappers call `.cuda()` on their inputs; there is no GPU here).
Everything is fp32 on the CPU (`fp16=False`: the reference's own no-autocast path).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import cref  # noqa: E402
import make_golden_reference as mgr  # noqa: E402  (the pywt subprocess bridge)

C, R, SCALE, HID, BOUND = 16, 32, 4, 64, 1.5
N_RAYS, MAX_STEPS, LAM, BG = 256, 64, 0.2, 0.25


# ------------------------------------------------------------------------------------------------------------------
# stand-ins (see the module docstring)
# ------------------------------------------------------------------------------------------------------------------
THIRD_PARTY = {"imageio", "tensorboardX", "cv2", "trimesh", "mcubes", "lpips", "torch_ema", "torchmetrics",
               "torchvision", "matplotlib", "kornia", "PIL", "open3d", "plyfile", "nerfacc", "tinycudann", "clip",
               "dearpygui", "skimage", "pycolmap", "pytorch_lightning", "imageio_ffmpeg"}


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, n):
        return _Anything()


class _Placeholder(types.ModuleType):
    __path__ = []

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Anything


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] not in THIRD_PARTY:
            return None
        return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Placeholder(spec.name)

    def exec_module(self, module):
        pass


_WORKER_ADJ = r"""
import sys, numpy as np, pywt
wave, path = sys.argv[1:3]
w = pywt.Wavelet(wave)
wt = pywt.Wavelet(wave + '_T', filter_bank=[w.rec_lo[::-1], w.rec_hi[::-1], w.rec_lo, w.rec_hi])
g = np.load(path)['g']
ll, hs = [], []
for b in range(g.shape[0]):
    l_, h_ = [], []
    for c in range(g.shape[1]):
        a, (h, v, dd) = pywt.dwt2(g[b, c], wt, mode='zero')
        l_.append(a); h_.append(np.stack([h, v, dd]))
    ll.append(np.stack(l_)); hs.append(np.stack(h_))
np.savez(path + '.out.npz', yl=np.stack(ll), yh=np.stack(hs))
"""


def _pywt_adjoint(wave, g):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.npz")
        np.savez(path, g=g)
        subprocess.check_call([mgr.PY39, "-W", "ignore", "-c", _WORKER_ADJ, wave, path])
        d = np.load(path + ".out.npz")
        return d["yl"], d["yh"]


class _IDWT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yl, yh, wave):
        ctx.wave, ctx.shapes = wave, (yl.shape, yh.shape)
        out = mgr._call_pywt("idwt2", wave, yl=yl.detach().double().numpy(), yh=yh.detach().double().numpy())
        return torch.from_numpy(out).to(yl.dtype)

    @staticmethod
    def backward(ctx, g):
        dyl, dyh = _pywt_adjoint(ctx.wave, g.detach().double().numpy())
        assert dyl.shape == tuple(ctx.shapes[0]) and dyh.shape == tuple(ctx.shapes[1]), (dyl.shape, ctx.shapes)
        return torch.from_numpy(dyl).to(g.dtype), torch.from_numpy(dyh).to(g.dtype), None


class DWTInverse(torch.nn.Module):
    def __init__(self, wave="db1", mode="zero"):
        super().__init__()
        assert mode == "zero"
        self.wave = wave

    def forward(self, coeffs):
        yl, (yh,) = coeffs
        return _IDWT.apply(yl, yh, self.wave)


def _check_adjoint(wave="bior6.8"):
    g = torch.Generator().manual_seed(0)
    yl = torch.randn(1, 2, 12, 12, generator=g, dtype=torch.float64, requires_grad=True)
    yh = torch.randn(1, 2, 3, 12, 12, generator=g, dtype=torch.float64, requires_grad=True)
    out = _IDWT.apply(yl, yh, wave)
    cot = torch.randn(out.shape, generator=g, dtype=torch.float64)
    dyl, dyh = torch.autograd.grad(out, (yl, yh), cot)
    lhs = float((out * cot).sum())
    rhs = float((yl * dyl).sum() + (yh * dyh).sum())
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs)), (lhs, rhs)
    print(f"adapter adjoint identity ({wave}): <Ax,y> = {lhs:.12f}, <x,A^T y> = {rhs:.12f}")


def _np(t):
    return t.detach().cpu().numpy().copy()      # a copy: parameters are updated in place later on


def _oracle_raymarching():
    """`_raymarching` with the prototypes of raymarching.h:7-17, every function the C oracle's restatement of the
    kernel of that name.  Outputs are written into the caller's tensors, as the CUDA module does."""
    m = types.ModuleType("_raymarching")
    m.log = {}

    def put(dst, src):
        dst.copy_(torch.from_numpy(np.ascontiguousarray(src)).view_as(dst) if dst.numel() == src.size
                  else torch.from_numpy(np.ascontiguousarray(src)))

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        n, f = cref.near_far_from_aabb(_np(rays_o), _np(rays_d), _np(aabb), min_near)
        put(nears, n), put(fars, f)

    def morton3D(coords, N, indices):
        put(indices, cref.morton3D(_np(coords)))

    def morton3D_invert(indices, N, coords):
        put(coords, cref.morton3D_invert(_np(indices)))

    def packbits(grid, N, density_thresh, bitfield):
        put(bitfield, cref.packbits(_np(grid), density_thresh))

    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C_, H, M, nears, fars, xyzs, dirs, deltas,
                         rays, counter, noises):
        m.log["noises"] = _np(noises).copy()
        cnt = _np(counter).copy()
        x, d, l, r, c = cref.march_rays_train(_np(rays_o), _np(rays_d), bound, _np(grid), C_, H, _np(nears), _np(fars),
                                              _np(noises), M, dt_gamma, max_steps, counter=cnt)
        put(xyzs, x), put(dirs, d), put(deltas, l), put(rays, r), put(counter, c)

    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        w, d, i = cref.composite_rays_train_forward(_np(sigmas), _np(rgbs), _np(deltas), _np(rays), T_thresh)
        put(weights_sum, w), put(depth, d), put(image, i)

    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                      T_thresh, grad_sigmas, grad_rgbs):
        gs, gc = cref.composite_rays_train_backward(_np(grad_weights_sum), _np(grad_image), _np(sigmas), _np(rgbs),
                                                    _np(deltas), _np(rays), _np(weights_sum), _np(image), T_thresh)
        put(grad_sigmas, gs), put(grad_rgbs, gc)

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C_, H, grid, nears,
                   fars, xyzs, dirs, deltas, noises):
        x, d, l = cref.march_rays(n_alive, n_step, _np(rays_alive), _np(rays_t), _np(rays_o), _np(rays_d), bound,
                                  _np(grid), C_, H, _np(nears), _np(fars), _np(noises), -1, dt_gamma, max_steps)
        k = x.shape[0]
        xyzs[:k].copy_(torch.from_numpy(x)), dirs[:k].copy_(torch.from_numpy(d)), deltas[:k].copy_(torch.from_numpy(l))

    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        ra, rt = _np(rays_alive).copy(), _np(rays_t).copy()
        w, d, i = _np(weights_sum).copy(), _np(depth).copy(), _np(image).copy()
        cref.composite_rays(n_alive, n_step, ra, rt, _np(sigmas), _np(rgbs), _np(deltas), w, d, i, T_thresh)
        put(rays_alive, ra), put(rays_t, rt), put(weights_sum, w), put(depth, d), put(image, i)
        m.log.setdefault("alive", []).append(ra.copy())

    def sph_from_ray(*a):
        raise NotImplementedError

    for f in (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, march_rays_train,
              composite_rays_train_forward, composite_rays_train_backward, march_rays, composite_rays):
        setattr(m, f.__name__, f)
    return m


def _oracle_shencoder():
    m = types.ModuleType("_shencoder")

    def sh_encode_forward(inputs, outputs, B, D, C_, dy_dx=None):
        assert D == 3 and C_ == 4 and dy_dx is None
        outputs.copy_(torch.from_numpy(cref.sh4(_np(inputs))))

    def sh_encode_backward(*a):
        raise NotImplementedError

    m.sh_encode_forward, m.sh_encode_backward = sh_encode_forward, sh_encode_backward
    return m


def _sphere_bitfield(cascade, H, bound, radius):
    """Cells whose centre lies inside a sphere, both cascades, Morton order (the layout packbits produces)."""
    ax = np.arange(H, dtype=np.int32)
    coords = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    idx = cref.morton3D(coords)
    grid = np.zeros((cascade, H ** 3), np.float32)
    for c in range(cascade):
        s = min(2.0 ** c, bound)
        centre = ((coords + 0.5) / H * 2 - 1) * s
        grid[c, idx] = (np.linalg.norm(centre, axis=1) < radius).astype(np.float32)
    return cref.packbits(grid, 0.5), grid


def import_reference(check_adjoint=True):
    """Installs the stand-ins of the module docstring and imports the reference's own modules from /root/reference,
    unmodified.  Returns (NeRFNetwork, nerf.utils, get_params, the `_raymarching` stand-in).  Also used by
    make_golden_grid.py."""
    sys.meta_path.append(_Finder())
    adapter = types.ModuleType("pytorch_wavelets")
    adapter.DWTForward, adapter.DWTInverse = mgr.DWTForward, DWTInverse
    sys.modules["pytorch_wavelets"] = adapter
    if check_adjoint:
        _check_adjoint()
    rm_native = sys.modules["_raymarching"] = _oracle_raymarching()
    sys.modules["_shencoder"] = _oracle_shencoder()
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, os.path.join(REF, "reconstruction"))
    sys.path.insert(0, os.path.join(REF, "aux_libs"))
    os.chdir(os.path.join(REF, "reconstruction"))
    from nerf.network import NeRFNetwork            # the reference's classes, unmodified
    from nerf import utils as U
    from run_utils import get_params
    import raymarching as ref_rm
    assert ref_rm.raymarching._backend is rm_native and NeRFNetwork.__module__ == "nerf.network"
    assert sys.modules["nerf.network"].__file__.startswith(REF)
    return NeRFNetwork, U, get_params, rm_native


MODEL_KEYS = ['triplane_channels', 'triplane_resolution', 'triplane_wavelet_levels', 'wavelet_type', 'hidden_dim',
              'hidden_dim_color', 'hidden_dim_bg', 'learn_rotation_axis', 'dropout', 'inner_bound', 'lbound_auto_scale',
              'upscale_ratio_bound', 'upscale_levels', 'density_blob_scale', 'density_blob_std', 'mlp_weight_decay',
              'wavelet_base_resolution', 'nerfacc_renderer']


def main():
    NeRFNetwork, U, get_params, rm_native = import_reference()

    out = {}
    torch.manual_seed(0)
    workspace = "/tmp/_tnl_golden_ws"
    sys.argv = ["main_nerf.py", "--path", "/nonexistent", "--workspace", workspace, "--cuda_ray", "--bound", str(BOUND),
                "--scale", "1", "--dt_gamma", "0", "--iters", "100", "--num_rays", str(N_RAYS), "--background_color",
                str(BG), "--triplane_wavelet", "--triplane_channels", str(C), "--triplane_wavelet_levels", str(SCALE),
                "--triplane_resolution", str(R), "--wavelet_regularization", str(LAM), "--ckpt", "scratch",
                "--ema_decay", "-1", "--warmup_steps", "0", "--max_steps", str(MAX_STEPS), "--fast_training"]
    opt = get_params()
    for k, v in list(vars(opt).items()):                      # main_nerf.py:172-205: stage 0 of the list-valued flags
        if isinstance(v, list) and len(v) == 1:
            setattr(opt, k, v[0])
    keys = MODEL_KEYS
    model = NeRFNetwork(encoding="triplane_wavelet", bound=opt.bound, cuda_ray=True, density_scale=opt.density_scale,
                        min_near=opt.min_near, density_thresh=opt.density_thresh, bg_radius=opt.bg_radius,
                        **{k: vars(opt)[k] for k in keys})
    enc = model.encoder
    assert opt.wavelet_type == "bior6.8" and model.hidden_dim == HID
    assert [tuple(p.shape) for p in enc.planes_features_wavelet_coefs] == [(3, C, 3, 8, 8), (3, C, 3, 16, 16)]
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for i, p in enumerate(enc.planes_features_wavelet_coefs):
            p.copy_(torch.randn(p.shape, generator=g) * 0.05 * 0.5 ** i)
        enc.planes_features.copy_(torch.randn(enc.planes_features.shape, generator=g) * 0.25)
    names = ["sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.1.weight", "color_net.2.weight"]
    sd = dict(model.named_parameters())
    out["cfg"] = np.array([C, R, SCALE, HID, N_RAYS, MAX_STEPS], np.int64)
    out["cfg_f"] = np.array([BOUND, LAM, BG, opt.lr, opt.min_near, opt.density_scale], np.float64)
    out["param/ll"] = _np(enc.planes_features)
    for i, p in enumerate(enc.planes_features_wavelet_coefs):
        out[f"param/coef{i}"] = _np(p)
    for k, n in enumerate(names):
        out[f"param/W{k}"] = _np(sd[n])

    # ---- planes of these parameters (reference get_planes through the adapter)
    enc.reset_cahce()
    with torch.no_grad():
        planes = enc.get_planes().clone()
    out["planes"] = _np(planes)
    enc.reset_cahce()

    # ---- F-MLP: NeRFNetwork.forward / density / color on given planes, positions, directions, + VJP
    M = 384
    xyz = (torch.rand(M, 3, generator=g) * 2 - 1) * BOUND
    xyz[:8] = torch.tensor([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=torch.float32) * BOUND
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    planes_leaf = planes.clone().requires_grad_(True)
    enc.last_used_planes = planes_leaf
    model.train()
    sigma, rgb = model(xyz, dirs)
    cot_s = torch.randn(M, generator=g) * 0.1
    cot_c = torch.randn(M, 3, generator=g)
    params = [sd[n] for n in names]
    grads = torch.autograd.grad([sigma, rgb], [planes_leaf] + params, [cot_s, cot_c])
    dens = model.density(xyz)
    mask = torch.rand(M, generator=g) > 0.4
    col = model.color(xyz, dirs, mask=mask, geo_feat=dens["geo_feat"])
    out.update({"mlp/xyz": _np(xyz), "mlp/dirs": _np(dirs), "mlp/sigma": _np(sigma), "mlp/rgb": _np(rgb),
                "mlp/cot_sigma": _np(cot_s), "mlp/cot_rgb": _np(cot_c), "mlp/dplanes": _np(grads[0]),
                "mlp/density_sigma": _np(dens["sigma"]), "mlp/geo_feat": _np(dens["geo_feat"]),
                "mlp/mask": _np(mask), "mlp/color_masked": _np(col)})
    for k in range(5):
        out[f"mlp/dW{k}"] = _np(grads[1 + k])
    enc.reset_cahce()

    # ---- rays: 4 hemisphere cameras looking at the origin (get_rays of the reference, utils.py:65-149)
    poses = []
    for a in range(4):
        th, ph = 0.4 + 0.35 * a, 1.3 * a
        eye = 3.2 * np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, fwd, eye
        poses.append(pose)
    poses = np.stack(poses).astype(np.float32)
    Hh = Ww = 24
    intr = np.array([30.0, 30.0, Ww / 2, Hh / 2], np.float32)
    r = U.get_rays(torch.from_numpy(poses), intr, Hh, Ww, -1)
    pick = torch.randperm(4 * Hh * Ww, generator=g)[:N_RAYS]
    rays_o = r["rays_o"].reshape(-1, 3)[pick].contiguous()
    rays_d = r["rays_d"].reshape(-1, 3)[pick].contiguous()
    rays_d[:2] = torch.tensor([[0.0, 0.0, -1.0], [0.6, 0.8, 0.0]])          # axis-parallel directions
    rays_o[2] = rays_o[2] + 20.0                                             # a ray that misses the box
    images = torch.rand(1, N_RAYS, 4, generator=g)
    out.update({"rays/o": _np(rays_o), "rays/d": _np(rays_d), "rays/images": _np(images)})

    # ---- F-RUN: NeRFRenderer.run (A14), eval mode (aabb_infer, deterministic sample_pdf), no / with upsampling
    model.eval()
    for tag, ups in (("run64", 0), ("run32u16", 16)):
        enc.reset_cahce()
        with torch.no_grad():
            res = model.run(rays_o[None], rays_d[None], num_steps=64 if ups == 0 else 32, upsample_steps=ups,
                            bg_color=BG, perturb=False)
        out.update({f"{tag}/image": _np(res["image"][0]), f"{tag}/depth": _np(res["depth"][0]),
                    f"{tag}/weights_sum": _np(res["weights_sum"])})

    # ---- occupancy: analytic sphere, both cascades
    bitfield, grid = _sphere_bitfield(model.cascade, model.grid_size, BOUND, 0.9)
    model.density_bitfield.copy_(torch.from_numpy(bitfield))
    out["bitfield"] = bitfield

    # ---- F-INFER: run_cuda eval branch (renderer.py:324-374) -- the reference's loop policy over the oracle kernels
    enc.reset_cahce()
    rm_native.log.clear()
    with torch.no_grad():
        res = model.render(rays_o[None], rays_d[None], staged=True, bg_color=BG, perturb=False, dt_gamma=0,
                           max_steps=MAX_STEPS, T_thresh=1e-4)
    out.update({"infer/image": _np(res["image"][0]), "infer/depth": _np(res["depth"][0]),
                "infer/weights_sum": _np(res["weights_sum"][0]),
                "infer/n_alive_after": np.array([int((a >= 0).sum()) for a in rm_native.log["alive"]], np.int64)})
    enc.reset_cahce()

    # ---- F-STEP: two iterations of train_one_epoch2's body (utils.py:1134-1175) around Trainer.train_step
    os.makedirs(workspace, exist_ok=True)
    criterion = torch.nn.MSELoss(reduction='none')
    optimizer = lambda model: torch.optim.Adam(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)   # main_nerf.py:119
    scheduler = lambda optimizer: torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: U.decay_function(it, opt))
    trainer = U.Trainer('trinerflet', opt, model, device=torch.device('cpu'), workspace=workspace, optimizer=optimizer,
                        criterion=criterion, ema_decay=None, fp16=False, lr_scheduler=scheduler,
                        scheduler_update_every_step=True, metrics=[], use_checkpoint="scratch", mute=True,
                        use_tensorboardX=False)
    trainer.error_map = None
    model.train()
    model.mean_count = 0
    data = {"rays_o": rays_o[None], "rays_d": rays_d[None], "images": images}
    for it in range(2):
        torch.manual_seed(100 + it)                      # seeds the wrapper's torch.rand(N) perturbation
        enc.reset_cahce()
        enc.get_planes()
        trainer.optimizer.zero_grad()
        preds, truths, loss, aux = trainer.train_step({k: v.clone() for k, v in data.items()})
        enc.reset_cahce()
        trainer.scaler.scale(loss).backward()
        slot = (model.local_step - 1) % 16
        out.update({f"step{it}/noises": rm_native.log["noises"], f"step{it}/pred": _np(preds[0]),
                    f"step{it}/gt": _np(truths[0]), f"step{it}/loss": np.array(float(loss)),
                    f"step{it}/mse": np.array(aux["mse"]), f"step{it}/wavelet_reg": np.array(aux["wavelet_reg"]),
                    f"step{it}/counter": _np(model.step_counter[slot]),
                    f"step{it}/lr": np.array(trainer.optimizer.param_groups[0]["lr"]),
                    f"step{it}/mean_count": np.array(model.mean_count)})
        out[f"step{it}/g_ll"] = _np(enc.planes_features.grad)
        for i, p in enumerate(enc.planes_features_wavelet_coefs):
            out[f"step{it}/g_coef{i}"] = _np(p.grad)
        for k, n in enumerate(names):
            out[f"step{it}/g_W{k}"] = _np(sd[n].grad)
        trainer.scaler.step(trainer.optimizer)
        trainer.scaler.update()
        trainer.lr_scheduler.step()
        out[f"step{it}/ll_after"] = _np(enc.planes_features)
        for i, p in enumerate(enc.planes_features_wavelet_coefs):
            out[f"step{it}/coef{i}_after"] = _np(p)
        for k, n in enumerate(names):
            out[f"step{it}/W{k}_after"] = _np(sd[n])
        # the second iteration runs under a sample budget (raymarching.py:200-203): the running-mean rule of
        # renderer.py:537-540 over the one slot used so far
        model.mean_count = int(model.step_counter[slot, 0].item())
    # training-branch depth normalisation / background mix of run_cuda (renderer.py:317-318), taken separately since
    # train_step drops the depth: same rays, same noise as iteration 1, parameters after two updates
    torch.manual_seed(101)
    enc.reset_cahce()
    with torch.no_grad():
        enc.get_planes()
        res = model.render(rays_o[None], rays_d[None], staged=False, bg_color=BG, perturb=True, force_all_rays=False,
                           dt_gamma=0, max_steps=MAX_STEPS)
    out.update({"glue/image": _np(res["image"][0]), "glue/depth": _np(res["depth"][0]),
                "glue/weights_sum": _np(res["weights_sum"]), "glue/mean_count": np.array(model.mean_count),
                "glue/noises": rm_native.log["noises"]})
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim > 0 and k != "cfg_f" else v) for k, v in out.items()}
    path = os.path.join(HERE, "network_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
