"""NeRFNetwork -- mirror of reconstruction/nerf/network.py (reference): same constructor keywords,
sub-module names (encoder, encoder_dir, sigma_net.{0,1}, color_net.{0,1,2}: state-dict compatible) and
methods forward / density / color / get_params.

forward() and density() run the fused HIP field kernel whenever the configuration is one the kernel is
built for (triplane_wavelet encoder, channels 16/32/48, hidden 64 or 128: every README configuration);
other shapes take the modular path (HIP lookup + HIP SH + torch nn.Linear), which is also what `color()`
uses (only NeRFRenderer.run, the non-cuda_ray renderer, calls it).  The background network
(bg_radius > 0, network.py:79-100) is not on the hot path and raises NotImplementedError.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..activation import trunc_exp
from ..encoding import get_encoder
from . import field as _field
from .. import occupancy
from .renderer import NeRFRenderer


class _WindowProvider:
    """encoder.window_provider: the owning network's occupancy window, through a WEAK reference -- a bound method would
    close a cycle network -> encoder -> network, and a model's 5 GB of device tensors would wait for the cycle collector
    instead of being freed with the last reference."""

    def __init__(self, owner):
        self._ref = weakref.ref(owner)

    def __call__(self):
        owner = self._ref()
        return owner._occupancy_window() if owner is not None else None

    def __deepcopy__(self, memo):           # copy.deepcopy(model): the copy's encoder asks the COPY
        owner = self._ref()
        return _WindowProvider(memo.get(id(owner), owner)) if owner is not None else self

    def __reduce__(self):                   # pickled models: re-attached by NeRFNetwork.__setstate__
        return (_no_window, ())


def _no_window():
    return None


class NeRFNetwork(NeRFRenderer):
    def __init__(self,
                 encoding="triplane_wavelet",
                 encoding_dir="sphere_harmonics",
                 encoding_bg="hashgrid",
                 num_layers=2,
                 hidden_dim=64,
                 geo_feat_dim=15,
                 num_layers_color=3,
                 hidden_dim_color=64,
                 num_layers_bg=2,
                 hidden_dim_bg=64,
                 bound=1,
                 density_blob_scale=0,
                 density_blob_std=0.5,
                 mlp_weight_decay=0,
                 nerfacc_renderer=False,
                 **kwargs,
                 ):
        super().__init__(bound, **kwargs)
        if self.bg_radius > 0:
            raise NotImplementedError("background network (bg_radius > 0) is outside the hot-path tier")
        if nerfacc_renderer:
            raise NotImplementedError("nerfacc renderer is outside the hot-path tier")
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound, bound=bound, **kwargs)
        if hasattr(self.encoder, "window_provider"):
            # a training iteration's get_planes() (under autograd) rebuilds only what this density grid's samples can read
            self.encoder.window_provider = _WindowProvider(self)

        # sigma network (network.py:37-52): bias-free Linear layers
        sigma_net = []
        for l in range(num_layers):
            in_dim = self.in_dim if l == 0 else hidden_dim
            out_dim = 1 + self.geo_feat_dim if l == num_layers - 1 else hidden_dim
            sigma_net.append(nn.Linear(in_dim, out_dim, bias=False))
        self.sigma_net = nn.ModuleList(sigma_net)

        # colour network (network.py:55-76)
        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        color_net = []
        for l in range(num_layers_color):
            in_dim = self.in_dim_dir + self.geo_feat_dim if l == 0 else hidden_dim_color
            out_dim = 3 if l == num_layers_color - 1 else hidden_dim_color
            color_net.append(nn.Linear(in_dim, out_dim, bias=False))
        self.color_net = nn.ModuleList(color_net)
        self.bg_net = None

        self.density_blob_scale = density_blob_scale
        self.density_blob_std = density_blob_std
        self.mlp_weight_decay = mlp_weight_decay
        self.force_modular = False  # tests flip this to compare the two GPU paths

    def __setstate__(self, state):
        super().__setstate__(state)
        if hasattr(self.encoder, "window_provider"):
            self.encoder.window_provider = _WindowProvider(self)

    # ------------------------------------------------------------------------------------------
    def _fused_ok(self):
        enc = self.encoder
        ok = (not self.force_modular and hasattr(enc, 'get_planes_texel_major') and enc.is_plain() and enc.dropout is None
              and self.num_layers == 2 and self.num_layers_color == 3 and self.geo_feat_dim == 15
              and getattr(self.encoder_dir, 'degree', 0) == 4 and self.density_blob_scale <= 1e-5
              and _field.supported(enc.number_of_features, self.hidden_dim, self.hidden_dim_color))
        if not ok and not self.force_modular and not getattr(self, "_modular_warned", False):
            # the reference's MLP is generic (network.py:32-76); the hand-written field exists for the three README shapes
            self._modular_warned = True
            import warnings
            warnings.warn(
                f"trinerflet_amd: this network (channels {getattr(enc, 'number_of_features', '?')}, hidden {self.hidden_dim} / "
                f"{self.hidden_dim_color}, layers {self.num_layers} / {self.num_layers_color}, geo {self.geo_feat_dim}) is outside "
                "the fused field's shapes {(16, 64), (32, 64), (48, 128)} x 2 / 3 layers x geo 15 x SH degree 4 (or the encoder uses "
                "an option the fused lookup does not take): it runs the MODULAR path -- HIP triplane lookup + torch nn.Linear "
                "(rocBLAS) + HIP SH / compositing -- correct, several times slower per sample", RuntimeWarning, stacklevel=3)
        return ok

    def density_op(self, x, density):
        if self.density_blob_scale > 1e-5:  # network.py:111-117 (x is the ENCODED position there, kept as is)
            w = self.density_blob_scale * torch.exp(-0.5 * x.pow(2).sum(dim=-1) / self.density_blob_std ** 2)
            density = density * w
        return density

    def _sigma_mlp(self, x):
        h = self.encoder(x, bound=self.bound)
        enc = h
        for l in range(self.num_layers):
            h = self.sigma_net[l](h)
            if l != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        sigma = trunc_exp(self.density_op(enc, h[..., 0]))
        return sigma, h[..., 1:]

    def _color_mlp(self, d, geo_feat):
        d = self.encoder_dir(d)
        h = torch.cat([d, geo_feat.to(d.dtype)], dim=-1)
        for l in range(self.num_layers_color):
            h = self.color_net[l](h.to(self.color_net[l].weight.dtype) if not torch.is_autocast_enabled() else h)
            if l != self.num_layers_color - 1:
                h = F.relu(h, inplace=True)
        return torch.sigmoid(h)

    def forward(self, x, d):
        # x: [N, 3] in [-bound, bound], d: [N, 3] unit directions -> sigma [N], color [N, 3]  (network.py:118-147)
        if self._fused_ok():
            enc = self.encoder
            Ws = (self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight, self.color_net[1].weight,
                  self.color_net[2].weight)
            if torch.is_grad_enabled() and enc.planes_features.requires_grad and enc.plane_resolution % 32 == 0:
                # training through autograd (the reference's own loop): the graph runs through the (3,C,R,R) planes of
                # get_planes(); the sampler's texel-major copy is data, and the kernel's backward hands the planes'
                # gradient back in (3,C,R,R) directly
                planes_cm = enc.get_planes()
                mw = tuple(int(v) for v in self._march_window) if self._march_window is not None else None
                pw = getattr(planes_cm, "_tnl_window", None)
                if pw is not None and pw != mw:
                    # only a window of the planes exists and these positions are not known to lie inside it (a call from
                    # outside run_cuda, or a density grid that changed since get_planes()): whole, differentiable planes
                    planes_cm, pw = enc.get_planes_whole(), None
                with torch.no_grad():
                    # called from run_cuda on a marched batch: only the occupancy window of the copy is made
                    tm = enc.get_planes_texel_major(window=mw)
                return _field.fused_field(tm, x, d, *Ws, self.bound, planes_cm, self._march_count, pw)
            tm = enc.get_planes_texel_major()
            return _field.fused_field(tm, x, d, *Ws, self.bound, None, self._march_count)
        sigma, geo_feat = self._sigma_mlp(x)
        return sigma, self._color_mlp(d, geo_feat)

    def _occupancy_window(self):
        """occupancy.window of the current density bitfield at the encoder's plane resolution, cached until the bitfield
        changes (one small host read-back per density-grid refresh); None: whole planes."""
        enc = self.encoder
        R = getattr(enc, "plane_resolution", 0)
        if not self.use_occupancy_window or not self.cuda_ray or R % 64 != 0 or not hasattr(enc, "get_planes_texel_major"):
            return None
        bf = self.density_bitfield
        key = (bf.data_ptr(), bf._version, R)
        if self._occ_window_key != key:
            self._occ_window = occupancy.window(bf, self.cascade, self.grid_size, self.bound, R)
            self._occ_window_key = key
        return self._occ_window

    @torch.no_grad()
    def packed_weights(self):
        """The five weight matrices as fp16 MFMA fragments (field_common.h); a render loop packs them once."""
        enc = self.encoder
        return _field.pack_weights(self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight,
                                   self.color_net[1].weight, self.color_net[2].weight, enc.number_of_features,
                                   self.hidden_dim)

    @torch.no_grad()
    def field_rows(self, xyzs, dirs, m_actual, packed=None):
        """sigma, rgb of the first m_actual[0] rows (device int32) of fixed-capacity buffers: the inference loop's
        field query when its sizes live on the device (NeRFRenderer.run_cuda).  Fused path only.  packed: the result of
        packed_weights() (the loop's weights do not change between its iterations)."""
        enc = self.encoder
        tm = enc.get_planes_texel_major()
        if packed is None:
            packed = self.packed_weights()
        sigma, rgb, _ = _field.field_forward(tm, xyzs, dirs, packed, float(self.bound), enc.number_of_features,
                                             enc.plane_resolution, self.hidden_dim, m_actual=m_actual)
        return sigma, rgb

    def density(self, x):
        # network.py:149-166
        if self._fused_ok() and not torch.is_grad_enabled():
            enc = self.encoder
            tm = enc.get_planes_texel_major()
            packed = _field.pack_weights(self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight,
                                         self.color_net[1].weight, self.color_net[2].weight, enc.number_of_features,
                                         self.hidden_dim)
            x = x.detach().to(torch.float32).contiguous()
            sigma, geo, _ = _field.field_forward(tm, x, None, packed, float(self.bound), enc.number_of_features,
                                                 enc.plane_resolution, self.hidden_dim, geo_out=True)
            return {'sigma': sigma, 'geo_feat': geo}
        sigma, geo_feat = self._sigma_mlp(x)
        return {'sigma': sigma, 'geo_feat': geo_feat}

    def density_sigma(self, x):
        """sigma of density() without the 15 geo features (the density-grid refresh needs nothing else; the fused
        kernel then skips their 60 bytes per sample of scattered stores).  No reference counterpart."""
        if self._fused_ok() and not torch.is_grad_enabled():
            enc = self.encoder
            tm = enc.get_planes_texel_major()
            packed = _field.pack_weights(self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight,
                                         self.color_net[1].weight, self.color_net[2].weight, enc.number_of_features,
                                         self.hidden_dim)
            x = x.detach().to(torch.float32).contiguous()
            sigma, _, _ = _field.field_forward(tm, x, None, packed, float(self.bound), enc.number_of_features,
                                               enc.plane_resolution, self.hidden_dim, geo_out=False)
            return sigma
        return self.density(x)['sigma']

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        # network.py:186-214 (masked colour query)
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d = d[mask]
            geo_feat = geo_feat[mask]
        h = self._color_mlp(d, geo_feat)
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
        else:
            rgbs = h
        return rgbs

    def get_params(self, lr):
        # network.py:217-243
        params = [
            {'params': self.encoder.parameters(), 'lr': lr},
            {'params': self.encoder_dir.parameters(), 'lr': lr},
        ]
        extra = {'weight_decay': self.mlp_weight_decay} if self.mlp_weight_decay > 0 else {}
        params += [
            {'params': self.sigma_net.parameters(), 'lr': lr, **extra},
            {'params': self.color_net.parameters(), 'lr': lr, **extra},
        ]
        return params


# the density() the sigma-only fused form of the grid refresh stands for (renderer._sigma_for_grid): a subclass that
# overrides density() is queried through its own
NeRFNetwork._stock_density = NeRFNetwork.density
