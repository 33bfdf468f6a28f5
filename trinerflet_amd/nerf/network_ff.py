"""NeRFNetwork of the reference's `--ff` switch (reconstruction/nerf/network_ff.py:10-148, selected at main_nerf.py:31-34):
the same constructor keywords, sub-module names (encoder, sigma_net.weights, encoder_dir, color_net.weights: state-dict
compatible) and methods forward / density / color / get_params, on this package's HIP encoder, HIP spherical harmonics and
renderer.  The two MLPs are `trinerflet_amd.ffmlp.FFMLP` (rocBLAS fp16 GEMMs): a module-API tier, SURVEY.md 8(f)-4 -- the
`--ff` architecture has one more matrix per network than network.py's, so the hand-written fused field does not apply and
TrainStep refuses it (`_fused_ok()` is False); `Trainer`-style loops drive it through autograd."""
import torch

from ..activation import trunc_exp
from ..encoding import get_encoder
from ..ffmlp import FFMLP
from .renderer import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self,
                 encoding="triplane_wavelet",
                 encoding_dir="sphere_harmonics",
                 num_layers=2,
                 hidden_dim=64,
                 geo_feat_dim=15,
                 num_layers_color=3,
                 hidden_dim_color=64,
                 bound=1,
                 **kwargs
                 ):
        super().__init__(bound, **kwargs)
        if self.bg_radius > 0:
            raise NotImplementedError("background model is not implemented for --ff (main_nerf.py:33)")
        # sigma network (network_ff.py:25-36)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound, bound=bound, **kwargs)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=self.hidden_dim,
                               num_layers=self.num_layers)
        # colour network (:38-50)
        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += self.geo_feat_dim + 1          # "a manual fixing to make it 32" (:43)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=self.hidden_dim_color,
                               num_layers=self.num_layers_color)
        self.bg_net = None

    def _fused_ok(self):
        return False                                         # another architecture than csrc/field.hip's

    def _sigma(self, x):
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _rgb(self, d, geo_feat):
        d = self.encoder_dir(d)
        p = torch.zeros_like(geo_feat[..., :1])              # manual input padding (:68, :115)
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, p], dim=-1)
        return torch.sigmoid(self.color_net(h))

    def forward(self, x, d):
        # network_ff.py:52-75
        sigma, geo_feat = self._sigma(x)
        return sigma, self._rgb(d, geo_feat)

    def density(self, x):
        # :77-90
        sigma, geo_feat = self._sigma(x)
        return {'sigma': sigma, 'geo_feat': geo_feat}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        # :93-134 (masked colour query)
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d = d[mask]
            geo_feat = geo_feat[mask]
        h = self._rgb(d, geo_feat)
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
        else:
            rgbs = h
        return rgbs

    def get_params(self, lr):
        # :137-148
        return [
            {'params': self.encoder.parameters(), 'lr': lr},
            {'params': self.sigma_net.parameters(), 'lr': lr},
            {'params': self.encoder_dir.parameters(), 'lr': lr},
            {'params': self.color_net.parameters(), 'lr': lr},
        ]
